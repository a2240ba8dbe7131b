mkdir -p gpurun_out/r5f
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fused_sampling.py -q -s -k "cross_attention or layernorm_dim_fold or graph or quantile" > gpurun_out/r5f/new_tests.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed\|Error\|\[cross-attention operator\|^E  " gpurun_out/r5f/new_tests.log | cut -c1-250 | tail -40
timeout 600 python bench.py --steps 10 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg > gpurun_out/r5f/bench.json 2> gpurun_out/r5f/bench.err; echo "bench rc $?"; tail -3 gpurun_out/r5f/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5f/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'sampler', d['roofline_hbm']['avg_launch_ms'])
print('off_ideal', json.dumps(d.get('off_ideal'))[:1500])
PY
