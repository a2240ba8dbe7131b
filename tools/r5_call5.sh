mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fused_sampling.py -q -s -k "layernorm_dim_fold or graph or quantile" > gpurun_out/r5e/new_tests.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed\|Error\|\[ln-fold\|\[fused bound\|^E  " gpurun_out/r5e/new_tests.log | cut -c1-330 | tail -30
timeout 600 python bench.py --steps 10 --warmup 2 --no-parity-tier --no-cpu-baseline > gpurun_out/r5e/bench.json 2> gpurun_out/r5e/bench.err; echo "bench rc $?"; tail -3 gpurun_out/r5e/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5e/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'sampler', d['roofline_hbm']['avg_launch_ms'])
print('fused', d['fused_sampling'])
print('graph', json.dumps(d.get('hip_graph_replay'))[:300])
print('off_ideal', json.dumps(d.get('off_ideal'))[:1500])
PY
