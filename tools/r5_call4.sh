mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fused_sampling.py -q -x -s -k "cross_attention_block or layernorm_dim_fold or graph" > gpurun_out/r5d/new_tests.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed\|Error\|error\|\[ln-fold\|\[cross-attention operator\|assert" gpurun_out/r5d/new_tests.log | cut -c1-300 | tail -40
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r5d/bench.json 2> gpurun_out/r5d/bench.err; echo "bench rc $?"; tail -3 gpurun_out/r5d/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5d/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print('graph', json.dumps(d.get('hip_graph_replay'))[:600])
print('off_ideal', json.dumps(d.get('off_ideal'))[:1500])
pt=d.get('parity_tier',{})
print('parity', pt.get('value'), pt.get('fp32_checkpoint',{}).get('value'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
