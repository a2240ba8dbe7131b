mkdir -p gpurun_out/r5h
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5h/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/r5h/gpu_tests.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 2 --no-parity-tier --no-cpu-baseline > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5h/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'sampler', d['roofline_hbm']['avg_launch_ms'])
print('fused', d['fused_sampling'])
print('graph', json.dumps(d.get('hip_graph_replay'))[:200])
print('off_ideal', json.dumps(d.get('off_ideal'))[:1600])
PY
