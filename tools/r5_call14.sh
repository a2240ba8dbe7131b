mkdir -p gpurun_out/r5n
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r5n/bench.json 2> gpurun_out/r5n/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5n/bench.json'))
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), 'sampler', round(d['roofline_hbm']['avg_launch_ms'],4))
print(' graph', d['hip_graph_replay'].get('value'), 'parity', d['parity_tier']['value'], d['parity_tier']['fp32_checkpoint']['value'], 'cpu', d['cpu_baseline']['value'])
print(' off', {k:(round(v['value'],1), round(v['x_headline_time'],3)) for k,v in d['off_ideal'].items() if isinstance(v,dict)})
PY
