mkdir -p gpurun_out/r2p
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2p/tests.log 2>&1; tail -4 gpurun_out/r2p/tests.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r2p/bench_c2.json 2> gpurun_out/r2p/bench_c2.err; python -c "
import json; d=json.load(open('gpurun_out/r2p/bench_c2.json')); print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm']['avg_launch_ms'], d['fused_sampling'], d['executed_mfma_frac_decode_loop'])"
for c in c4 c5; do timeout 300 python bench.py --config $c --steps 4 --warmup 1 > gpurun_out/r2p/bench_$c.json 2> gpurun_out/r2p/bench_$c.err; python -c "
import json; d=json.load(open('gpurun_out/r2p/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['fused_sampling'])"; done
