#!/bin/bash
# session-3 GPU call 7: cross_fold with 64 queries per workgroup vs 32
cd /root/repo
O=gpurun_out/s3r7; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_fuzz_forward.py tests/test_gpu_base_size.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
bash tools/r4_kstats.sh s3r7_new 0 > $O/kstats_new.log 2>&1; grep "cross_fold\|gemm_wide_fused" gpurun_out/s3r7_new_kstats.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > $O/bench_q64_$i.log 2>&1
  MM_CROSS_FOLD_Q32=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > $O/bench_q32_$i.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s3r7/bench_*.log')):
    l=[x for x in open(f) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['value'], d['ms_per_step'])
    else: print(f, 'NO LINE', open(f).read()[-400:])
PY
