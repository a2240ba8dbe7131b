#!/bin/bash
# session-3 GPU call 6: cross-attention block as one kernel (q projection inside)
cd /root/repo
O=gpurun_out/s3r6; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_fuzz_forward.py tests/test_gpu_base_size.py tests/test_gpu_fuzz_generate.py tests/test_gpu_zz_full_size_determinism.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -5 $O/tests.log
bash tools/r4_kstats.sh s3r6_new 0 > $O/kstats_new.log 2>&1; head -12 gpurun_out/s3r6_new_kstats.txt; grep "cross_fold" gpurun_out/s3r6_new_kstats.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > $O/bench_new$i.log 2>&1
  MM_DEBUG=0x80000000 timeout 300 python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > $O/bench_old$i.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s3r6/bench_*.log')):
    l=[x for x in open(f) if x.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['value'], d['ms_per_step'])
    else: print(f, 'NO LINE', open(f).read()[-400:])
PY
