#!/bin/bash
# session-3 GPU call 1: new dense gather of the fused sampler -- correctness, microbench A/B + phase ablations, bench line
cd /root/repo
O=gpurun_out/s3r1; mkdir -p $O
python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fuzz_sampling.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
for v in hip exp9 exp1 exp2 exp3 exp10; do
  echo "== $v" >> $O/fused_bench.log
  MM_LIB=/root/repo/muse_maskgit_pytorch_amd/libmuse_$v.so timeout 300 python tools/fused_bench.py >> $O/fused_bench.log 2>&1
done
grep -E "==|fused_sample|fail flag" $O/fused_bench.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; tail -c 3000 $O/bench.log
