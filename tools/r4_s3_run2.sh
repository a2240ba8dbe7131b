#!/bin/bash
# session-3 GPU call 2: finisher (Gumbel transform, merged collect / squeeze sweep, statistics prefetch) + packed fp32 in the logits emission
cd /root/repo
O=gpurun_out/s3r2; mkdir -p $O
python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fuzz_sampling.py tests/test_gpu_ops.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
for v in hip exp20 exp21 exp1 exp2 hip exp20 exp21; do
  echo "== $v" >> $O/fused_bench.log
  MM_LIB=/root/repo/muse_maskgit_pytorch_amd/libmuse_$v.so timeout 300 python tools/fused_bench.py >> $O/fused_bench.log 2>&1
done
grep -E "==|fused_sample|fused guidance|fail flag" $O/fused_bench.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; tail -c 600 $O/bench.log | head -c 300; python - <<'PY'
import json
l=[x for x in open('gpurun_out/s3r2/bench.log') if x.startswith('{')][-1]
d=json.loads(l); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_hbm']['avg_launch_ms'])
PY
