mkdir -p gpurun_out/r5g
H=tools/build/gemm_harness; HT=tools/build/gemm_harness_t
( echo "== logits (staged candidate write-out)"; timeout 90 $H logits sample 2>&1 | tail -4
  echo "== stamps"; timeout 90 $HT stamps 2>&1 | tail -14 ) > gpurun_out/r5g/harness.log 2>&1
cat gpurun_out/r5g/harness.log
timeout 900 python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fuzz_sampling.py tests/test_gpu_fuzz_generate.py -q -x > gpurun_out/r5g/tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r5g/tests.log
for bnd in quantile gaussian quantile gaussian; do timeout 300 python bench.py --steps 10 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --fused-bound $bnd 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$bnd', round(d['value'],1), 'img/s; logits', round(d['roofline']['avg_launch_ms'],4), 'ms frac', round(d['roofline']['frac'],3), '; sampler', round(d['roofline_hbm']['avg_launch_ms'],4))"; done 2>&1 | tee gpurun_out/r5g/ab.log
