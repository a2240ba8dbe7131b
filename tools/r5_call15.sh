mkdir -p gpurun_out/r5o
( for i in 1 2 3; do echo "== new"; timeout 90 tools/build/gemm_harness logits 2>&1 | tail -2; echo "== old"; timeout 90 tools/build/gemm_harness_old logits 2>&1 | tail -2; done
  echo "== stamps new"; timeout 90 tools/build/gemm_harness_t stamps 2>&1 | grep "mean over" ) > gpurun_out/r5o/ab.log 2>&1
cat gpurun_out/r5o/ab.log
timeout 600 python -m pytest tests/test_gpu_fused_sampling.py -q -x > gpurun_out/r5o/t.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r5o/t.log
