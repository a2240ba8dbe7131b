mkdir -p gpurun_out/r5a
H=tools/build/gemm_harness
for v in 0 230 231 232 240 241 130; do echo "== MM_PP=$v"; MM_PP=$v timeout 90 $H logits 2>&1 | tail -3; done > gpurun_out/r5a/harness.log 2>&1
for d in 0 24000; do echo "== MM_PP=130 delay $d"; MM_PP=130 MM_PP_DELAY=$d timeout 90 $H logits 2>&1 | tail -3; done >> gpurun_out/r5a/harness.log 2>&1
cat gpurun_out/r5a/harness.log
timeout 900 python -m pytest tests/test_gpu_base_size.py -q -s -k "base_size or unscanned or vqgan" > gpurun_out/r5a/base_size.log 2>&1; echo "pytest rc $?"
grep "parity\]\|passed\|failed\|Error\|assert" gpurun_out/r5a/base_size.log | cut -c1-330 | tail -60
bash tools/r5_kstats.sh r05_f16x2_fp32w --precision f16x2 > gpurun_out/r5a/k1.log 2>&1; tail -32 gpurun_out/r5a/k1.log
bash tools/r5_kstats.sh r05_f16x2_bf16w --precision f16x2 --bf16-round-weights > gpurun_out/r5a/k2.log 2>&1; tail -32 gpurun_out/r5a/k2.log
