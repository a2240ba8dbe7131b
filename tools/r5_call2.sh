mkdir -p gpurun_out/r5b
H=tools/build/gemm_harness; HT=tools/build/gemm_harness_t
( for v in 0 230 130; do echo "== stamps MM_PP=$v"; MM_PP=$v timeout 90 $HT stamps 2>&1 | tail -16; done
  for a in 0 1 3 5 9 13 11 7; do echo "== MM_PP=230 ABL=$a"; MM_PP=230 MM_PP_ABL=$a timeout 90 $H logits 2>&1 | grep logits; done
  for a in 1 5; do echo "== MM_PP=130 ABL=$a"; MM_PP=130 MM_PP_ABL=$a timeout 90 $H logits 2>&1 | grep logits; done
  for a in 1; do echo "== MM_PP=240 ABL=$a"; MM_PP=240 MM_PP_ABL=$a timeout 90 $H logits 2>&1 | grep logits; done
) > gpurun_out/r5b/stamps.log 2>&1
cat gpurun_out/r5b/stamps.log
