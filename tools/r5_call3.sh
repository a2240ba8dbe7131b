mkdir -p gpurun_out/r5c
H=tools/build/gemm_harness; HT=tools/build/gemm_harness_t
( echo "== pairs MM_PP=130 delay 12000"; MM_PP=130 timeout 90 $HT pairs 2>&1 | tail -24
  echo "== pairs MM_PP=134 delay 12000 (second-half guess)"; MM_PP=134 timeout 90 $HT pairs 2>&1 | tail -3
  echo "== pairs MM_PP=130 delay 0"; MM_PP=130 MM_PP_DELAY=0 timeout 90 $HT pairs 2>&1 | tail -3
  for d in 0 6000 12000 18000 26000 40000; do echo "== MM_PP=130 delay $d"; MM_PP=130 MM_PP_DELAY=$d timeout 90 $H logits 2>&1 | grep logits; done
  echo "== MM_PP=0"; timeout 90 $H logits 2>&1 | grep logits
) > gpurun_out/r5c/pairs.log 2>&1
cat gpurun_out/r5c/pairs.log
