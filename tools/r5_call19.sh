mkdir -p gpurun_out/r5s
timeout 600 python -m pytest tests/test_gpu_terms_gemm.py -q -x -s -k "convolution" > gpurun_out/r5s/t_conv.log 2>&1; echo "conv pytest rc $?"; grep -E "^\[terms\]|\.\[terms\]|passed|failed|Error" gpurun_out/r5s/t_conv.log | tail -8
timeout 600 python -m pytest tests/test_gpu_base_size.py tests/test_gpu_parity_mode.py -q -x -s -k "vae or vqgan" > gpurun_out/r5s/t_vae.log 2>&1; echo "vae pytest rc $?"; grep -E "f16x2|passed|failed" gpurun_out/r5s/t_vae.log | tail -8
python tools/terms_gemm_timing.py 2>&1 | tee gpurun_out/r5s/terms_gemm_timing.txt
for w in "" "--bf16-round-weights"; do
  for d in 0 2; do
  echo "== MM_DEBUG2=$d $w"; MM_DEBUG2=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
  done
done 2>&1 | tee gpurun_out/r5s/ab.log
