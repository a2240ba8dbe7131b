mkdir -p gpurun_out/r5t
timeout 900 python -m pytest tests/test_gpu_base_size.py tests/test_gpu_terms_gemm.py tests/test_gpu_bf16x3.py tests/test_gpu_zz_full_size_determinism.py tests/test_gpu_fuzz_forward.py -q -x -k "f16x2 or terms or tier or determinism or fuzz" > gpurun_out/r5t/t.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5t/t.log
for d in 0 2 0 2; do
  echo "== MM_DEBUG2=$d"; MM_DEBUG2=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
done 2>&1 | tee gpurun_out/r5t/ab.log
