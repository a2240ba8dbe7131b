mkdir -p gpurun_out/r5u
timeout 900 python -m pytest tests/test_gpu_attention_small.py tests/test_gpu_parity_mode.py tests/test_gpu_bf16x3.py -q -x -s > gpurun_out/r5u/t1.log 2>&1; echo "pytest 1 rc $?"; grep -E "attention_small\]|passed|failed" gpurun_out/r5u/t1.log | tail -12
timeout 900 python -m pytest tests/test_gpu_base_size.py tests/test_gpu_fuzz_forward.py -q -x -k "(f16x2 or bf16x3 or fuzz) and not vae" > gpurun_out/r5u/t2.log 2>&1; echo "pytest 2 rc $?"; tail -2 gpurun_out/r5u/t2.log
for w in "" "--bf16-round-weights"; do
for d in 0 4 0 4; do
  echo "== MM_DEBUG2=$d $w"; MM_DEBUG2=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
done; done 2>&1 | tee gpurun_out/r5u/ab.log
timeout 600 python tools/determinism_stress.py --precision f16x2 --configs 0 --iters 60 --gen-iters 10 --batch 32 2>&1 | tail -8 | tee gpurun_out/r5u/stress_f16x2.txt
