mkdir -p gpurun_out/r5r
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5r/gpu_tests.txt 2>&1; echo "full gpu suite rc $?"; tail -5 gpurun_out/r5r/gpu_tests.txt
for w in "" "--bf16-round-weights"; do
  for d in 0 2; do
  echo "== MM_DEBUG2=$d $w"; MM_DEBUG2=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
  done
done 2>&1 | tee gpurun_out/r5r/ab.log
