mkdir -p gpurun_out/r5v
for d in 0 0x10 0x20 0x30; do echo "== MM_DEBUG2=$d"; MM_DEBUG2=$d python tools/terms_gemm_timing.py 2>&1 | grep -E "q\|k\|v|FF w1" ; done | tee gpurun_out/r5v/dpos_timing.txt
for d in 0 0x10 0x20 0x30 0 0x10; do
  echo "== MM_DEBUG2=$d"; MM_DEBUG2=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
done 2>&1 | tee gpurun_out/r5v/dpos_ab.log
