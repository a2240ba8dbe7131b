mkdir -p gpurun_out/r5p
timeout 600 python -m pytest tests/test_gpu_terms_gemm.py -q -x -s > gpurun_out/r5p/t_terms.log 2>&1; echo "terms pytest rc $?"; grep -E "^\[terms\]|passed|failed|Error|error" gpurun_out/r5p/t_terms.log | tail -30
timeout 900 python -m pytest tests/test_gpu_base_size.py -q -x -s -k "f16x2" > gpurun_out/r5p/t_base.log 2>&1; echo "base pytest rc $?"; grep -E "f16x2|passed|failed" gpurun_out/r5p/t_base.log | tail -40
for w in "" "--bf16-round-weights"; do
  echo "== new $w"; timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
  echo "== old $w"; MM_DEBUG2=2 timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
done 2>&1 | tee gpurun_out/r5p/ab.log
