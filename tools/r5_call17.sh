mkdir -p gpurun_out/r5q
timeout 900 python -m pytest tests/test_gpu_terms_gemm.py tests/test_gpu_bf16x3.py -q -x -s > gpurun_out/r5q/t_terms.log 2>&1; echo "terms+bf16x3 pytest rc $?"; grep -E "residual|generate B|passed|failed|Error|error" gpurun_out/r5q/t_terms.log | tail -30
timeout 900 python -m pytest tests/test_gpu_base_size.py -q -x -s -k "f16x2 and (base_size or unscanned)" > gpurun_out/r5q/t_base.log 2>&1; echo "base pytest rc $?"; grep -E "passed|failed" gpurun_out/r5q/t_base.log | tail -5
for w in "" "--bf16-round-weights"; do
  echo "== new $w"; timeout 300 python bench.py --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2 $w 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('decode_loop_ms_per_step'))"
done 2>&1 | tee gpurun_out/r5q/ab.log
bash tools/r5_kstats.sh r05_f16x2_fp32w_terms --precision f16x2 | cut -c1-170
bash tools/r5_kstats.sh r05_f16x2_bf16w_terms --precision f16x2 --bf16-round-weights | cut -c1-170
