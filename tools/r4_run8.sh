cd $GRAFT_REPO_ROOT
tools/build/gemm_harness 0 out w2 xq 2>&1 | tail -5
python -m pytest tests/test_gpu_ops.py tests/test_gpu_zz_full_size_determinism.py tests/test_gpu_fuzz_vae.py -q 2>&1 | tail -4
bash tools/r4_kstats.sh r4i 0 | head -9
python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],2), round(d['decode_loop_ms_per_step'],2))"
