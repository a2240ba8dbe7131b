cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused_sampling.py tests/test_gpu_fuzz_sampling.py tests/test_gpu_fuzz_generate.py -q 2>&1 | tail -6
bash tools/r4_kstats.sh r4g 0 | head -9
python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],2), round(d['decode_loop_ms_per_step'],2), d['fused_sampling'])"
