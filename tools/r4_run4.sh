cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r4e/t_all.log
tail -25 gpurun_out/r4e/t_all.log
bash tools/r4_kstats.sh r4e_fold 0 | head -14
python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold', round(d['value'],1), round(d['ms_per_step'],2), round(d['decode_loop_ms_per_step'],2))"
