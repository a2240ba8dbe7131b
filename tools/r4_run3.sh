cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_model.py tests/test_gpu_fuzz_forward.py -q 2>&1 | tail -12 > gpurun_out/r4d/t_model.log
python -m pytest tests/test_gpu_zz_full_size_determinism.py -q 2>&1 | tail -8 > gpurun_out/r4d/t_det.log
for f in t_model t_det; do echo "== $f"; tail -12 gpurun_out/r4d/$f.log; done
bash tools/r4_kstats.sh r4d_fold 0
bash tools/r4_kstats.sh r4d_nofold 0x20000000
python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold', round(d['value'],1), round(d['ms_per_step'],2), round(d['decode_loop_ms_per_step'],2))"
MM_DEBUG=0x20000000 python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nofold', round(d['value'],1), round(d['ms_per_step'],2), round(d['decode_loop_ms_per_step'],2))"
