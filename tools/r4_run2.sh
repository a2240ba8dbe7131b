cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -15 > gpurun_out/r4c/t_model.log
python -m pytest tests/test_gpu_base_size.py -q -k "generate_at_base_size and fp32w" -s 2>&1 | grep -E "base-size parity|passed|failed|Error" | cut -c1-260 > gpurun_out/r4c/t_fp32w.log
python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_fuzz_forward.py -q 2>&1 | tail -8 > gpurun_out/r4c/t_tier.log
python -m pytest tests/test_gpu_zz_full_size_determinism.py tests/test_gpu_ops.py -q 2>&1 | tail -8 > gpurun_out/r4c/t_det.log
python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > gpurun_out/r4c/bench_fold.json 2> gpurun_out/r4c/bench_fold.err
MM_DEBUG=0x20000000 python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > gpurun_out/r4c/bench_nofold.json 2> gpurun_out/r4c/bench_nofold.err
python bench.py --steps 10 --warmup 3 --no-parity-tier --no-cpu-baseline > gpurun_out/r4c/bench_fold2.json 2>> gpurun_out/r4c/bench_fold.err
for f in t_model t_fp32w t_tier t_det; do echo "== $f"; tail -12 gpurun_out/r4c/$f.log; done
for f in bench_fold bench_nofold bench_fold2; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r4c/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],2), round(d['decode_loop_ms_per_step'],2))"; done
