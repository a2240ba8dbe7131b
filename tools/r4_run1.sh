set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_bf16x3.py -q -k "f16 or split_rows" -s 2>&1 | tail -40 > gpurun_out/r4a/t1.log
python -m pytest tests/test_gpu_parity_mode.py -q -k "fp32_checkpoint or forward_and_guidance or generate_ids" -s 2>&1 | tail -60 > gpurun_out/r4a/t2.log
python -m pytest tests/test_gpu_base_size.py -q -k "base_size or dim_256" -s 2>&1 | grep -v "^$" | tail -120 > gpurun_out/r4a/t3.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -c 3000 gpurun_out/r4a/t1.log; tail -c 1500 gpurun_out/r4a/t2.log; tail -c 2500 gpurun_out/r4a/t3.log
