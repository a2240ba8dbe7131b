#!/bin/bash
# Collects the round's committed measurements on a GPU box (run from the repo root through gpurun):
#   kernel trace + stats, two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no other trace domains), their summary, then the
#   plain bench line (which reads the fresh summary for roofline.traffic).  Outputs land in gpurun_out/final/.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_k -o bench --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline > $OUT/prof_k.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/p_f -o bench --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline > $OUT/prof_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/p_w -o bench --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline > $OUT/prof_w.log 2>&1
cd $ROOT
python tools/summarize_profile.py $OUT/p_k $OUT/p_f $OUT/p_w $OUT/r03_bench_b32 3 > $OUT/summary.txt 2>&1
cp $OUT/r03_bench_b32_pmc_summary.json profiles/ 2>/dev/null
rm -f $OUT/p_k/bench_kernel_trace.csv $OUT/p_f/bench_counter_collection.csv $OUT/p_w/bench_counter_collection.csv      # large raw files
timeout 300 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
MM_BENCH_FORCE_DIST=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 > $OUT/bench_dist.json 2> $OUT/bench_dist.err
tail -n 20 $OUT/summary.txt
tail -c 600 $OUT/bench.json; echo; tail -c 300 $OUT/bench_dist.json
