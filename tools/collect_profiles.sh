#!/bin/bash
# Collects the round's committed measurements on a GPU box (run from the repo root through gpurun):
#   kernel trace + stats, two HBM PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no other trace domains), two SQ passes
#   (matrix pipe + LDS conflicts; wait / issue cycles), their summaries, then the plain bench lines (which read the fresh PMC summary
#   for roofline.traffic).  Outputs land in gpurun_out/final/ under the names they are committed with in profiles/.
#   usage: [ROUND=r04] [FULL=1] tools/collect_profiles.sh        FULL=1 adds the c4 / c5 / c5 --fp8 / --train lines
set -u
ROOT=$PWD
R=${ROUND:-r06}
OUT=$ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
# which box (the pool's boxes differ by up to 9 % on the headline): every file of this collection comes from the ONE box recorded here
{ echo "collected $(date -u +%Y-%m-%dT%H:%M:%SZ) on host $(hostname)"; cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1 | sed 's/^/gpu unique_id /';
  rocm-smi --showuniqueid --showproductname --showpower --showclocks --showmaxpower 2>/dev/null | grep -v "^=\|^$" | head -24; } > $OUT/${R}_box.txt 2>&1
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_k -o bench --output-format csv -- $BENCH > $OUT/prof_k.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/p_f -o bench --output-format csv -- $BENCH > $OUT/prof_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/p_w -o bench --output-format csv -- $BENCH > $OUT/prof_w.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p_s1 -o bench --output-format csv -- $BENCH > $OUT/prof_s1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/p_s2 -o bench --output-format csv -- $BENCH > $OUT/prof_s2.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d $OUT/p_s3 -o bench --output-format csv -- $BENCH > $OUT/prof_s3.log 2>&1
cd $ROOT
python tools/summarize_profile.py $OUT/p_k $OUT/p_f $OUT/p_w $OUT/${R}_bench_b32 3 > $OUT/${R}_bench_b32_summary.txt 2>&1
python tools/summarize_profile.py --sq $OUT/p_s1 $OUT/p_s2 $OUT/p_s3 $OUT/${R}_bench_b32_sq_counters.json >> $OUT/${R}_bench_b32_summary.txt 2>&1
cp $OUT/${R}_bench_b32_pmc_summary.json profiles/ 2>/dev/null
rm -f $OUT/p_k/bench_kernel_trace.csv $OUT/p_*/bench_counter_collection.csv      # large raw files
timeout 400 python bench.py --steps 10 --warmup 2 > $OUT/${R}_bench_b32.json 2> $OUT/bench.err
MM_BENCH_FORCE_DIST=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal > $OUT/${R}_bench_b32_rccl_world1.json 2> $OUT/bench_dist.err
if [ "${FULL:-0}" = 1 ]; then
  timeout 200 python bench.py --config c4 --batch 8 --steps 3 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal > $OUT/${R}_bench_c4_b8.json 2> $OUT/bench_c4.err
  timeout 200 python bench.py --config c5 --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal > $OUT/${R}_bench_c5_b32.json 2> $OUT/bench_c5.err
  timeout 200 python bench.py --config c5 --fp8 --steps 5 --warmup 2 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal > $OUT/${R}_bench_c5_b32_fp8.json 2> $OUT/bench_c5f.err
  timeout 200 python bench.py --train --steps 10 --warmup 3 > $OUT/${R}_bench_train_b32.json 2> $OUT/bench_train.err
  bash tools/r5_kstats.sh ${R}_f16x2_fp32w_final --precision f16x2 > $OUT/k_tier1.log 2>&1
  bash tools/r5_kstats.sh ${R}_f16x2_bf16w_final --precision f16x2 --bf16-round-weights > $OUT/k_tier2.log 2>&1
  # round 6: the tier's HBM counters (parity_tier.roofline.traffic): kernel trace + FETCH_SIZE + WRITE_SIZE passes of the general-fp32-checkpoint tier
  TB="python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2"
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/t_k -o bench --output-format csv -- $TB > $OUT/prof_tk.log 2>&1
    timeout 500 rocprofv3 --pmc FETCH_SIZE -d $OUT/t_f -o bench --output-format csv -- $TB > $OUT/prof_tf.log 2>&1
    timeout 500 rocprofv3 --pmc WRITE_SIZE -d $OUT/t_w -o bench --output-format csv -- $TB > $OUT/prof_tw.log 2>&1 )
  python tools/summarize_profile.py $OUT/t_k $OUT/t_f $OUT/t_w $OUT/${R}_f16x2 3 > $OUT/${R}_f16x2_summary.txt 2>&1
  cp $OUT/${R}_f16x2_pmc_summary.json profiles/ 2>/dev/null
  rm -f $OUT/t_k/bench_kernel_trace.csv $OUT/t_*/bench_counter_collection.csv
fi
{ echo "# box: $(head -2 $OUT/${R}_box.txt | tr '\n' ' ')"; cat $OUT/${R}_bench_b32_summary.txt; } > $OUT/tmp_sum && mv $OUT/tmp_sum $OUT/${R}_bench_b32_summary.txt
tail -n 45 $OUT/${R}_bench_b32_summary.txt
for f in $OUT/${R}_bench_*.json; do echo "== $f"; tail -c 500 $f; echo; done
