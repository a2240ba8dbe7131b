#!/bin/bash
# Round 5, second half: what changed is the 'f16x2' tier (term sharing, csrc/gemm_terms.hip) -- the bf16 engine's kernels are instruction-identical to the ones the
# r05 kernel-stat / PMC / SQ files were collected on.  This collects: the -m gpu suite, the driver-style bench line, the tier's kernel stats on both checkpoints and
# (SQ=1) the tier's SQ counter passes (matrix pipe, LDS conflicts; wait / issue cycles).  Outputs: gpurun_out/final/ under the names they are committed with in profiles/.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/r05_gpu_tests.txt 2>&1; echo "gpu suite rc $?"; tail -3 $OUT/r05_gpu_tests.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/r05_bench_b32.json 2> $OUT/bench.err; tail -c 300 $OUT/r05_bench_b32.json; echo
bash tools/r5_kstats.sh r05_f16x2_fp32w_final --precision f16x2 > $OUT/k_tier1.log 2>&1
bash tools/r5_kstats.sh r05_f16x2_bf16w_final --precision f16x2 --bf16-round-weights > $OUT/k_tier2.log 2>&1
if [ "${SQ:-0}" = 1 ]; then
  cd /tmp && export TMPDIR=/tmp
  BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal --precision f16x2"
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p_s1 -o bench --output-format csv -- $BENCH > $OUT/prof_s1.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/p_s2 -o bench --output-format csv -- $BENCH > $OUT/prof_s2.log 2>&1
  cd $ROOT
  python tools/summarize_profile.py --sq $OUT/p_s1 $OUT/p_s2 $OUT/r05_f16x2_fp32w_sq_counters.json > $OUT/sq_summary.txt 2>&1
  rm -rf $OUT/p_s1 $OUT/p_s2
  tail -20 $OUT/sq_summary.txt | cut -c1-200
fi
head -20 gpurun_out/r05_f16x2_fp32w_final_kstats.txt | cut -c1-150
