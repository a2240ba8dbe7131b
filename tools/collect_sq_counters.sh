#!/bin/bash
# MFMA-pipe and LDS counters of the bench command (separate --pmc passes, no trace domains): per kernel, per launch.
#   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES -> matrix-pipe utilisation;  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE -> LDS conflict share
# Output: gpurun_out/sq/summary.json (copied to profiles/ by hand).
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout 400 rocprofv3 --pmc $c -d $OUT/$c -o bench --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-tier > $OUT/$c.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, collections, json, glob
out = collections.defaultdict(dict)
for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):
    fs = glob.glob(f'gpurun_out/sq/{c}/**/bench_counter_collection.csv', recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        a = acc[r['Kernel_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items():
        out[k][c] = v / n
        out[k]['launches'] = n
keep = {}
for k, v in out.items():
    if 'SQ_BUSY_CU_CYCLES' in v and v.get('SQ_BUSY_CU_CYCLES', 0) > 0 and ('gemm' in k or 'attention' in k or 'sample' in k):
        v['mfma_busy_per_cu_busy'] = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / v['SQ_BUSY_CU_CYCLES']
        v['mfma_pipe_utilisation'] = v['mfma_busy_per_cu_busy'] / 4.0      # MFMA busy cycles are summed over the 4 SIMDs of a CU
        if v.get('SQ_LDS_IDX_ACTIVE'):
            v['lds_conflict_share'] = v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE']
        keep[k[:90]] = v
json.dump(keep, open('gpurun_out/sq/summary.json', 'w'), indent=1)
for k, v in sorted(keep.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', 0) * kv[1]['launches'])[:12]:
    print(f"{k[:70]:70s} launches {v['launches']:5d}  mfma/cu_busy {v.get('mfma_busy_per_cu_busy', 0):.3f}  lds conflict share {v.get('lds_conflict_share', 0):.3f}")
PY
rm -rf $OUT/SQ_*/
