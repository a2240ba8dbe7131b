# usage: bash tools/r5_kstats.sh <tag> [bench.py args ...]  -> gpurun_out/<tag>_kstats.txt : per-kernel calls / avg us / ms per generate (3 generates traced: 1 warm-up + 2 timed)
# e.g.  bash tools/r5_kstats.sh r05_f16x2_fp32w --precision f16x2         bash tools/r5_kstats.sh r05_f16x2_bf16w --precision f16x2 --bf16-round-weights
TAG=$1; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/p_k -o bench --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal "$@" > $OUT/prof_k.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob('$OUT/p_k/**/bench_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = 0
with open('gpurun_out/${TAG}_kstats.txt', 'w') as o:
    o.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal $*\n# ms per generate = total / 3 traced generates\n')
    for r in rows[:45]:
        ms = float(r['TotalDurationNs']) / 1e6 / 3
        tot += ms
        o.write(f"{ms:8.2f} ms/gen {float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls'])//3:5d}  {r['Name'][:150]}\n")
    o.write(f'sum of top 45: {tot:.2f} ms/gen\n')
PY
rm -rf $OUT/p_k
tail -1 $OUT/prof_k.log | cut -c1-400
head -30 gpurun_out/${TAG}_kstats.txt
