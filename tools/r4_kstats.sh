# usage: bash tools/r4_kstats.sh <tag> [MM_DEBUG value]  -> gpurun_out/<tag>_kstats.txt : per-kernel calls / avg us / ms per generate (3 generates traced)
TAG=$1; DBG=${2:-0}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MM_DEBUG=$DBG timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_k -o bench --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal > $OUT/prof_k.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob('$OUT/p_k/**/bench_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = 0
with open('gpurun_out/${TAG}_kstats.txt', 'w') as o:
    for r in rows[:40]:
        ms = float(r['TotalDurationNs']) / 1e6 / 3
        tot += ms
        o.write(f"{ms:8.2f} ms/gen {float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls'])//3:5d}  {r['Name'][:110]}\n")
    o.write(f'sum of top 40: {tot:.2f} ms/gen\n')
PY
rm -rf $OUT/p_k
head -24 gpurun_out/${TAG}_kstats.txt
