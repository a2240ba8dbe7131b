TAG=$1
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_k -o bench --output-format csv -- python $ROOT/bench.py --train --steps 4 --warmup 2 > $OUT/prof_k.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob('$OUT/p_k/**/bench_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = 0
with open('gpurun_out/${TAG}_kstats.txt', 'w') as o:
    for r in rows[:45]:
        ms = float(r['TotalDurationNs']) / 1e6 / 6
        tot += ms
        o.write(f"{ms:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us x{int(r['Calls'])/6:7.1f}  {r['Name'][:120]}\n")
    o.write(f'sum of top 45: {tot:.2f} ms/step\n')
PY
rm -rf $OUT/p_k
head -46 gpurun_out/${TAG}_kstats.txt
