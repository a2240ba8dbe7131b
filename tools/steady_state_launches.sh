#!/bin/bash
# Launches of ONE steady-state generate (+ VAE decode) of the bench's timed region: kernel-trace two run lengths (2 and 5 timed steps) and divide the difference
# of every kernel's call count by 3 -- setup, warm-up and one-off launches cancel.  Output: gpurun_out/steady_state_launches.txt (committed as profiles/<round>_steady_state_launches.txt)
ROOT=$PWD; OUT=$ROOT/gpurun_out/steady; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 2 5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p$n -o bench --output-format csv -- python $ROOT/bench.py --steps $n --warmup 1 --no-parity-tier --no-cpu-baseline --no-graph-leg --no-off-ideal > $OUT/prof$n.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob
def load(n):
    f = glob.glob(f'gpurun_out/steady/p{n}/**/bench_kernel_stats.csv', recursive=True)[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load(2), load(5)
rows = []
for k in sorted(set(a) | set(b)):
    dc = (b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0]) / 3.0
    dt = (b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1]) / 3.0 / 1e6
    if abs(dc) > 1e-9:
        rows.append((dc, dt, k))
own = [r for r in rows if '(anonymous namespace)' in r[2] or 'mm_' in r[2]]
other = [r for r in rows if r not in own]
with open('gpurun_out/steady_state_launches.txt', 'w') as o:
    o.write(f'launches per steady-state generate + VAE decode (difference of a 5-step and a 2-step trace, / 3): {sum(r[0] for r in rows):.1f} '
            f'({sum(r[1] for r in rows):.2f} ms of kernel time); kernels of this library {sum(r[0] for r in own):.1f}, others {sum(r[0] for r in other):.1f}\n')
    for title, rs in (('this library', own), ('torch / runtime', other)):
        o.write(f'-- {title}\n')
        for dc, dt, k in sorted(rs, key=lambda r: -r[1]):
            o.write(f'{dc:8.1f} x  {dt:8.3f} ms  {k[:120]}\n')
print(open('gpurun_out/steady_state_launches.txt').read()[:3500])
PY
rm -rf $OUT/p2 $OUT/p5
