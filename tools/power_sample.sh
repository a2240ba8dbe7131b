#!/bin/bash
# Samples rocm-smi (shader clock, power) every 0.5 s while the bench command runs (200 generates: the steady state of the decode loop): evidence for the
# power-limited clock the loop runs at.  Output: gpurun_out/power_sample.txt = the samples with non-idle shader clocks.
OUT=${1:-gpurun_out/power_sample.txt}
mkdir -p $(dirname $OUT); : > $OUT.raw
python bench.py --steps 200 --warmup 5 --no-parity-tier --no-cpu-baseline > /dev/null 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | tr '\n' ' ' >> $OUT.raw; echo >> $OUT.raw
  sleep 0.5
done
awk '{ if (match($0, /\(([0-9]+)Mhz\)/)) { s = substr($0, RSTART + 1, RLENGTH - 5); if (s + 0 > 400) print } }' $OUT.raw > $OUT
echo "busy samples: $(wc -l < $OUT) of $(wc -l < $OUT.raw)"; head -n 12 $OUT
