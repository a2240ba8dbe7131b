#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the training step, summarised by tools/train_trace_summary.py into gpurun_out/<dir>/train_trace.txt (run on the GPU box)
OUT=${1:-gpurun_out/train_trace}
mkdir -p $OUT; export TMPDIR=/tmp
python bench.py --train --steps 20 --warmup 5 > $OUT/train.json 2> $OUT/train.err
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/prof -o tr --output-format csv -- python bench.py --train --steps 6 --warmup 2 > $OUT/prof.log 2>&1
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); MT=$(find $OUT/prof -name "*memory_copy_trace.csv" | head -1)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
( echo "# python bench.py --train --steps 20 --warmup 5:"; python -c "import json; d=json.load(open('$OUT/train.json')); print('#   %.3f ms per step, %.0f tokens/s' % (d['ms_per_step'], d['value']))"
  echo "# rocprofv3 --kernel-trace --memory-copy-trace -- python bench.py --train --steps 6 --warmup 2 (the profiler lengthens the step); box: $(rocm-smi --showuniqueid 2>/dev/null | grep -m1 -o '0x[0-9a-f]*')"
  python tools/train_trace_summary.py $KT $MT ) > $OUT/train_trace.txt 2>&1
rm -rf $OUT/prof
cat $OUT/train_trace.txt
