#!/bin/bash
# session-3 GPU call 5: whole GPU suite on the current tree + bench line
cd /root/repo
O=gpurun_out/s3r5; mkdir -p $O
python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -6 $O/tests.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/s3r5/bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('BENCH', d['value'], d['ms_per_step'], d['parity_tier']['value'], d['parity_tier']['fp32_checkpoint']['value'])
else: print(open('gpurun_out/s3r5/bench.log').read()[-800:])
PY
