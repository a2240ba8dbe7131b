cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train_step.py -q -x 2>&1 | tail -15
python -m pytest tests/test_gpu_fuzz_train.py tests/test_gpu_model.py -q -k "train" 2>&1 | tail -4
python bench.py --train --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train C step', round(d['value']), d['ms_per_step'])"
MM_TRAIN_PY=1 python bench.py --train --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train python driver', round(d['value']), d['ms_per_step'])"
