mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -x -s > gpurun_out/r5j/t1.log 2>&1; echo "pytest rc $?"; grep -n "term-product\|passed\|failed\|^E " gpurun_out/r5j/t1.log | cut -c1-250 | tail -12
timeout 900 python -m pytest tests/test_gpu_base_size.py -q -x -k "f16x2" > gpurun_out/r5j/t2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5j/t2.log | cut -c1-250
for a in "--precision f16x2" "--precision f16x2 --bf16-round-weights"; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"; done 2>&1 | tee gpurun_out/r5j/tier.log
MM_DEBUG=32768 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision f16x2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32-MFMA attention (debug 32768), fp32w:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" | tee -a gpurun_out/r5j/tier.log
