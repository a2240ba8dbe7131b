mkdir -p gpurun_out/r5l
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "TCC_\(HIT\|MISS\|EA0_RDREQ\|EA0_WRREQ\|REQ\|READ\|BUBBLE\|TAG_STALL\)\|MALL\|DRAM" | sort -u | head -60 > $GRAFT_REPO_ROOT/gpurun_out/r5l/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
head -60 gpurun_out/r5l/counters.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -x -k "term_product or split_rows_f16" > gpurun_out/r5l/t1.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5l/t1.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_ops.py tests/test_gpu_fuzz_train.py -q -x > gpurun_out/r5l/t2.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5l/t2.log | cut -c1-250
