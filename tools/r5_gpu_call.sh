mkdir -p gpurun_out/r5w
timeout 900 python -m pytest tests/test_gpu_terms_gemm.py tests/test_gpu_train_step.py -q -x -s > gpurun_out/r5w/t1.log 2>&1; echo "pytest 1 rc $?"; grep -E "feed-forward|fold refused|conv 3x3|passed|failed|Error" gpurun_out/r5w/t1.log | tail -12
timeout 900 python -m pytest tests/test_gpu_base_size.py tests/test_gpu_parity_mode.py -q -x -k "vae or vqgan" > gpurun_out/r5w/t2.log 2>&1; echo "pytest 2 rc $?"; tail -2 gpurun_out/r5w/t2.log
