mkdir -p gpurun_out/r2q
timeout 900 python -m pytest tests/test_gpu_base_size.py -q -s -k superres > gpurun_out/r2q/c4.log 2>&1; grep -E "super-res parity\]|passed|failed|rror|assert" gpurun_out/r2q/c4.log | tail -30
