mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_parity_mode.py tests/test_gpu_base_size.py "tests/test_gpu_model.py::test_muse_cascade_base_to_superres" -q -s > gpurun_out/r2k/parity.log 2>&1
grep -E "base-size parity\] (parity|bf16) (generate|LFQ)|passed|failed|Error|rror at|assert|^E " gpurun_out/r2k/parity.log | tail -40
