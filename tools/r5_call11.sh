mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -q -x -s -k "term_product_attention" > gpurun_out/r5k/t1.log 2>&1; echo "pytest rc $?"; grep -n "term-product\|passed\|failed\|^E " gpurun_out/r5k/t1.log | cut -c1-250 | tail -12
bash tools/r5_kstats.sh r05b_f16x2_fp32w --precision f16x2 > gpurun_out/r5k/k1.log 2>&1; head -14 gpurun_out/r05b_f16x2_fp32w_kstats.txt | cut -c1-160
