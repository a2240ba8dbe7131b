ROUND=r05 FULL=1 bash tools/collect_profiles.sh > gpurun_out/final_collect.log 2>&1
tail -60 gpurun_out/final_collect.log | cut -c1-260
ls gpurun_out/final/
