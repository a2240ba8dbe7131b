cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pipe
timeout 300 python tools/pipeline_probe.py 20 > gpurun_out/pipe/out.txt 2> gpurun_out/pipe/err.txt
cat gpurun_out/pipe/out.txt; tail -5 gpurun_out/pipe/err.txt
