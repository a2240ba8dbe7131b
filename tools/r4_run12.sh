cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/yard
python tools/hipblaslt_yardstick.py > gpurun_out/yard/yard.txt 2> gpurun_out/yard/yard.err
cat gpurun_out/yard/yard.txt | head -12
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/yard/p -o y --output-format csv -- python $GRAFT_REPO_ROOT/tools/hipblaslt_yardstick.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/yard/p/y_kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/yard/p/y_kernel_stats.csv')))
for r in rows[:24]: print(r['Name'][:150], r['Calls'], r['AverageNs'])
PY
