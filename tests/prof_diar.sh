cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_diar -- python3 $GRAFT_REPO_ROOT/tests/micro/diar_bench.py > $GRAFT_REPO_ROOT/gpurun_out/diar_bench.log 2>&1
python3 - <<PY
import csv,glob,os
f=sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_diar/*/*kernel_stats.csv"),key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:18]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
PY
