#!/bin/bash
# usage: tests/prof_kernels.sh <tag> [bench args...]   (run on the GPU box via gpurun)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass "$@" > $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log 2>&1
grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/bench_$TAG.log | cut -c1-400
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
PY
