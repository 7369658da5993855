#!/bin/bash
# usage: tests/prof_pmc.sh <tag> [bench args...]  -- HBM traffic counters, one counter per pass
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_%s/*/*counter_collection.csv"%C)
    if not fs: print(C,"no counter file", glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_%s/*/*"%C)); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name")!=C: continue
        k=r["Kernel_Name"][:60]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:8]:
        print(C, f"{k:60s} launches={n:6d} per_launch={v/n:12.1f}")
PY
