#!/bin/bash
# usage: tests/prof_counters.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...] -- [bench args...]
# One rocprofv3 --pmc pass per quoted group (counters in their own run, --kernel-trace only: MI355X_MICROARCH.md,
# rocprofv3 PMC slots), then a per-kernel summary (sum over launches and per-launch mean of every counter).
TAG=$1; shift
GROUPS_=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
[ "$1" == "--" ] && shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for G in "${GROUPS_[@]}"; do
  if [ -n "$PROF_SCRIPT" ]; then      # another driver script instead of bench.py (e.g. tests/micro/diar_timeline.py)
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -- python3 $GRAFT_REPO_ROOT/$PROF_SCRIPT "$@" > $OUT/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i ($G) failed"; tail -5 $OUT/pmc_${TAG}_$i.log; }
  else
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass --no-buffered "$@" > $OUT/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i ($G) failed"; tail -5 $OUT/pmc_${TAG}_$i.log; }
  fi
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sorted(glob.glob("$OUT/pmc_${TAG}_[0-9]*")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"][:70]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for k, cs in agg.items():
    n = max(v[0] for v in cs.values())
    rows.append(dict(kernel=k, launches=n, per_launch={c: v[1] / v[0] for c, v in cs.items()}))
rows.sort(key=lambda r: -r["launches"] * max(r["per_launch"].values(), default=0))
json.dump(rows, open("$OUT/pmc_${TAG}_summary.json", "w"), indent=1)
for r in rows[:12]:
    print(f'{r["kernel"]:70s} n={r["launches"]:6d} ' + " ".join(f"{c}={v:.4g}" for c, v in sorted(r["per_launch"].items())))
PY
