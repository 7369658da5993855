#!/bin/bash
# usage: tests/prof_r4.sh <tag> [bench args...]      (on the GPU box, through gpurun)
# The profiles of a round in one go, for ONE bench configuration:
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/<tag>_kernel_stats.{csv,md}
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, MI355X_MICROARCH.md)
#                                                  -> gpurun_out/r4_pmc_traffic_<tag>.json (per kernel: HBM bytes per launch,
#                                                     FETCH_SIZE x 2 + WRITE_SIZE, KiB counters -> bytes)
# PROF_MODE="" profiles pipelined steps instead (the profiler serialises the queues: same kernels as the pipelined path, each alone on the chip).
# Runs synchronous steps (one launch chain at a time: the kernels alone on the chip), one timed region, < 60 graph
# steps in all: rocprofv3 (ROCm 7.2) crashes in its queue interceptor after ~15 000 graph-launched kernels in one process.
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="--no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --no-host-pcm --no-b512 --no-f32-engine --regions 1 ${PROF_MODE---sync-steps} --steps ${PROF_STEPS:-40} --warmup 3 $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/prof_$TAG.log 2>&1
echo "stats pass rc=$? segv=$(grep -c SIGSEGV $OUT/prof_$TAG.log) $(grep -o '"ms_per_step": [0-9.]*' $OUT/prof_$TAG.log | head -1)"
for C in ${PROF_COUNTERS-FETCH_SIZE WRITE_SIZE}; do      # PROF_COUNTERS="" skips the PMC passes
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_${TAG}_$C.log 2>&1
  echo "$C pass rc=$? segv=$(grep -c SIGSEGV $OUT/pmc_${TAG}_$C.log)"
done
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"; tag = "$TAG"
fs = glob.glob(f"{out}/prof_{tag}/*/*kernel_stats.csv")
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    with open(f"{out}/{tag}_kernel_stats.csv", "w") as f:
        f.write(open(fs[0]).read())
    with open(f"{out}/{tag}_kernel_stats.md", "w") as f:
        f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:24]:
            f.write(f"| \`{r['Name'][:90]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")
    for r in rows[:14]:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
agg = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/pmc_{tag}_{C}/*/*counter_collection.csv")
    if not fs:
        print(C, "no counter file"); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != C: continue
        a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        agg[k][C + "_per_launch"] = round(v / n, 1); agg[k]["launches_" + C] = n
for k, d in agg.items():
    if "FETCH_SIZE_per_launch" in d and "WRITE_SIZE_per_launch" in d:
        d["hbm_bytes_per_launch_corrected"] = int(2 * d["FETCH_SIZE_per_launch"] * 1024 + d["WRITE_SIZE_per_launch"] * 1024)
json.dump({"command": f"tests/prof_r4.sh {tag} $*  (rocprofv3 --pmc <C> --kernel-trace, one counter per pass; bench.py {'$ARGS'})",
           "units": "FETCH_SIZE/WRITE_SIZE are KiB-granular request counters; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read "
                    "(MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2*FETCH_SIZE*1024, hbm_write_bytes = WRITE_SIZE*1024",
           "kernels": dict(sorted(agg.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1].get("launches_FETCH_SIZE", 0)))},
          open(f"{out}/r4_pmc_traffic_{tag}.json", "w"), indent=1)
for k, d in list(agg.items())[:0]:
    pass
top = sorted(agg.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1].get("launches_FETCH_SIZE", 0))[:8]
for k, d in top:
    print(f"{k[:60]:60s} launches={d.get('launches_FETCH_SIZE')} hbm_bytes/launch={d.get('hbm_bytes_per_launch_corrected')}")
PY
rm -rf $OUT/prof_$TAG $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
