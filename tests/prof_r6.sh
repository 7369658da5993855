#!/bin/bash
# usage: tests/prof_r6.sh <tag> [bench args...]      (on the GPU box, through gpurun)
# Kernel-level evidence for ONE bench configuration in the regime it is TIMED in (pipelined steps unless PROF_MODE=--sync-steps):
#   1. rocprofv3 --kernel-trace --stats   -> gpurun_out/r6_<tag>_kernel_stats.{csv,md}   (per-kernel calls / average / total)
#                                            gpurun_out/r6_<tag>_rocprof_trace.json: from the dispatch time stamps of the trace, over the timed steps:
#                                            sum of kernel durations per step (all lanes), wall time covered by at least one kernel per step
#                                            ("busy"), overlap = sum / busy, time share with 1 / 2 / 3 / 4+ kernels in flight, the dominant kernel
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; MI355X_MICROARCH.md, HBM: bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB)
#                                         -> gpurun_out/r6_pmc_traffic_<tag>.json
# The program goes directly after `--` (no env / bash -c hop: the profiler's library has already initialised the GPU).  <= 45 graph steps
# in all: rocprofv3 (ROCm 7.2) crashes in its queue interceptor after ~15 000 graph-launched kernels in one process.
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
STEPS=${PROF_STEPS:-24}
ARGS="--no-grouped --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --no-host-pcm --no-b512 --no-f32-engine --regions 1 ${PROF_MODE-} --steps $STEPS --warmup 3 $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/prof_$TAG.log 2>&1
echo "stats pass rc=$? segv=$(grep -c SIGSEGV $OUT/prof_$TAG.log) $(grep -o '"ms_per_step": [0-9.]*' $OUT/prof_$TAG.log | head -1)"
for C in ${PROF_COUNTERS-FETCH_SIZE WRITE_SIZE}; do      # PROF_COUNTERS="" skips the PMC passes
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_${TAG}_$C.log 2>&1
  echo "$C pass rc=$? segv=$(grep -c SIGSEGV $OUT/pmc_${TAG}_$C.log)"
done
python3 - <<PY
import csv, glob, json, collections, re
out = "$OUT"; tag = "$TAG"; steps = int('$STEPS')
log = open(f"{out}/prof_{tag}.log", errors="replace").read()
m = re.search(r'"ms_per_step": ([0-9.]+)', log)
ms_per_step = float(m.group(1)) if m else None
fs = glob.glob(f"{out}/prof_{tag}/*/*kernel_stats.csv")
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    open(f"{out}/r6_{tag}_kernel_stats.csv", "w").write(open(fs[0]).read())
    with open(f"{out}/r6_{tag}_kernel_stats.md", "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS   (ms_per_step under the profiler: {ms_per_step})\n\n")
        f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:26]:
            f.write(f"| \`{r['Name'][:90]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")
    for r in rows[:12]:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
# ---- the trace: what runs beside what during the timed steps ----
ft = glob.glob(f"{out}/prof_{tag}/*/*kernel_trace.csv")
if ft:
    ev = []
    for r in csv.DictReader(open(ft[0])):
        try:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
        except (KeyError, ValueError):
            pass
    ev.sort()
    # one k_preemph launch per step: the timed region = the last STEPS of them
    pre = [a for a, b, n in ev if "k_preemph" in n]
    if len(pre) > steps:
        t0 = pre[len(pre) - steps]
        t1 = max(b for a, b, n in ev)
        win = [(a, b, n) for a, b, n in ev if a >= t0]
        tot = sum(b - a for a, b, n in win)
        # sweep: time with c kernels in flight
        pts = sorted([(a, 1) for a, b, n in win] + [(b, -1) for a, b, n in win])
        conc = collections.Counter(); c = 0; last = pts[0][0]
        for t, d in pts:
            if t > last: conc[min(c, 4)] += t - last
            c += d; last = t
        busy = sum(v for k, v in conc.items() if k >= 1)
        per = collections.defaultdict(lambda: [0, 0])
        for a, b, n in win:
            per[n][0] += 1; per[n][1] += b - a
        dom = max(per.items(), key=lambda kv: kv[1][1])
        j = dict(command=f"tests/prof_r6.sh {tag} $*  (rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS)",
                 steps=steps, ms_per_step=ms_per_step, window_ms_per_step=round((t1 - t0) / 1e6 / steps, 4),
                 kernel_ms_per_step=round(tot / 1e6 / steps, 4), busy_ms_per_step=round(busy / 1e6 / steps, 4),
                 overlap=round(tot / busy, 3), idle_share=round(conc[0] / max(1, t1 - t0), 4),
                 in_flight_share={str(k) + ("+" if k == 4 else ""): round(v / max(1, t1 - t0), 4) for k, v in sorted(conc.items())},
                 dominant=dict(kernel=dom[0][:80], calls_per_step=round(dom[1][0] / steps, 2), avg_us=round(dom[1][1] / dom[1][0] / 1e3, 3),
                               ms_per_step=round(dom[1][1] / 1e6 / steps, 4)),
                 kernels_per_step={n[:80]: dict(calls=round(v[0] / steps, 2), avg_us=round(v[1] / v[0] / 1e3, 3), ms=round(v[1] / 1e6 / steps, 4))
                                   for n, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]})
        json.dump(j, open(f"{out}/r6_{tag}_rocprof_trace.json", "w"), indent=1)
        print("trace:", {k: j[k] for k in ("ms_per_step", "window_ms_per_step", "kernel_ms_per_step", "busy_ms_per_step", "overlap", "in_flight_share")})
        print("dominant:", j["dominant"])
    else:
        print("trace: fewer k_preemph launches than steps", len(pre), steps)
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
if agg:
    json.dump({"command": f"tests/prof_r6.sh {tag} $*  (rocprofv3 --pmc <C> --kernel-trace, one counter per pass; bench.py $ARGS)",
               "units": "FETCH_SIZE/WRITE_SIZE are KiB-granular request counters; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read "
                        "(MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2*FETCH_SIZE*1024, hbm_write_bytes = WRITE_SIZE*1024",
               "kernels": dict(sorted(agg.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1].get("launches_FETCH_SIZE", 0)))},
              open(f"{out}/r6_pmc_traffic_{tag}.json", "w"), indent=1)
    top = sorted(agg.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0) * kv[1].get("launches_FETCH_SIZE", 0))[:8]
    for k, d in top:
        print(f"{k[:60]:60s} launches={d.get('launches_FETCH_SIZE')} hbm_bytes/launch={d.get('hbm_bytes_per_launch_corrected')}")
PY
rm -rf $OUT/prof_$TAG $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
