#!/bin/bash
# usage: tests/prof_trace.sh <tag> [bench args...]   (on the GPU box via gpurun)
# rocprofv3 --kernel-trace of a short bench run; prints how many kernels are in flight over time (0 = the chip idles
# between dependent launches) and the per-kernel durations of the steady state.  For looking at pipelined steps.
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass --no-buffered --no-extra-configs --regions 1 --no-host-pcm "$@" > $OUT.log 2>&1
grep -E '^\{' $OUT.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
n = len(rows)
lo, hi = rows[n // 3][0], rows[2 * n // 3][0]          # middle third: steady state of the timed regions
sel = [r for r in rows if lo <= r[0] < hi]
ev = []
for s, e, _ in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    last = t; cur += d
tot = sum(hist.values())
print("kernels in flight (share of wall time):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
dur = collections.defaultdict(list)
for s, e, nme in sel:
    dur[nme[:60]].append(e - s)
busy = sum(e - s for s, e, _ in sel)
print(f"window {tot/1e6:.2f} ms, {len(sel)} kernels, sum of durations {busy/1e6:.2f} ms ({busy/tot:.2f} x wall)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"{k:60s} n={len(v):6d} avg_us={sum(v)/len(v)/1e3:7.2f}")
PY
