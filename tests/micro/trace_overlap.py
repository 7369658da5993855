"""Overlap structure of a rocprofv3 kernel trace (tests/micro/trace_overlap.sh): concurrency of the GEMM launches of the lanes."""
import collections
import csv
import glob
import json
import sys

import numpy as np

d, out = sys.argv[1], sys.argv[2]
fs = glob.glob(f"{d}/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(fs[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
# window: the last 40 % of the trace that precedes the final 5 % (steady pipelined steps; the bench's cold pass and drain are outside)
t_start = ev[0][0]
span = t_end - t_start
is_gemm = lambda n: "k_gemm_roles" in n or "k_gemm_t64" in n or "k_gemm_tiled2" in n
# find the steady window from the GEMM launches themselves: the last 4000 GEMM launches but the final 400
g = [e for e in ev if is_gemm(e[2])]
g = g[-4400:-400] if len(g) > 6000 else g[len(g) // 3: -len(g) // 10]
w0, w1 = g[0][0], g[-1][1]
inw = [e for e in ev if e[0] >= w0 and e[1] <= w1]
res = {"window_ms": (w1 - w0) / 1e6, "kernels_in_window": len(inw), "gemm_launches": len(g)}
# time-weighted concurrency of GEMM launches
pts = []
for s, e, n, q in g:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
conc = collections.Counter()
cur, last = 0, w0
for t, dlt in pts:
    conc[cur] += t - last
    cur += dlt; last = t
tot = sum(conc.values())
res["gemm_concurrency_time_share"] = {str(k): round(v / tot, 4) for k, v in sorted(conc.items())}
res["gemm_busy_ms"] = sum(e - s for s, e, _, _ in g) / 1e6
byname = collections.defaultdict(list)
for s, e, n, q in inw:
    byname[n.split("(")[0][-60:]].append((e - s) / 1e3)
res["durations_us"] = {k: {"n": len(v), "mean": round(float(np.mean(v)), 2), "p10": round(float(np.percentile(v, 10)), 2), "p90": round(float(np.percentile(v, 90)), 2)}
                       for k, v in sorted(byname.items(), key=lambda kv: -sum(kv[1]))[:10]}
# duration of a GEMM launch against the number of other GEMM launches that overlap it
ov = collections.defaultdict(list)
gs = sorted(g)
for i, (s, e, n, q) in enumerate(gs):
    k = 0
    for j in range(max(0, i - 8), min(len(gs), i + 9)):
        if j != i and gs[j][0] < e and gs[j][1] > s:
            k += 1
    ov[k].append((e - s) / 1e3)
res["gemm_duration_by_overlapping_gemms_us"] = {str(k): {"n": len(v), "mean": round(float(np.mean(v)), 2)} for k, v in sorted(ov.items())}
# start alignment: gap from a GEMM's start to the nearest start of a GEMM on another queue
starts = sorted((s, q) for s, e, n, q in g)
gaps = []
for i, (s, q) in enumerate(starts):
    best = None
    for j in range(max(0, i - 6), min(len(starts), i + 7)):
        if starts[j][1] != q:
            dd = abs(starts[j][0] - s)
            best = dd if best is None or dd < best else best
    if best is not None:
        gaps.append(best / 1e3)
if gaps:
    res["nearest_other_queue_gemm_start_us"] = {"p25": round(float(np.percentile(gaps, 25)), 2), "p50": round(float(np.percentile(gaps, 50)), 2), "p75": round(float(np.percentile(gaps, 75)), 2)}
res["queues"] = sorted({q for _, _, _, q in g})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
