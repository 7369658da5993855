#!/usr/bin/env python3
"""Time line of pipelined steps from the in-kernel stamps of the diagnostic build (make -C nemotron-asr.cpp_amd/csrc stamps).

  NASR_LIB_PATH=nemotron-asr.cpp_amd/libnemotron_asr_amd_stamps.so NASR_STAMPS_OUT=gpurun_out/stamps.txt \
      python bench.py --pipeline-depth 2 --steps 100 --no-cpu-baseline --no-extra-configs --no-buffered --no-profile-pass --no-host-pcm --regions 1
  python tests/micro/stamps_timeline.py gpurun_out/stamps.txt

Each line of the dump: pipeline slot, launch index (8 x layer + k), 8 stamps of the first workgroup, 8 of the last (10 ns ticks of
s_memrealtime: one clock for the whole chip).  A slot's region holds the LAST replay of that slot's graphs, so the file is the
time line of the last few steps: which kernels of which steps were on the chip together, how long a kernel takes next to the
other chain(s), and the gap from the end of a kernel to the start of the next one of its chain."""
import sys
from collections import defaultdict

GAP_TICKS = 15000     # 150 us without a kernel of the slot = the boundary between two pieces (graphs) of that step
NAMES = ["ln+W1", "W2", "ln+QKV", "attn+Wo", "ln+pw1", "dw+pw2", "ln+W1'", "W2'"]


def main(path):
    rows = []
    for ln in open(path):
        v = [int(x) for x in ln.split()]
        ps, k, a, b = v[0], v[1], v[2:10], v[10:18]
        st = [x for x in (a[0], b[0]) if x]
        en = [x for x in a + b if x]
        rows.append(dict(ps=ps, k=k, start=min(st), end=max(en), first_end=max(x for x in a if x), last_start=b[0] or a[0]))
    if not rows:
        print("no stamps"); return
    t0 = min(r["start"] for r in rows)
    by_ps = defaultdict(list)
    for r in rows:
        by_ps[r["ps"]].append(r)
    print("slot: kernels, span of its last replay (us, relative), duration")
    for ps, rs in sorted(by_ps.items()):
        rs.sort(key=lambda r: r["k"])
        # pieces: a jump in time between consecutive launches larger than 50 us marks the boundary between pieces
        pieces, cur = [], [rs[0]]
        for r0, r1 in zip(rs, rs[1:]):
            if r1["start"] - r0["end"] > GAP_TICKS or r1["start"] < r0["start"]:
                pieces.append(cur); cur = []
            cur.append(r1)
        pieces.append(cur)
        for i, pc in enumerate(pieces):
            s, e = pc[0]["start"], pc[-1]["end"]
            dur = [(r["end"] - r["start"]) / 100 for r in pc]
            gaps = [(b["start"] - a["end"]) / 100 for a, b in zip(pc, pc[1:])]
            print(f"  slot {ps} piece {i}: launches {pc[0]['k']:3d}..{pc[-1]['k']:3d}  {(s - t0) / 100:9.1f} .. {(e - t0) / 100:9.1f} us  = {(e - s) / 100:7.1f} us;"
                  f" kernel {sum(dur) / len(dur):5.2f} us avg, gap {sum(gaps) / max(1, len(gaps)):5.2f} us avg")
    # concurrency over the window covered by the most recent two slots' spans
    ev = []
    for r in rows:
        ev.append((r["start"], 1)); ev.append((r["end"], -1))
    ev.sort()
    lo = sorted(r["start"] for r in rows)[len(rows) // 4]
    hi = max(r["end"] for r in rows)
    occ, lvl, prev = defaultdict(int), 0, None
    for t, d in ev:
        if prev is not None and t > lo:
            occ[lvl] += t - max(prev, lo)
        lvl += d; prev = t
    tot = sum(occ.values())
    print("kernels in flight (share of the window from the first quartile of starts to the end):",
          ", ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(occ.items())), f" window {(hi - lo) / 100:.1f} us")
    # distributions
    durs = sorted((r["end"] - r["start"]) / 100 for r in rows)
    gaps = []
    for rs in by_ps.values():
        gaps += [(b["start"] - a["end"]) / 100 for a, b in zip(rs, rs[1:]) if 0 <= b["start"] - a["end"] <= GAP_TICKS]
    gaps.sort()
    pct = lambda v, q: v[min(len(v) - 1, int(q * len(v)))]
    print("kernel duration us: p10 %.2f p50 %.2f p90 %.2f p99 %.2f | gap to the next kernel of the chain us: p10 %.2f p50 %.2f p90 %.2f p99 %.2f"
          % (pct(durs, .1), pct(durs, .5), pct(durs, .9), pct(durs, .99), pct(gaps, .1), pct(gaps, .5), pct(gaps, .9), pct(gaps, .99)))
    # per kernel type
    byk = defaultdict(list)
    for r in rows:
        byk[r["k"] % 8].append((r["end"] - r["start"]) / 100)
    print("duration by kernel of the layer (us):", ", ".join(f"{NAMES[k]} {sum(v) / len(v):.2f}" for k, v in sorted(byk.items())))
    if len(sys.argv) > 2:
        # round 5: the same numbers as JSON (bench.py's `timed_regime` block reads profiles/r5_<tag>_pipelined_trace.json).  rocprofv3 cannot give
        # them: its queue interception serialises the lanes (profiles/r5_b1_R0_pipelined_kernel_stats.md: 4.0 ms per step under the profiler)
        import json
        n_launch = len({(r["ps"], r["k"]) for r in rows})
        per_step = max(r["k"] for r in rows) + 1                     # fused launches of one step (8 x layers)
        avg_us = sum(durs) / len(durs)
        busy = sum(v for k, v in occ.items() if k >= 1)
        clipped = sum(min(r["end"], hi) - max(r["start"], lo) for r in rows if r["end"] > lo)
        ms_per_step = float(sys.argv[3]) if len(sys.argv) > 3 else None
        j = dict(source="in-kernel s_memrealtime stamps of the diagnostic build (make -C nemotron-asr.cpp_amd/csrc stamps; tests/micro/stamps_timeline.py): first / last "
                        "workgroup of every fused-layer launch of the last five steps; front end and decode launches are not stamped",
                 steps_in_file=round(n_launch / per_step, 2), launches_per_step=per_step, ms_per_step=ms_per_step,
                 kernel_ms_per_step=round(per_step * avg_us / 1e3, 4),
                 busy_ms_per_step=round(per_step * avg_us / 1e3 * busy / max(1, clipped), 4),
                 overlap=round(clipped / max(1, busy), 3),          # over the stamped window, which ends with the drain of the pipeline (fewer lanes busy)
                 avg_kernels_in_flight=round(per_step * avg_us / 1e3 / ms_per_step, 3) if ms_per_step else None,      # sum of kernel time per step / ms_per_step: the steady-state figure
                 in_flight_share={str(k): round(v / tot, 4) for k, v in sorted(occ.items())},
                 dominant=dict(kernel="k_fused_skinny<LN> (ln+W1, ln+QKV, ln+pw1, ln+W1': 4 of a layer's 8 launches)",
                               calls_per_step=per_step // 2, avg_us=round(sum(sum(byk[k]) for k in (0, 2, 4, 6)) / max(1, sum(len(byk[k]) for k in (0, 2, 4, 6))), 3)),
                 kernel_us=dict(p10=pct(durs, .1), p50=pct(durs, .5), p90=pct(durs, .9)), gap_us=dict(p10=pct(gaps, .1), p50=pct(gaps, .5), p90=pct(gaps, .9)),
                 by_kernel_us={NAMES[k]: round(sum(v) / len(v), 3) for k, v in sorted(byk.items())})
        json.dump(j, open(sys.argv[2], "w"), indent=1)
        print("json ->", sys.argv[2])


if __name__ == "__main__":
    main(sys.argv[1])
