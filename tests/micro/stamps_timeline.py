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


if __name__ == "__main__":
    main(sys.argv[1])
