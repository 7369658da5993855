"""soak test (GPU): a pipelined engine (option pipeline = 4: four lanes) against a synchronous one over thousands of calls with random push
sizes (partial chunks, several chunks at once, ragged groups), random subsets of the streams per call, resets, finalize / collect
in between -- token streams must be identical.  The parity suite covers each of these once; this looks for the rare ordering bug.
usage: python tests/micro/soak_pipeline.py [calls] [seed] [pipeline mode (default 4; 8 = grouped pipeline)] [L<layers> (default 4; 8 with mode 8)]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import capi, synth  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pmode = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rng = np.random.default_rng(seed)
    n_layers = int(sys.argv[4][1:]) if len(sys.argv) > 4 else (8 if pmode == 8 else 4)      # "L8", "L24"
    # SOAK_STREAMS="13x72,0x4": 72 streams of R = 13 and 4 of R = 0 (default: 0, 0, 0, 1, 13).  With ~58 of 72 R = 13 streams per call
    # the row count of a step crosses 768 both ways: deep-ring and co-resident GEMM kernels alternate between the steps in flight.
    # "13x100": multi-chunk pushes (2-4 chunks of ~80 streams) cross 3 584 rows: steps cut into three and into four pieces alternate
    # (round 4: pipe_step waits for the steps in flight before one that is cut differently), 224-row tiles come and go.
    spec = __import__("os").environ.get("SOAK_STREAMS")
    Rs = [0, 0, 0, 1, 13]
    if spec:
        Rs = []
        for part in spec.split(","):
            r, n = part.split("x")
            Rs += [int(r)] * int(n)
    B = len(Rs)
    W = synth.make_weights(n_layers=n_layers)
    engs = []
    for mode in (0, pmode):
        e = capi.Engine(W, n_layers=n_layers, dtype=capi.DTYPE_BF16, max_streams=B)
        e.set_option("pipeline", mode)
        if mode:
            for kv in __import__("os").environ.get("SOAK_OPTS", "").split():      # SOAK_OPTS="large_step_pieces=2 wide_tiles=3": options of the pipelined engine
                k, v = kv.split("=")
                e.set_option(k, int(v))
        engs.append(e)
    streams = [[e.stream(R) for R in Rs] for e in engs]
    pcm = [synth.make_pcm(500 + b, 900.0 if B <= 8 else 150.0) for b in range(B)]
    pos = [0] * B
    toks = [[[] for _ in range(B)] for _ in engs]
    n_tok = 0
    log = []
    for c in range(calls):
        # a group = streams of one right context (the engine batches equal T only)
        R = int(rng.choice(sorted(set(Rs))))
        group = [b for b in range(B) if Rs[b] == R and rng.random() < 0.8]
        if not group:
            continue
        piece = synth.shift_samples(R)
        mode = rng.random()
        feat = int(__import__("os").environ.get("SOAK_FEATURES", "7"))      # bisecting aid: 1 ragged pushes, 2 multi-chunk pushes, 4 finalize / reset / collect
        if mode >= 0.7 and mode < 0.85 and not feat & 1:
            mode = 0.0
        if mode >= 0.85 and not feat & 2:
            mode = 0.0
        if mode < 0.7:
            n = [piece] * len(group)                                  # the steady-state shape: one chunk each (graph replay)
        elif mode < 0.85:
            n = [int(rng.integers(1, 3 * piece)) for _ in group]      # ragged
        else:
            k = int(rng.integers(2, 5))
            n = [k * piece] * len(group)                              # several chunks per push
        chunks = []
        for b, nb in zip(group, n):
            if pos[b] + nb > pcm[b].size:
                pos[b] = 0
            chunks.append(pcm[b][pos[b]:pos[b] + nb])
            pos[b] += nb
        for ei, e in enumerate(engs):
            out = e.step([streams[ei][b] for b in group], chunks)
            for b, o in zip(group, out):
                toks[ei][b] += o
        r = rng.random()
        if not feat & 4:
            r = 1.0
        if r < 0.01:                                                  # finalize one stream, then reset it
            b = int(rng.integers(0, B))
            for ei, e in enumerate(engs):
                toks[ei][b] += e.finalize([streams[ei][b]])[0]
                streams[ei][b].reset()
        elif r < 0.03:                                                # drain the pipeline
            for ei, e in enumerate(engs):
                for Rg in sorted(set(Rs)):                            # one call per right context
                    idx = [b for b in range(B) if Rs[b] == Rg]
                    for b, o in zip(idx, e.collect([streams[ei][b] for b in idx])):
                        toks[ei][b] += o
        log.append((c, R, group, n, round(r, 3)))
        chk = int(__import__("os").environ.get("SOAK_CHECK", "0"))
        if chk and c % chk == chk - 1:                                # localise a divergence: drain both engines, compare so far
            for ei, e in enumerate(engs):
                for Rg in sorted(set(Rs)):
                    idx = [b for b in range(B) if Rs[b] == Rg]
                    for b, o in zip(idx, e.collect([streams[ei][b] for b in idx])):
                        toks[ei][b] += o
            for b in range(B):
                if toks[0][b] != toks[1][b]:
                    print(f"divergence on stream {b} within calls {c - chk + 1}..{c}: sync {toks[0][b][-6:]} pipelined {toks[1][b][-6:]}")
                    for item in log[-chk:]:
                        print("   ", item)
                    sys.exit(1)
        if c % 500 == 499:
            print(f"call {c + 1}: tokens so far {sum(len(t) for t in toks[0])}", flush=True)
    for ei, e in enumerate(engs):
        for Rg in sorted(set(Rs)):
            idx = [b for b in range(B) if Rs[b] == Rg]
            for b, o in zip(idx, e.finalize([streams[ei][b] for b in idx])):
                toks[ei][b] += o
    for b in range(B):
        assert toks[0][b] == toks[1][b], f"stream {b}: pipelined tokens differ from synchronous ones at index " \
            f"{next((i for i, (x, y) in enumerate(zip(toks[0][b], toks[1][b])) if x != y), min(len(toks[0][b]), len(toks[1][b])))}"
        n_tok += len(toks[0][b])
    print(f"OK: {calls} calls, {n_tok} tokens, pipelined == synchronous on all {B} streams")


if __name__ == "__main__":
    main()
