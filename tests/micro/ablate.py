#!/usr/bin/env python3
"""What each non-GEMM launch class costs a PIPELINED step (engine option "ablate": measurement only, results invalid).

    python tests/micro/ablate.py [B] [R] [masks...]        default 64 13, masks 0 1 2 4 8 16 7 15 31 32

One weight set, one engine per mask, bench.py's own Run / timed_regions (primed, K timed calls).  ms per step with a launch class left out
against the full step = the wall time that class costs in the four-lane regime (its kernels' own durations overlap other lanes' work).
"""
import json
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

bench.ge.load_package()
from nemotron_asr_amd import capi, synth  # noqa: E402

NAMES = {1: "k_post", 2: "attention", 4: "dwconv", 8: "decode iterations", 16: "front end", 32: "encoder GEMMs"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    masks = [int(x) for x in sys.argv[3:]] or [0, 1, 2, 4, 8, 16, 7, 15, 31, 32]
    steps = 100 if B * (1 + R) <= 1792 else 20
    W = synth.make_weights(n_layers=24, margins="speech")
    engW, _ = synth.quantize_weights(W, "q8_0")
    del W
    rows = []
    for m in masks:
        bench.ENGINE_OPTIONS[:] = [f"ablate={m}"]
        run = bench.Run(capi, synth, engW, 24, capi.DTYPE_BF16, B, R, 0, list(range(B)), 1, pipeline=4, audio_s=30.0, speech=True)
        for _ in range(8):
            run.step()
        run.drain()
        reg = bench.timed_regions(run, steps, run.eng.synchronize, lambda x: x, repeats=3, prime=bench.PRIME)
        run.drain()
        ms = 1e3 * statistics.median(reg) / steps
        run.eng.set_option("pipeline", 0)
        for _ in range(3):
            run.step()
        run.eng.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(20):
            run.step()
        run.eng.synchronize()
        sync_ms = 1e3 * (time.perf_counter() - t0) / 20
        left_out = " + ".join(v for k, v in NAMES.items() if m & k) or "(nothing: the full step)"
        rows.append(dict(mask=m, left_out=left_out, pipelined_ms=round(ms, 4), synchronous_ms=round(sync_ms, 4)))
        print(json.dumps(rows[-1]), flush=True)
        run.close()
    base = next((r for r in rows if r["mask"] == 0), None)
    if base:
        for r in rows:
            r["pipelined_saving_ms"] = round(base["pipelined_ms"] - r["pipelined_ms"], 4)
            r["synchronous_saving_ms"] = round(base["synchronous_ms"] - r["synchronous_ms"], 4)
    out = ROOT / "gpurun_out" / f"r5_ablation_b{B}_R{R}.json"
    out.parent.mkdir(exist_ok=True)
    out.write_text(json.dumps(dict(batch=B, right_context=R, steps_per_region=steps, rows=rows), indent=1))
    print("->", out)


if __name__ == "__main__":
    main()
