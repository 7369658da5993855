"""Bit-identity of GEMM kernel variants: 16 / 64 / 100 / 128 / 256 / 260 / 512 streams x R = 13 (M = 224 ... 7 168) on a 3-layer bf16 engine, a few
steps; prints one digest per setting.  A setting is a space-separated list of engine options ("opt:persistent_gemm=0 opt:gemm_cores=1":
nasr_engine_set_option before the first step) and / or environment variables ("VAR=value"); the first digest is the build's default.
    python gemm_variant_identity.py "opt:gemm_cores=1" "opt:gemm_cores=0" "opt:persistent_gemm=1"
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import __graft_entry__ as ge
    ge.load_package()
    from nemotron_asr_amd import capi, synth
    L, R = 3, 13
    W = synth.make_weights(n_layers=L)
    h = hashlib.sha256()
    n = synth.shift_samples(R)
    pool = [synth.make_pcm(300 + b, 6 * n / 16000 + 0.01)[:6 * n] for b in range(128)]
    # 100 streams pipelined (1 400 rows = 6.25 tiles of 224) and 260 streams synchronous (3 640 rows = 14.2 tiles of 256): ragged last tiles of k_gemm_wide
    for B, pipeline in ((16, 0), (64, 0), (64, 4), (100, 4), (128, 0), (256, 0), (260, 0), (512, 0), (512, 4)):
        eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=B)
        for kv in os.environ.get("NASR_VARIANT_OPTS", "").split():
            k, v = kv.split("=", 1)
            eng.set_option(k, int(v))
        eng.set_option("pipeline", pipeline)
        pcms = [np.roll(pool[b % 128], 977 * (b // 128)) for b in range(B)]
        sts = [eng.stream(R) for _ in range(B)]
        for k in range(6):
            for t in eng.step(sts, [p[k * n:(k + 1) * n] for p in pcms]):
                h.update(np.asarray(t, np.int32).tobytes())
        for t in eng.finalize(sts):
            h.update(np.asarray(t, np.int32).tobytes())
        for s in sts:
            h.update(s.tap(capi.TAP_ENCODER_OUT).tobytes())
            for l in range(L):
                h.update(s.tap(capi.TAP_K_CACHE, l, cap=70 * 1024).tobytes())
                h.update(s.tap(capi.TAP_CONV_CACHE, l, cap=8 * 1024).tobytes())
        eng.close()
    print("DIGEST", h.hexdigest())


if __name__ == "__main__":
    if os.environ.get("NASR_VARIANT_CHILD"):
        child()
        sys.exit(0)
    digests = []
    for setting in [""] + sys.argv[1:]:
        env = dict(os.environ, NASR_VARIANT_CHILD="1")
        env.update(kv.split("=", 1) for kv in setting.split() if not kv.startswith("opt:"))
        env["NASR_VARIANT_OPTS"] = " ".join(kv[4:] for kv in setting.split() if kv.startswith("opt:"))
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        d = [l.split()[1] for l in out.stdout.splitlines() if l.startswith("DIGEST")]
        if not d:
            print(out.stdout[-2000:], out.stderr[-2000:])
            sys.exit(1)
        digests.append(d[0])
        print(f"{setting or 'default':60s} {d[0][:16]}  {'==' if d[0] == digests[0] else '!= DEFAULT'}")
    sys.exit(0 if all(d == digests[0] for d in digests) else 1)
