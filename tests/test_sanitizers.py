"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code that parses bytes from outside:
the GGUF reader and the wire-frame / chunk arithmetic of the server (host/fuzz_harness.cpp, built by `make -C host san`) over a
corpus of mutated files, and the CPU oracle (oracle/_san/libnasr_oracle_san.so, `make -C oracle san`) through a streaming session.
Sanitizers run on the CPU build only (GPU AddressSanitizer is not available on the pool).  A malformed input must be REJECTED with a
message -- the assertion is that no input makes a sanitizer speak or the process die."""
import os
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from nemotron_asr_amd import gguf_io, synth

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "nemotron-asr.cpp_amd" / "host"
HARNESS = ROOT / "nemotron-asr.cpp_amd" / "bin" / "host_fuzz_san"
ORACLE_SAN = ROOT / "oracle" / "_san" / "libnasr_oracle_san.so"
SAN_ENV = dict(ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")


def _build(args):
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build failed here: " + r.stderr[-300:])


@pytest.fixture(scope="module")
def harness():
    _build(["make", "-C", str(HOST), "san"])
    return HARNESS


def _run(harness, mode, files):
    out = []
    for i in range(0, len(files), 64):              # a batch per process: a crash names its batch
        r = subprocess.run([str(harness), mode] + [str(f) for f in files[i:i + 64]], capture_output=True, text=True, errors="replace", env=dict(os.environ, **SAN_ENV), timeout=120)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stderr[-2000:], r.stdout[-500:])
        out += r.stdout.splitlines()
    return out


def _base_gguf(path):
    rng = np.random.default_rng(5)
    W = {"encoder.pre_encode.conv.0.weight": rng.standard_normal((8, 1, 3, 3)).astype(np.float32),
         "encoder.layers.0.self_attn.linear_q.weight": (synth.TYPE_Q8_0, synth.pack_q8_0(rng.standard_normal((32, 64)).astype(np.float32)), (32, 64)),
         "encoder.layers.0.feed_forward1.linear1.weight": (synth.TYPE_F16, rng.standard_normal((16, 32)).astype(np.float16), (16, 32)),
         "encoder.layers.0.conv.pointwise_conv1.weight": (synth.TYPE_Q4_0, synth.pack_q4_0(rng.standard_normal((32, 32)).astype(np.float32)), (32, 32)),
         "decoder.prediction.embed.weight": rng.standard_normal((17, 4)).astype(np.float32)}
    gguf_io.write_gguf(path, W, gguf_io.default_hparams(n_layers=1, num_prompts=2), gguf_io.synthetic_vocab(64), prompt_dict={"en": 0, "de": 1})
    return path.read_bytes()


def test_mutated_gguf_files_are_rejected_not_crashed_on(harness, tmp_path):
    """400 mutations of a small GGUF v3 file with every supported tensor type, both vocabulary forms and a prompt dictionary: byte flips
    in the header / KV / tensor-info region, 64-bit count and length fields blown up, offsets moved, truncations at every kind of
    boundary.  The reader opens or rejects each one; ASan / UBSan stay silent; the intact file still parses."""
    base = _base_gguf(tmp_path / "base.gguf")
    rng = np.random.default_rng(11)
    meta_end = base.rfind(b"decoder.prediction.embed.weight") + 80          # tensor infos end shortly behind the last name
    files = [tmp_path / "base.gguf"]
    for i in range(400):
        b = bytearray(base)
        kind = i % 5
        if kind == 0:                                    # random byte flips in the metadata
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, meta_end))] ^= int(rng.integers(1, 256))
        elif kind == 1:                                  # a 64-bit field becomes huge / negative
            o = int(rng.integers(8, meta_end - 8))
            b[o:o + 8] = struct.pack("<q", int(rng.choice([2 ** 62, -1, 2 ** 40, 2 ** 31, len(base) + 1, 0])))
        elif kind == 2:                                  # truncation
            b = b[:int(rng.integers(0, len(base)))]
        elif kind == 3:                                  # a 32-bit field (type ids, n_dims, string lengths' low words)
            o = int(rng.integers(8, meta_end - 4))
            b[o:o + 4] = struct.pack("<I", int(rng.choice([0xffffffff, 0x7fffffff, 13, 255, 5, 0])))
        else:                                            # header counts
            o = int(rng.choice([8, 16]))
            b[o:o + 8] = struct.pack("<q", int(rng.choice([2 ** 33, -5, 10 ** 6, 0, 6, 5])))
        f = tmp_path / f"m{i:03d}.gguf"
        f.write_bytes(bytes(b))
        files.append(f)
    out = _run(harness, "gguf", files)
    assert out[0].startswith("ok: ") and "5 tensors" in out[0], out[0]
    assert len(out) == len(files)
    assert sum(line.startswith("rejected") for line in out) > 100          # the corpus does hit the reader's checks


def test_mutated_wire_frames(harness, tmp_path):
    """Byte streams of the reference's wire protocol (src/server-protocol.h:24-41) as a client might send them -- and as it must not:
    valid sessions, payload lengths beyond the limit or beyond the stream, unknown opcodes, STREAM_START payloads that are not JSON,
    right_context values outside {0, 1, 6, 13}, odd-sized PCM.  The frame decoder and the batch former's chunk arithmetic
    (host/server_protocol.h) take all of it; the harness checks samples_for_chunks / chunks_after against each other on the way."""
    rng = np.random.default_rng(3)

    def frame(op, sid, payload=b"", lie=None):
        return struct.pack("<BII", op, sid, len(payload) if lie is None else lie) + payload

    files = []
    for i in range(200):
        s = b""
        R = int(rng.choice([0, 1, 6, 13]))
        cfg = rng.choice(['{"lang":"auto","right_context":%d}' % R, '{"right_context":%d' % R, "right_context", '{"lang":"en"}', '{"right_context":-7}',
                          '{"lang":"' + "x" * 300 + '","right_context": 99999999999999999999}', ""])
        s += frame(0x01, 0, str(cfg).encode())
        for _ in range(int(rng.integers(0, 30))):
            n = int(rng.choice([1280 * (1 + R), 1, 3, 0, 17920, int(rng.integers(0, 40000))]))
            s += frame(0x02, 1, rng.integers(-3000, 3000, n).astype("<i2").tobytes()[:2 * n - int(rng.integers(0, 2))])
        s += frame(int(rng.choice([0x03, 0x04, 0x7e, 0xff])), 1, b"de" if rng.integers(0, 2) else b"")
        k = i % 4
        if k == 1:
            s += frame(0x02, 1, b"\0" * 10, lie=int(rng.choice([0xffffffff, (256 << 20) + 1, 11, 4000])))
        elif k == 2:
            s = s[:int(rng.integers(0, len(s) + 1))]
        elif k == 3:
            b = bytearray(s)
            for _ in range(4):
                if b:
                    b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
            s = bytes(b)
        f = tmp_path / f"frames{i:03d}.bin"
        f.write_bytes(s)
        files.append(f)
    out = _run(harness, "frames", files)
    assert len(out) == len(files) and all(line.startswith("ok: ") for line in out), [line for line in out if not line.startswith("ok: ")][:3]


def test_json_int_keeps_the_default_on_non_numbers(harness):
    """The reference's json_get_int (src/nemo-server.cpp:172-188) returns false on anything that is not a number and the server keeps its
    --right-context default; round 4's atoi form returned 0 for null / "13" / a bare word and silently overrode it (advisor)."""
    cases = ['{"right_context":13}', '{"right_context": 6 }', '{"right_context":\t1}', '{"right_context":-7}', '{"right_context":null}',
             '{"right_context":"13"}', '{"right_context":abc}', '{"right_context":}', '{"lang":"en"}', '{"right_context": 99999999999999999999}',
             '{"right_context":-99999999999999999999}', '{"right_context":']
    out = subprocess.run([str(harness), "jsonint", "13"] + cases, capture_output=True, text=True, check=True).stdout.split("\n")
    got = [tuple(int(x) for x in line.split()) for line in out if line]
    assert got == [(1, 13), (1, 6), (1, 1), (1, -7), (0, 13), (0, 13), (0, 13), (0, 13), (0, 13), (1, 2147483647), (1, -2147483647), (0, 13)]


def test_oracle_streaming_session_under_asan_and_ubsan(tmp_path):
    """The oracle (the checker of every parity test) itself under ASan + UBSan: a two-layer model, ragged pushes at R = 0 and R = 13,
    the bf16-emulating mode, the decision log, reset in the reference's mode, the tail flush, the
    diarization oracle's VAD window."""
    _build(["make", "-C", str(ROOT / "oracle"), "san"])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not Path(libasan).exists():
        pytest.skip("libasan.so not found")
    script = tmp_path / "run.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {str(ROOT)!r})
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from nemotron_asr_amd import synth
from oracle import binding as ob, diar_binding as db
W = synth.make_weights(n_layers=2)
pcm = synth.make_pcm(5, 2.6)          # two 1.12 s chunks at R = 13, 30 at R = 0: enough for every code path, a third less time under ASan than 4 s
toks = []
for kwargs in ({{}}, {{"emulate_bf16": True}}):
    om = ob.OracleModel(W, 2, **kwargs)
    for R in (0, 13):
        st = ob.OracleStream(om, R)
        st.enable_decision_log()
        t, o = [], 0
        for n in (1, 999, 1280, 17920, 5, 12000):
            t += st.process(pcm[o:o + n]); o += n
        t += st.process(pcm[o:]) + st.finalize()
        toks.append(t)
        assert len(st.decision_log()["margin"]) > 0
        st.reset(reference=True)
        st.process(pcm[:19000])
Wv = synth.make_diar_weights(spk=False)
p = db.DiarModel(Wv).vad_window(pcm[:10080].astype(np.float32) / 32768.0)
import json
print("TOKENS " + json.dumps(dict(tokens=toks, vad=round(p, 5))))
""")
    env = dict(os.environ, **SAN_ENV, LD_PRELOAD=f"{libasan}:{libubsan}", NASR_ORACLE_LIB=str(ORACLE_SAN), NASR_ORACLE_THREADS="4", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stderr[-3000:])
    san = r.stdout.strip().splitlines()[-1]
    env2 = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "NASR_ORACLE_LIB")}
    ref = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env2, timeout=900)
    assert ref.returncode == 0, ref.stderr[-2000:]
    # -O1 without -march=native sums in another order than the optimised build: on the near-tie synthetic checkpoint a decision may
    # flip (tests/test_gpu_configs.py: first divergences at margins < 0.05), so the two builds are compared loosely -- what this test
    # asserts is the sanitizers' silence above
    import json
    a, b = (json.loads(x[len("TOKENS "):]) for x in (san, ref.stdout.strip().splitlines()[-1]))
    assert abs(a["vad"] - b["vad"]) < 1e-4
    for ta, tb in zip(a["tokens"], b["tokens"]):
        assert len(ta) > 3 and abs(len(ta) - len(tb)) <= 3 and ta[:3] == tb[:3], (ta, tb)


def test_batch_former_chunk_arithmetic_equals_the_oracles_stream_manager(harness):
    """The server hands a session whole chunks only, computed from the samples it has been handed (host/server_protocol.h:
    chunks_after / samples_for_chunks).  That arithmetic against the oracle's stream manager (reference src/nemo-stream.cpp:1145-1206:
    chunks processed after n samples), every lookahead, ragged cumulative sample counts incl. the exact boundaries."""
    from oracle import binding as ob
    W = synth.make_weights(n_layers=1)
    om = ob.OracleModel(W, 1)
    rng = np.random.default_rng(2)
    for R in (0, 1, 6, 13):
        T = 1 + R
        cuts = sorted(set([0, 1, 255, 256, 257, 1279, 1280] + [int(x) for x in rng.integers(0, 90000, 40)]
                          + [160 * (9 + 8 * T + k * 8 * T - 1) + 256 + d for k in range(4) for d in (-1, 0, 1)]))
        cuts = [c for c in cuts if c >= 0]
        pcm = synth.make_pcm(3, cuts[-1] / 16000.0 + 0.1)
        ost, prev, want = ob.OracleStream(om, R), 0, []
        for c in cuts:
            ost.process(pcm[prev:c])
            prev = c
            want.append(ost.total_chunks)
        r = subprocess.run([str(harness), "chunks", str(T)] + [str(c) for c in cuts], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=60)
        assert r.returncode == 0, r.stderr[-500:]
        rows = [tuple(int(x) for x in line.split()) for line in r.stdout.splitlines()]
        assert [k for _, k, _ in rows] == want, (R, list(zip(cuts, want, [k for _, k, _ in rows]))[:8])
        for S, k, need in rows:
            assert need <= S and (k == 0 or need > 0)


def _pick_call(harness, T, budget, cap, pending):
    r = subprocess.run([str(harness), "pickcall", str(T), str(budget), str(cap)] + [str(x) for x in pending], capture_output=True, text=True, env=dict(os.environ, **SAN_ENV), timeout=60)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-500:]
    out = [int(x) for x in r.stdout.split()]
    return out[0], out[1:]


def test_batch_former_picks_the_chunk_count_that_carries_the_most_rows(harness):
    """host/server_protocol.h: pick_call -- what one engine call of the server carries when sessions hold several whole chunks (round 6: `--backlog-chunks`, nasr_engine_create_ex).
    The cases that went wrong or could: one late session must not pull a 64-stream call down to one chunk; sessions with a deep backlog must not crowd the others out (the first
    version of the rule picked 16 chunks x 16 streams); a group larger than the launch's row budget is cut; and on random inputs the choice is a power of two within every bound
    that carries at least as many rows as any other admissible power of two."""
    T, budget, cap = 14, 64 * 14 * 4, 64                       # 64 streams x R = 13, --backlog-chunks 4
    G, take = _pick_call(harness, T, budget, cap, [4] * 64)
    assert G == 4 and sum(take) == 64
    G, take = _pick_call(harness, T, budget, cap, [4] * 63 + [1])
    assert G == 4 and sum(take) == 63 and take[-1] == 0          # 63 x 4 rows beat 64 x 1; the late session waits for the next call
    G, take = _pick_call(harness, T, budget, cap, [16] * 64)
    assert G == 4 and sum(take) == 64                            # never more chunks per session than leave room for every session of a full server
    G, take = _pick_call(harness, T, budget, cap, [1] * 64)
    assert G == 1 and sum(take) == 64                            # live streams: one chunk each, everybody
    G, take = _pick_call(harness, T, 64 * 14, 64, [3] * 100)
    assert G == 1 and sum(take) == 64 and take[:64] == [1] * 64  # --backlog-chunks 1: the launch takes 64 sessions, first come first served
    G, take = _pick_call(harness, 1, 3584, 64, [1, 1, 1, 8])
    assert G == 8 and take == [0, 0, 0, 1]                       # R = 0: eight chunks of one session are more rows than one chunk of four
    rng = np.random.default_rng(11)
    for _ in range(120):
        T = int(rng.choice([1, 2, 7, 14]))
        cap = int(rng.integers(1, 96))
        budget = max(cap * 14 * int(rng.integers(1, 9)), 256)
        n = int(rng.integers(1, 120))
        pending = [int(x) for x in rng.integers(1, 20, n)]
        G, take = _pick_call(harness, T, budget, cap, pending)
        gmax = max(1, min(248 // T, budget // (cap * T)))
        assert G >= 1 and G & (G - 1) == 0 and G <= gmax, (G, gmax)
        assert all(p >= G for p, t in zip(pending, take) if t) and sum(take) >= 1
        assert sum(take) * G * T <= budget or sum(take) == 1
        rows = sum(take) * G
        g = 1
        while g <= gmax:
            n_g = min(sum(p >= g for p in pending), budget // (g * T))
            assert rows >= n_g * g, (T, cap, budget, pending, G, g)
            g *= 2
