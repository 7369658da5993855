"""Host side above the C ABI: the ggml-free GGUF v3 reader and the CLI.
CPU tests cover the file format; the GPU test runs the CLI end to end."""
import json
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from nemotron_asr_amd import gguf_io, synth

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "nemotron-asr.cpp_amd" / "bin"


def _fnv(data: bytes) -> str:
    h = 0
    for b in data:
        h = (h * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def _ensure_built():
    if not (BIN / "gguf_dump").exists():
        subprocess.check_call(["make", "-C", str(ROOT / "nemotron-asr.cpp_amd" / "host"), "../bin/gguf_dump"])


def _small_file(tmp_path, legacy=False):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((4, 64)).astype(np.float32)
    w = {
        "encoder.layers.0.conv.depthwise_conv.weight": rng.standard_normal((9, 32)).astype(np.float32),
        "t.f32": a,
        "t.f16": (gguf_io.GGML_F16, a.astype(np.float16), a.shape),
        "t.q8": (gguf_io.GGML_Q8_0, synth.pack_q8_0(a), a.shape),
        "t.q4": (gguf_io.GGML_Q4_0, synth.pack_q4_0(a), a.shape),
        "t.vec": np.arange(7, dtype=np.float32),
    }
    path = tmp_path / "small.gguf"
    gguf_io.write_gguf(path, w, gguf_io.default_hparams(n_layers=3, num_prompts=2), gguf_io.synthetic_vocab(),
                       prompt_dict={"en-US": 0, "auto": 1}, legacy_vocab_blob=legacy)
    return path, w


def test_cpp_reader_matches_python_writer(tmp_path):
    _ensure_built()
    path, w = _small_file(tmp_path)
    out = json.loads(subprocess.check_output([str(BIN / "gguf_dump"), str(path)]))
    kv, tensors, start = gguf_io.read_gguf(path)
    assert out["version"] == 3 and out["data_start"] == start and start % 32 == 0
    assert out["kv"]["nemo.n_layers"] == 3 and out["kv"]["nemo.num_prompts"] == 2 and out["kv"]["nemo.d_model"] == 1024
    assert out["vocab_list"] == 1024 == len(kv["tokenizer.vocab_list"])
    raw = path.read_bytes()
    assert [t["name"] for t in out["tensors"]] == list(w)
    for t in out["tensors"]:
        ty, dims, off, nb = tensors[t["name"]]
        assert (t["type"], t["ne"][:t["n_dims"]], t["offset"], t["nbytes"]) == (ty, dims, off, nb)
        assert off % 32 == 0
        assert t["hash"] == _fnv(raw[start + off:start + off + nb])
    dw = next(t for t in out["tensors"] if "depthwise" in t["name"])
    assert dw["ne"][:2] == [32, 9]          # (k, C) stored -> ne[1] = kernel size (src/nemo-ggml.cpp:357-360)
    q8 = next(t for t in out["tensors"] if t["name"] == "t.q8")
    assert q8["nbytes"] == 4 * 64 // 32 * 34


def test_cpp_reader_rejects_bad_files(tmp_path):
    _ensure_built()
    path, _ = _small_file(tmp_path)
    raw = path.read_bytes()
    (tmp_path / "trunc.gguf").write_bytes(raw[:len(raw) - 100])
    (tmp_path / "magic.gguf").write_bytes(b"GGML" + raw[4:])
    for name in ("trunc.gguf", "magic.gguf", "missing.gguf"):
        r = subprocess.run([str(BIN / "gguf_dump"), str(tmp_path / name)], capture_output=True)
        assert r.returncode == 2 and b"error" in r.stderr


def test_cache_config_arithmetic_in_header():
    """nemo_cache_config mirrors reference src/nemo-stream.h:65-100."""
    hdr = (ROOT / "nemotron-asr.cpp_amd" / "host" / "nemo_amd.h").read_text()
    for name in ("nemo_stream_init", "nemo_stream_process_incremental", "nemo_stream_finalize", "nemo_stream_get_transcript",
                 "nemo_stream_get_tokens", "nemo_stream_reset", "nemo_stream_free", "nemo_stream_set_language", "tokens_to_text"):
        assert name in hdr


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    """GGUF file -> C++ loader -> engine -> transcript; tokens equal the oracle's, text follows the U+2581 rule."""
    from oracle import binding as ob
    n_layers = 2
    W = synth.make_weights(n_layers=n_layers)
    engW, deqW = synth.quantize_weights(W, "q8_0")
    vocab = gguf_io.synthetic_vocab()
    model = tmp_path / "model-q8.gguf"
    gguf_io.write_gguf(model, engW, gguf_io.default_hparams(n_layers=n_layers), vocab, legacy_vocab_blob=True)
    pcm = synth.make_pcm(2, 5.0)
    audio = tmp_path / "a.pcm"
    pcm.tofile(audio)
    cli = BIN / "nemotron-asr-amd"
    assert cli.exists(), "run __graft_entry__.build()"
    for R in (0, 13):
        r = subprocess.run([str(cli), str(model), str(audio), "80", str(R), "--f32", "--print-tokens"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        toks = [int(x) for x in lines[-1].split()[1:]]
        ost = ob.OracleStream(ob.OracleModel(deqW, n_layers), R)
        ref = []
        n = synth.chunk_mel_frames(R) * 160            # the CLI reads get_chunk_samples() per fread
        for o in range(0, pcm.size, n):
            ref += ost.process(pcm[o:o + n])
        ref += ost.finalize()
        assert toks == ref and len(ref) > 0
        text = "".join((" " + vocab[t][1:]) if vocab[t].startswith("▁") else vocab[t] for t in toks)
        assert lines[0] == text
        assert "Real-time factor" in r.stderr
    # file mode: 64 chunks per read share one launch sequence; same tokens; word timestamps = frame * 80 ms
    r = subprocess.run([str(cli), str(model), str(audio), "80", "0", "--f32", "--print-tokens", "--read-chunks", "64", "--timestamps"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    ost = ob.OracleStream(ob.OracleModel(deqW, n_layers), 0)
    ref = ost.process(pcm) + ost.finalize()
    assert [int(x) for x in lines[-1].split()[1:]] == ref
    stamped = "".join((" {%.2f}" % (np.float32(f) * 1280 / 16000) + vocab[t][1:]) if vocab[t].startswith("▁") else vocab[t]
                      for t, f in zip(ref, ost.token_frames()))
    assert lines[-2] == stamped
    # pipelined steps: same transcript and tokens (each delta is printed one read later, the rest by finalize)
    for flag in ("--pipeline", "--pipeline3"):
        r = subprocess.run([str(cli), str(model), str(audio), "80", "0", "--f32", "--print-tokens", flag], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert [int(x) for x in r.stdout.splitlines()[-1].split()[1:]] == ref, flag
    # the fastest way through a file: many chunks per read AND pipelined steps (bench.py buffered_audio.pipelined_value)
    r = subprocess.run([str(cli), str(model), str(audio), "80", "0", "--f32", "--print-tokens", "--read-chunks", "8", "--pipeline", "4"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert [int(x) for x in r.stdout.splitlines()[-1].split()[1:]] == ref
    r = subprocess.run([str(cli), str(model), str(audio), "80", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "right_context" in r.stderr
    # the default of nemo_init (bf16 engine, no --f32): against the bf16-emulating oracle -- same leading tokens, high aligned
    # agreement (greedy decoding leaves the oracle's path at the first near-tie; exactness is asserted on f32 above)
    import difflib
    omb = ob.OracleModel(deqW, n_layers, emulate_bf16=True)
    for R in (0, 13):
        r = subprocess.run([str(cli), str(model), str(audio), "80", str(R), "--print-tokens"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        toks = [int(x) for x in r.stdout.splitlines()[-1].split()[1:]]
        ost = ob.OracleStream(omb, R)
        refb = []
        n = synth.chunk_mel_frames(R) * 160
        for o in range(0, pcm.size, n):
            refb += ost.process(pcm[o:o + n])
        refb += ost.finalize()
        m = min(4, len(refb))
        assert toks[:m] == refb[:m] and len(refb) > 0
        assert difflib.SequenceMatcher(None, toks, refb, autojunk=False).ratio() > 0.7, (toks, refb)


def _send(sock, op, sid, payload=b""):
    sock.sendall(struct.pack("<BII", op, sid, len(payload)) + payload)


def _recv(sock):
    def full(n):
        buf = b""
        while len(buf) < n:
            part = sock.recv(n - len(buf))
            if not part:
                raise ConnectionError("server closed the connection")
            buf += part
        return buf
    op, sid, n = struct.unpack("<BII", full(9))
    return op, sid, full(n)


@pytest.mark.gpu
def test_cli_runs_the_reference_invocations_unchanged(tmp_path):
    """A script written for the reference binary keeps working: the invocations of the reference's README.md:14-21 and usage
    text (src/transcribe_stream.cpp:55-63) -- file and stdin input, chunk_ms 70 / 80, the --cpu / --cuda / --metal backend
    selectors (accepted, one stderr note each, no effect) -- give the same transcript; --pipeline N (0..4) too."""
    n_layers = 2
    W = synth.make_weights(n_layers=n_layers)
    model = tmp_path / "model.gguf"
    gguf_io.write_gguf(model, W, gguf_io.default_hparams(n_layers=n_layers), gguf_io.synthetic_vocab())
    pcm = synth.make_pcm(9, 4.0)
    audio = tmp_path / "raw-audio.pcm"
    pcm.tofile(audio)
    cli = str(BIN / "nemotron-asr-amd")

    def run(args, stdin=None):
        r = subprocess.run([cli, str(model)] + args, input=stdin, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        return r.stdout.decode(), r.stderr.decode()

    base, _ = run([str(audio), "70", "13", "--f32"])                      # README: ./nemotron-asr.cpp model.gguf raw-audio.pcm 70 13
    assert len(base.strip()) > 0
    out, _ = run(["-", "70", "13", "--f32"], stdin=pcm.tobytes())         # README: ffmpeg ... - | ./nemotron-asr.cpp model.gguf - 70 13
    assert out == base
    ref80, _ = run([str(audio), "80", "0", "--f32"])                      # usage: model.gguf audio.pcm 80 0
    for flag in ("--cuda", "--cpu", "--metal"):                           # usage: model.gguf audio.pcm 80 0 --cuda
        out, err = run([str(audio), "80", "0", flag, "--f32"])
        assert out == ref80 and f"{flag} ignored" in err
    for depth in ("0", "1", "2", "3", "4"):
        out, _ = run([str(audio), "80", "0", "--f32", "--pipeline", depth])
        assert out == ref80, depth
    out, _ = run([str(audio), "80", "0", "--pipeline", "--f32"])          # round-2 spelling: a bare --pipeline is depth 1
    assert out == ref80
    r = subprocess.run([cli, str(model), str(audio), "80", "0", "--vulkan"], capture_output=True)
    assert r.returncode == 1 and b"Unknown flag" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_flag,pipeline", [("--f32", 0), (None, 0), ("--f32", 3)])
def test_server_batches_streams(tmp_path, dtype_flag, pipeline):
    """Wire protocol of the reference server (src/server-protocol.h:24-41) served by the batch-forming worker:
    3 streams on 2 connections (two right_context groups), text == oracle tokens -> text, errors are framed.
    pipeline = 2: the worker's engine calls overlap on the GPU (text arrives calls later or when the FIFO runs empty):
    the concatenated text of every stream is still the oracle's."""
    import socket
    import time
    from oracle import binding as ob
    n_layers = 2
    W = synth.make_weights(n_layers=n_layers)
    vocab = gguf_io.synthetic_vocab()
    model = tmp_path / "model.gguf"
    gguf_io.write_gguf(model, W, gguf_io.default_hparams(n_layers=n_layers), vocab)
    srv = BIN / "nemo-server-amd"
    assert srv.exists(), "run __graft_entry__.build()"
    path = str(tmp_path / "asr.sock")
    # two lanes (engine + FIFO + worker each) on the one GPU of the test box: stream s is served by lane s mod 2
    proc = subprocess.Popen([str(srv), str(model), "--unix", path] + ([dtype_flag] if dtype_flag else []) + ["--max-streams", "8", "--devices", "0,0", "--cuda"]      # --cuda / --cpu: the reference's backend flags (src/nemo-server.cpp:405-406), accepted and ignored
                            + (["--pipeline", str(pipeline)] if pipeline else []), stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(600):
            if Path(path).exists() or proc.poll() is not None:
                break
            time.sleep(0.1)
        assert proc.poll() is None, proc.stderr.read()
        conns = [socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) for _ in range(2)]
        for c in conns:
            c.settimeout(60)
            c.connect(path)
        plan = [(conns[0], 0, 3), (conns[0], 13, 4), (conns[1], 0, 5)]      # (connection, right_context, pcm seed)
        ids, pcms, text = [], [], {}
        for c, R, seed in plan:
            _send(c, 0x01, 0, json.dumps({"lang": "auto", "right_context": R}).encode())
            op, sid, payload = _recv(c)
            assert op == 0x81 and json.loads(payload)["id"] == sid and sid not in ids
            ids.append(sid)
            pcms.append(synth.make_pcm(seed, 4.0))
            text[sid] = ""
        # interleave pushes of uneven sizes over all streams, then end them
        step = [2000, 5120, 7777]
        off = [0, 0, 0]
        while any(off[i] < pcms[i].size for i in range(3)):
            for i, (c, _, _) in enumerate(plan):
                if off[i] < pcms[i].size:
                    _send(c, 0x02, ids[i], pcms[i][off[i]:off[i] + step[i]].tobytes())
                    off[i] += step[i]
        for i, (c, _, _) in enumerate(plan):
            _send(c, 0x03, ids[i])
        ended = set()
        for c in conns:
            want = {ids[i] for i in range(3) if plan[i][0] is c}
            while not want <= ended:
                op, sid, payload = _recv(c)
                assert op in (0x82, 0x83, 0x84), (hex(op), payload)
                if op == 0x82:
                    assert "queued_samples" in json.loads(payload)
                else:
                    text[sid] += payload.decode()
                    if op == 0x84:
                        ended.add(sid)
        for i, (_, R, _) in enumerate(plan):
            ost = ob.OracleStream(ob.OracleModel(W, n_layers, emulate_bf16=dtype_flag is None), R)
            ref = ost.process(pcms[i]) + ost.finalize()
            want = "".join((" " + vocab[t][1:]) if vocab[t].startswith("▁") else vocab[t] for t in ref)
            if dtype_flag:
                assert len(ref) > 0 and text[ids[i]] == want, (i, R)
            else:       # the server's default engine (bf16): leading words and aligned agreement against the bf16 oracle
                import difflib
                gw, ww = text[ids[i]].split(), want.split()
                assert len(ref) > 0 and gw[:2] == ww[:2], (i, R, gw[:6], ww[:6])
                assert difflib.SequenceMatcher(None, gw, ww, autojunk=False).ratio() > 0.6, (i, R)
        # protocol errors come back as ERROR frames and the connection stays usable
        _send(conns[1], 0x7E, 0)
        op, _, payload = _recv(conns[1])
        assert op == 0x8F and b"opcode" in payload
        _send(conns[1], 0x01, 0, b'{"right_context": 5}')            # not one of the reference's latency modes (src/nemo-stream.h:15-20)
        op, _, payload = _recv(conns[1])
        assert op == 0x8F and b"right_context" in payload
        _send(conns[1], 0x01, 0, b"{}")
        op, sid, _ = _recv(conns[1])
        assert op == 0x81
        _send(conns[1], 0x04, sid, b"xx-XX")
        op, _, payload = _recv(conns[1])
        assert op == 0x8F and b"language" in payload
        for c in conns:
            c.close()
    except BaseException:
        proc.terminate()
        print("server stderr:\n" + proc.stderr.read())
        raise
    finally:
        proc.terminate()
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()
    err = proc.stderr.read()
    assert "streams per call" in err and "engine counters" in err and "--cuda has no effect" in err


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 4])
def test_server_language_switch_mid_stream_on_a_multilingual_model(tmp_path, pipeline):
    """SET_LANG through the socket (reference src/nemo-server.cpp: the audio queued before the switch uses the old language).  A multilingual GGUF
    (4 prompts), f32 engine, two streams started with different languages; stream 0 switches twice mid-stream at sample positions that fall INSIDE a
    chunk (the batch former hands over the sub-chunk remainder before the switch), stream 1 never.  Text == the oracle driven the same way
    (process what was sent, set_prompt, go on); LANG_SET frames carry the prompt index; both with synchronous and pipelined engine calls."""
    import socket
    import time
    from oracle import binding as ob
    n_layers, P, R = 2, 4, 1
    W = synth.make_weights(n_layers=n_layers, num_prompts=P)
    vocab = gguf_io.synthetic_vocab()
    langs = {"en": 0, "de": 1, "fr": 2, "auto": 3}
    model = tmp_path / "multi.gguf"
    gguf_io.write_gguf(model, W, gguf_io.default_hparams(n_layers=n_layers, num_prompts=P), vocab, prompt_dict=langs)
    path = str(tmp_path / "asr.sock")
    proc = subprocess.Popen([str(BIN / "nemo-server-amd"), str(model), "--unix", path, "--f32", "--max-streams", "4", "--right-context", str(R)]
                            + (["--pipeline", str(pipeline)] if pipeline else []), stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(600):
            if Path(path).exists() or proc.poll() is not None:
                break
            time.sleep(0.1)
        assert proc.poll() is None, proc.stderr.read()
        c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        c.settimeout(60)
        c.connect(path)
        start = ["en", "de"]
        ids, text, lang_set = [], {}, []
        for lg in start:
            _send(c, 0x01, 0, json.dumps({"lang": lg}).encode())
            op, sid, _ = _recv(c)
            assert op == 0x81
            ids.append(sid)
            text[sid] = ""
        pcms = [synth.make_pcm(90 + b, 5.0) for b in range(2)]
        piece, off = 3000, [0, 0]                       # 3 000-sample pushes: not a multiple of the 2 560-sample chunk shift
        switches = {7: "fr", 15: "en"}                  # before push k of stream 0
        plan = []                                       # what stream 0 saw: (samples before the switch, language)
        k = 0
        while any(off[b] < pcms[b].size for b in range(2)):
            if k in switches:
                _send(c, 0x04, ids[0], switches[k].encode())
                plan.append((off[0], switches[k]))
            for b in range(2):
                if off[b] < pcms[b].size:
                    _send(c, 0x02, ids[b], pcms[b][off[b]:off[b] + piece].tobytes())
                    off[b] += piece
            k += 1
        for b in range(2):
            _send(c, 0x03, ids[b])
        ended = set()
        while len(ended) < 2:
            op, sid, payload = _recv(c)
            assert op in (0x82, 0x83, 0x84, 0x85), (hex(op), payload)
            if op == 0x85:
                lang_set.append(json.loads(payload))
            elif op in (0x83, 0x84):
                text[sid] += payload.decode()
                if op == 0x84:
                    ended.add(sid)
        c.close()
        assert [(m["id"], m["lang"], m["index"]) for m in lang_set] == [(ids[0], "fr", 2), (ids[0], "en", 0)]
        om = ob.OracleModel(W, n_layers, num_prompts=P)
        for b in range(2):
            ost = ob.OracleStream(om, R, langs[start[b]])
            ref, prev = [], 0
            for cut, lg in (plan if b == 0 else []):
                ref += ost.process(pcms[b][prev:cut])
                ost.set_prompt(langs[lg])
                prev = cut
            ref += ost.process(pcms[b][prev:]) + ost.finalize()
            want = "".join((" " + vocab[t][1:]) if vocab[t].startswith("\u2581") else vocab[t] for t in ref)
            assert len(ref) > 3 and text[ids[b]] == want, (b, text[ids[b]][:80], want[:80])
    except BaseException:
        proc.terminate()
        print("server stderr:\n" + proc.stderr.read())
        raise
    finally:
        proc.terminate()
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()


@pytest.mark.gpu
def test_diarize_cli_end_to_end(tmp_path):
    """diarize.gguf ("vad.*" + "spk.*", layouts of scripts/convert_diarize_to_gguf.py) -> C++ loader -> side-car engine:
    window probabilities through the onset/offset rule, sub-segment embeddings == the oracle's."""
    from oracle import diar_binding as db
    W = synth.make_diar_weights()
    model = tmp_path / "diarize.gguf"
    gguf_io.write_gguf(model, W, {}, [], name="nemo-diarize-synthetic")
    pcm = synth.make_pcm(4, 3.2)
    audio = tmp_path / "d.pcm"
    pcm.tofile(audio)
    a = pcm.astype(np.float32) / 32768.0
    om = db.DiarModel(W)
    probs = om.vad_batch(a)
    onset = float(np.median(probs))                     # a threshold that splits this (random-weight) probability track
    cli = BIN / "diarize-amd"
    assert cli.exists(), "run __graft_entry__.build()"
    r = subprocess.run([str(cli), str(model), str(audio), "--f32", "--onset", repr(onset), "--offset", repr(onset), "--sub-shift", "0.75"],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == f"WINDOWS {probs.size}"
    segs = [tuple(float(x) for x in ln.split()[1:]) for ln in lines if ln.startswith("SEGMENT")]
    want, start = [], None                              # src/diarize_vad.cpp:507-563 with the default post-processing
    for t, p in enumerate(probs):
        if start is None and p >= onset:
            start = t
        elif start is not None and p < onset:
            want.append((start * 0.01, t * 0.01)); start = None
    if start is not None:
        want.append((start * 0.01, probs.size * 0.01))
    # a window whose probability sits within float noise of the threshold may flip: allow it, compare the rest
    assert abs(len(segs) - len(want)) <= 2 and len(want) >= 1
    if len(segs) == len(want):
        assert np.abs(np.array(segs) - np.array(want)).max() < 0.011
    emb = [[float(x) for x in ln.split()[2:]] for ln in lines if ln.startswith("EMBED")]
    starts = list(range(0, a.size - 24000 + 1, 12000))
    assert len(emb) == len(starts) and len(emb[0]) == 192
    ref = np.stack([om.spk_embed(a[s:s + 24000]) for s in starts])
    assert np.abs(np.array(emb) - ref).max() < 2e-3 * np.abs(ref).max()


def _pipeline_restatement(probs, total, onset, offset, min_off=60, shift=12000, window=24000, min_seg=8000):
    """The reference's try_advance / finalize_open_segment (src/diarize_pipeline.cpp:198-263, :341-363) fed one sample at a
    time: when VAD frame f is evaluated the audio reaches the end of its window, f * 160 + 10080."""
    segs, subs = [], []
    in_speech, off_run, start_f, k, seg_id, next_id = False, 0, -1, 0, -1, 0

    def tail(seg_end, at_eof):
        nonlocal k
        seg_start = start_f * 160
        covered = seg_start + ((k - 1) * shift + window if k > 0 else 0)
        left = seg_end - covered
        if left >= min_seg and (k > 0 or at_eof):
            subs.append((seg_id, covered, min(left, 24000))); k += 1
        elif k == 0 and seg_end - seg_start >= min_seg:
            subs.append((seg_id, seg_start, min(seg_end - seg_start, 24000))); k += 1

    for f, p in enumerate(probs):
        if not in_speech:
            if p >= onset:
                in_speech, seg_id, start_f, k, off_run = True, next_id, f, 0, 0
                next_id += 1
        elif p < offset:
            off_run += 1
            if off_run >= min_off:
                end_f = max(f + 1 - off_run, start_f)
                tail(end_f * 160, False)
                segs.append((start_f, end_f))
                in_speech, off_run = False, 0
        else:
            off_run = 0
        if in_speech:
            while start_f * 160 + k * shift + window <= f * 160 + 10080:
                subs.append((seg_id, start_f * 160 + k * shift, window)); k += 1
    if in_speech:
        tail(min(len(probs) * 160, total), True)
        segs.append((start_f, len(probs)))
    return segs, subs


@pytest.mark.gpu
def test_diarize_pipeline_segments_subsegments_clusters_rttm(tmp_path):
    """The whole side-car pipeline (host/diarize_pipeline_amd.cpp) through `diarize-amd --rttm`: batched VAD -> onset / offset
    state machine -> 1.5 s sub-segments -> batched embeddings -> NME-SC -> RTTM.  Segments and sub-segments equal a
    restatement of the reference's frame-by-frame loop; the result does not depend on the push size; the RTTM spans are
    the merged speaker timeline of the labelled sub-segments."""
    from nemotron_asr_amd import capi
    W = synth.make_diar_weights()
    model = tmp_path / "diarize.gguf"
    gguf_io.write_gguf(model, W, {}, [], name="nemo-diarize-synthetic")
    sr = 16000
    parts = [(0.9, None), (3.4, 5), (1.3, None), (0.8, 6), (1.0, None), (4.1, 7), (0.9, None), (0.3, 5), (1.2, None), (2.2, 6)]
    pcm = np.concatenate([np.zeros(int(d * sr), np.int16) if s is None else synth.make_pcm(s, d + 0.01)[:int(d * sr)] for d, s in parts])
    audio = tmp_path / "d.pcm"
    pcm.tofile(audio)
    eng = capi.Diar(W, dtype=capi.DTYPE_F32)
    probs = eng.vad([pcm.astype(np.float32) / 32768.0])[0]
    eng.close()
    lo, hi = np.percentile(probs, 30), np.percentile(probs, 70)
    onset, offset = float(lo + 0.6 * (hi - lo)), float(lo + 0.4 * (hi - lo))
    want_segs, want_subs = _pipeline_restatement(probs, pcm.size, np.float32(onset), np.float32(offset))
    assert len(want_subs) >= 4, "test audio should produce several sub-segments"
    cli = BIN / "diarize-amd"
    outs = []
    for extra in ([], ["--push-ms", "89"], ["--push-ms", "1000"]):
        rttm = tmp_path / f"o{len(outs)}.rttm"
        r = subprocess.run([str(cli), str(model), str(audio), "--f32", "--onset", repr(onset), "--offset", repr(offset), "--rttm", str(rttm),
                            "--num-speakers", "2"] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append((r.stdout, rttm.read_text()))
    assert outs[0] == outs[1] == outs[2]                                  # push-size independence
    lines = outs[0][0].splitlines()
    assert lines[0] == f"WINDOWS {probs.size}"
    segs = [tuple(float(x) for x in ln.split()[1:3]) for ln in lines if ln.startswith("SEGMENT")]
    assert len(segs) == len(want_segs)
    assert np.abs(np.array(segs) - np.array(want_segs) * 0.01).max() < 0.006
    subs = [ln.split() for ln in lines if ln.startswith("SUBSEG")]
    got_subs = [(int(s[4]), float(s[1]), float(s[2])) for s in subs]
    assert len(got_subs) == len(want_subs)
    for (gid, gs, ge), (wid, ws, wl) in zip(got_subs, want_subs):
        assert gid == wid and abs(gs - ws / sr) < 1e-3 and abs(ge - (ws + wl) / sr) < 1e-3
    labels = [int(s[6]) for s in subs]
    assert set(labels) <= {0, 1} and len(set(labels)) == 2               # --num-speakers 2
    # RTTM = the speaker timeline (src/diarize_pipeline.cpp:371-420): same speaker touching -> one span, else cut at the midpoint
    spans = []
    for (_, s, e), lab in sorted(zip(got_subs, labels), key=lambda t: t[0][1]):
        if spans and spans[-1][2] == lab and s <= spans[-1][1] + 1e-3:
            spans[-1][1] = max(spans[-1][1], e)
            continue
        if spans and s < spans[-1][1]:
            mid = 0.5 * (s + spans[-1][1])
            spans[-1][1] = mid
            s = mid
        spans.append([s, e, lab])
    rt = [ln.split() for ln in outs[0][1].splitlines()]
    assert len(rt) == len(spans) and all(r[0] == "SPEAKER" and r[1] == "session" and r[2] == "1" for r in rt)
    for r, (s, e, lab) in zip(rt, spans):
        assert abs(float(r[3]) - s) < 2e-3 and abs(float(r[4]) - (e - s)) < 2e-3 and r[7] == f"spk_{lab}"


@pytest.mark.gpu
def test_asr_cli_with_diarization(tmp_path):
    """`nemotron-asr-amd ... --diarize diarize.gguf --rttm F --json F` (reference src/transcribe_stream.cpp:146-170, :243-290):
    the transcript is unchanged, every word is tagged with a speaker by its emission time, RTTM and per-word JSON are written."""
    from nemotron_asr_amd import capi
    n_layers = 2
    W = synth.make_weights(n_layers=n_layers)
    vocab = gguf_io.synthetic_vocab()
    model = tmp_path / "model.gguf"
    gguf_io.write_gguf(model, W, gguf_io.default_hparams(n_layers=n_layers), vocab)
    dW = synth.make_diar_weights()
    dmodel = tmp_path / "diarize.gguf"
    gguf_io.write_gguf(dmodel, dW, {}, [], name="nemo-diarize-synthetic")
    pcm = np.concatenate([synth.make_pcm(2, 6.0), np.zeros(16000, np.int16), synth.make_pcm(3, 5.0)])
    audio = tmp_path / "a.pcm"
    pcm.tofile(audio)
    eng = capi.Diar(dW, dtype=capi.DTYPE_F32)
    probs = eng.vad([pcm.astype(np.float32) / 32768.0])[0]
    eng.close()
    onset = float(np.percentile(probs, 40))
    cli = BIN / "nemotron-asr-amd"
    plain = subprocess.run([str(cli), str(model), str(audio), "80", "0", "--f32"], capture_output=True, text=True, timeout=180)
    assert plain.returncode == 0, plain.stderr
    rttm, js = tmp_path / "o.rttm", tmp_path / "o.jsonl"
    r = subprocess.run([str(cli), str(model), str(audio), "80", "0", "--f32", "--diarize", str(dmodel), "--rttm", str(rttm), "--json", str(js),
                        "--num-speakers", "2", "--vad-onset", repr(onset), "--vad-offset", repr(onset)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == plain.stdout.splitlines()[0] and len(lines[0].split()) > 3        # same transcript
    tagged = [ln for ln in lines[1:] if ln.startswith("[spk_")]
    assert tagged and "Speaker-tagged transcript" in r.stderr and "Finalizing diarization" in r.stderr
    words = [w for ln in tagged for w in ln.split()[1:]]
    assert words == lines[0].split()                                                   # every word once, in order
    assert all(ln.split()[0] in ("[spk_0]", "[spk_1]", "[spk_-1]") for ln in tagged)
    spans = [ln.split() for ln in rttm.read_text().splitlines()]
    assert spans and all(s[0] == "SPEAKER" and s[7] in ("spk_0", "spk_1") and float(s[4]) > 0 for s in spans)
    recs = [json.loads(ln) for ln in js.read_text().splitlines()]
    assert [rec["word"] for rec in recs] == words and all(0 < rec["at"] <= pcm.size / 16000 + 0.1 for rec in recs)
    assert recs == sorted(recs, key=lambda rec: rec["at"])
