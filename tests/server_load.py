#!/usr/bin/env python3
"""Load client for nemo-server-amd (own code, stdlib sockets): N streams over a few connections against ONE server process.

    python tests/server_load.py --streams 64 --seconds 20 --right-context 0 --mode realtime [--pipeline 4] [--model m.gguf]

Modes: `realtime` = every stream pushes 80 ms x (1 + R) of audio every 80 ms x (1 + R) (a live microphone), `burst` = pushes
back to back (a backlog / files).  Wire protocol = the reference's (src/server-protocol.h:24-41).  The audio is
synth.make_speech_pcm with the 'speech' synthetic checkpoint, so the expected transcript of every stream is known without
running anything: the phone sequence of its audio (tests/test_speech_checkpoint.py pins that against the oracle).

Reports one JSON object: aggregate RTFx (audio seconds of all streams / wall seconds from the first push to the last ENDED),
per-token latency p50 / p99 / max (time from SENDING the push that completes the token's chunk to RECEIVING the text that
contains it; realtime mode), the histogram of streams per engine call (the server prints it at exit), transcripts correct."""
from __future__ import annotations

import argparse
import json
import re
import signal
import socket
import struct
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from nemotron_asr_amd import gguf_io, synth  # noqa: E402

BIN = ROOT / "nemotron-asr.cpp_amd" / "bin"
OP_START, OP_PUSH, OP_END, OP_STARTED, OP_ACK, OP_TEXT, OP_ENDED, OP_ERROR = 0x01, 0x02, 0x03, 0x81, 0x82, 0x83, 0x84, 0x8F


def send_frame(sock, op, sid, payload=b""):
    sock.sendall(struct.pack("<BII", op, sid, len(payload)) + payload)


def recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("server closed the connection")
        buf += part
    return bytes(buf)


def recv_frame(sock):
    op, sid, ln = struct.unpack("<BII", recv_exact(sock, 9))
    return op, sid, recv_exact(sock, ln) if ln else b""


def write_model(path: Path, n_layers=24, kind="q8_0"):
    """the speech checkpoint as a GGUF file in the reference converter's layout"""
    W = synth.make_weights(n_layers, margins="speech" if n_layers == 24 else "random")
    engW = synth.quantize_weights(W, kind)[0] if kind != "f32" else W
    gguf_io.write_gguf(path, engW, gguf_io.default_hparams(n_layers=n_layers), gguf_io.synthetic_vocab())
    return path


def expected_text(events):
    vocab = gguf_io.synthetic_vocab()
    pieces = [vocab[synth.phone_token(k)] for k, _, _ in events]
    text, ends = "", []
    for p in pieces:
        text += (" " + p[1:]) if p.startswith("▁") else p
        ends.append(len(text))
    return text, ends


class StreamState:
    def __init__(self, idx, R, seconds, delay=0.0):
        self.idx, self.R, self.delay = idx, R, delay
        self.pcm, self.events = synth.make_speech_pcm(idx, seconds)
        self.n_push = synth.shift_samples(R)
        self.sid = None
        self.send_times = []             # wall time of every push
        self.text = ""
        self.arrivals = []               # (wall time, cumulative text length)
        self.ended = None


def _exchange_python(streams, sock_path, R, mode, n_conns, timeout):
    """the stdlib-socket client of round 3 (one Python process: it bounds the burst figures itself)"""
    conns = []
    for c in range(min(n_conns, len(streams))):
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.settimeout(timeout)
        s.connect(sock_path)
        conns.append(s)
    by_conn = [[st for i, st in enumerate(streams) if i % len(conns) == c] for c in range(len(conns))]
    by_sid = {}
    for c, sock in enumerate(conns):
        for st in by_conn[c]:
            send_frame(sock, OP_START, 0, json.dumps({"lang": "auto", "right_context": st.R}).encode())
            op, sid, payload = recv_frame(sock)
            assert op == OP_STARTED, (hex(op), payload)
            st.sid = sid
            by_sid[sid] = st
    errors = []

    def receiver(c):
        sock, left = conns[c], len(by_conn[c])
        try:
            while left:
                op, sid, payload = recv_frame(sock)
                now = time.perf_counter()
                if op == OP_ACK:
                    continue
                st = by_sid[sid]
                if op in (OP_TEXT, OP_ENDED):
                    st.text += payload.decode()
                    st.arrivals.append((now, len(st.text)))
                    if op == OP_ENDED:
                        st.ended = now
                        left -= 1
                else:
                    errors.append((sid, hex(op), payload[:100]))
                    left -= 1
        except Exception as ex:      # noqa: BLE001
            errors.append((c, "receiver", repr(ex)))

    def sender(c, t_start):
        # every stream of the connection on its own clock: first push at t_start + delay, then one push per n_push samples of real time
        # (realtime) or as fast as the socket takes them (burst); STREAM_END right behind a stream's last push
        sock, mine = conns[c], by_conn[c]
        nxt = {st.idx: 0 for st in mine}
        live = list(mine)
        while live:
            if mode == "realtime":
                due = lambda st: t_start + st.delay + nxt[st.idx] * st.n_push / synth.SAMPLE_RATE
                st = min(live, key=due)
                d = due(st) - time.perf_counter()
                if d > 0:
                    time.sleep(d)
                batch = [st]
            else:
                batch = list(live)
            for st in batch:
                k = nxt[st.idx]
                piece = st.pcm[k * st.n_push:(k + 1) * st.n_push]
                if piece.size:
                    st.send_times.append(time.perf_counter())
                    send_frame(sock, OP_PUSH, st.sid, piece.tobytes())
                nxt[st.idx] = k + 1
                if (k + 1) * st.n_push >= st.pcm.size:
                    send_frame(sock, OP_END, st.sid)
                    live.remove(st)

    rx = [threading.Thread(target=receiver, args=(c,)) for c in range(len(conns))]
    for t in rx:
        t.start()
    t0 = time.perf_counter()
    tx = [threading.Thread(target=sender, args=(c, t0 + 0.05)) for c in range(len(conns))]
    for t in tx:
        t.start()
    for t in tx + rx:
        t.join(timeout)
    t_end = max([st.ended or time.perf_counter() for st in streams])
    for s in conns:
        s.close()
    return t_end - (t0 + 0.05), errors


def _exchange_native(streams, sock_path, R, mode, n_conns, timeout, workdir):
    """bin/nemo-load-amd (host/nemo_load_client.cpp): one sender + one receiver thread per connection, PCM from files"""
    d = Path(workdir) / f"pcm-{int(time.time() * 1e3) % 10 ** 7}"
    d.mkdir(parents=True, exist_ok=True)
    for i, st in enumerate(streams):
        st.pcm.astype("<i2").tofile(d / f"stream_{i:04d}.s16")
    rep_path = d / "report.json"
    r = subprocess.run([str(BIN / "nemo-load-amd"), "--unix", sock_path, "--pcm-dir", str(d), "--streams", str(len(streams)), "--conns", str(n_conns),
                        "--right-context", str(R), "--mode", mode, "--out", str(rep_path)], capture_output=True, text=True, timeout=timeout)
    errors = [("client", r.returncode, r.stderr[-300:])] if r.returncode != 0 else []
    rep = json.loads(rep_path.read_text()) if rep_path.exists() else {"per_stream": [], "wall_seconds": float("nan")}
    for st, ps in zip(streams, rep["per_stream"]):
        raw = ps["text"].encode()
        st.sid, st.text, st.send_times = ps["sid"], ps["text"], ps["send_times"]
        st.arrivals = [(t, len(raw[:nb].decode(errors="ignore"))) for t, nb in ps["arrivals"]]
        st.ended = ps["ended"]
        if ps["error"]:
            errors.append((ps["sid"], "ERROR", ps["error"][:100]))
    for f in d.iterdir():
        f.unlink()
    d.rmdir()
    return rep["wall_seconds"], errors


def run_load(sock_path, n_streams, seconds, R, mode, n_conns=8, timeout=300.0, client="python", workdir="/tmp/nasr_load", streams=None):
    """streams: a prepared list of StreamState (own lookahead, length and start delay per stream: Python client only); default: n_streams alike"""
    streams = streams if streams is not None else [StreamState(i, R, seconds) for i in range(n_streams)]
    n_streams = len(streams)
    if client == "native":
        wall, errors = _exchange_native(streams, sock_path, R, mode, n_conns, timeout, workdir)
    else:
        wall, errors = _exchange_python(streams, sock_path, R, mode, n_conns, timeout)
    # per-token latency: token k of a stream is emitted at the first frame wholly inside its phone; the chunk holding that
    # frame is complete once sample (chunk + 1) x 1280 T (+ the STFT's 400-sample reach) has been pushed
    lat, correct = [], 0
    for st in streams:
        T = 1 + st.R
        want, ends = expected_text(st.events)
        correct += st.text == want
        if st.text != want:
            continue
        ai = 0
        for (k, a, b), end in zip(st.events, ends):
            f = -(-a // 1280)
            need = ((f // T) + 1) * T * 1280 + 400
            push = min(need // st.n_push, len(st.send_times) - 1)
            while ai < len(st.arrivals) and st.arrivals[ai][1] < end:
                ai += 1
            if ai < len(st.arrivals):
                lat.append(st.arrivals[ai][0] - st.send_times[push])
    audio_s = sum(st.pcm.size for st in streams) / synth.SAMPLE_RATE
    lat = np.array(lat) if lat else np.zeros(1)
    return dict(mode=mode, client=client, streams=n_streams, right_context=R, audio_seconds=round(audio_s, 1), wall_seconds=round(wall, 3),
                aggregate_rtfx=round(audio_s / wall, 1), transcripts_correct=correct, tokens=int(lat.size), errors=errors[:5],
                token_latency_ms=dict(p50=round(1e3 * float(np.percentile(lat, 50)), 1), p99=round(1e3 * float(np.percentile(lat, 99)), 1),
                                      max=round(1e3 * float(lat.max()), 1)))


def start_server(model, sock_path, max_streams, pipeline, extra=()):
    cmd = [str(BIN / "nemo-server-amd"), str(model), "--unix", sock_path, "--max-streams", str(max_streams)] + (["--pipeline", str(pipeline)] if pipeline else []) + list(extra)
    proc = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)
    for _ in range(1800):
        if Path(sock_path).exists() or proc.poll() is not None:
            break
        time.sleep(0.1)
    if proc.poll() is not None:
        raise RuntimeError("server did not start: " + proc.stderr.read()[-2000:])
    return proc


def stop_server(proc):
    proc.send_signal(signal.SIGTERM)
    try:
        _, err = proc.communicate(timeout=60)
    except subprocess.TimeoutExpired:
        proc.kill()
        _, err = proc.communicate()
    hist = {}
    for line in err.splitlines():
        m = re.match(r"worker: B histogram(.*)", line)
        if m:
            for tok in m.group(1).split():
                b, n = tok.split(":")
                hist[int(b)] = hist.get(int(b), 0) + int(n)
    calls = sum(hist.values())
    counters, partial, tails = {}, 0, 0
    for line in err.splitlines():
        m = re.match(r"worker: engine counters(.*)", line)
        if m:
            for tok in m.group(1).split():
                k, v = tok.split(":")
                counters[k] = counters.get(k, 0) + int(v)
        m = re.search(r"(\d+) partial-chunk calls, (\d+) tail flushes", line)
        if m:
            partial += int(m.group(1))
            tails += int(m.group(2))
    eager = counters.get("eager_steps", 0)
    worker_time = [dict(engine_calls_s=float(m.group(1)), events_into_sessions_s=float(m.group(2)), waiting_s=float(m.group(3)))
                   for m in re.finditer(r"worker: wall time by activity: ([0-9.]+) s inside engine calls, ([0-9.]+) s moving received events into the sessions, ([0-9.]+) s waiting", err)]
    return dict(worker_time=worker_time, engine_calls=calls, partial_chunk_calls=partial, tail_flushes=tails, engine_counters=counters,
                eager_share=round(eager / max(calls, 1), 4),
                # what is eager by construction: the sub-chunk remainder of sessions that end (one call per group of sessions that end
                # together) and the tail flush behind it; everything else should be a graph replay
                eager_outside_stream_end=max(0, eager - partial - tails), streams_per_call_mean=round(sum(b * n for b, n in hist.items()) / max(calls, 1), 2),
                b_histogram={str(b): n for b, n in sorted(hist.items())}), err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--right-context", type=int, default=0, choices=[0, 1, 6, 13])
    ap.add_argument("--mode", default="realtime", choices=["realtime", "burst"])
    ap.add_argument("--pipeline", type=int, default=4)
    ap.add_argument("--model", default="")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--workdir", default="/tmp/nasr_load")
    ap.add_argument("--client", default="native", choices=["native", "python"])
    ap.add_argument("--no-prewarm", action="store_true", help="start the server with --no-prewarm (default: --prewarm <right-context>)")
    ap.add_argument("--warmup-seconds", type=float, default=0.0,
                    help="a first load of this many seconds per stream on the same server (graph captures, lane picking), not reported")
    ap.add_argument("--conns", type=int, default=8)
    ap.add_argument("--backlog-chunks", type=int, default=0, help="passed to the server (0: its default): whole chunks of every stream one engine call may carry")
    ap.add_argument("--devices", default="", help="passed to the server: one engine + worker per entry (\"0,0,0,0\": four engines on the one GPU of a test box), stream s on entry s mod count")
    args = ap.parse_args()
    wd = Path(args.workdir)
    wd.mkdir(parents=True, exist_ok=True)
    model = Path(args.model) if args.model else wd / f"speech-{args.layers}L-q8_0.gguf"
    if not model.exists():
        write_model(model, args.layers)
    sock_path = str(wd / f"asr-{int(time.time() * 1000) % 100000}.sock")
    extra = ("--no-prewarm",) if args.no_prewarm else ("--prewarm", str(args.right_context))
    n_dev = len(args.devices.split(",")) if args.devices else 1
    if args.devices:
        extra += ("--devices", args.devices)
    if args.backlog_chunks:
        extra += ("--backlog-chunks", str(args.backlog_chunks))
    proc = start_server(model, sock_path, -(-args.streams // n_dev) if args.devices else args.streams, args.pipeline, extra=extra)
    try:
        if args.warmup_seconds > 0:
            run_load(sock_path, args.streams, args.warmup_seconds, args.right_context, args.mode, n_conns=args.conns, client=args.client, workdir=args.workdir)
        rep = run_load(sock_path, args.streams, args.seconds, args.right_context, args.mode, n_conns=args.conns, client=args.client, workdir=args.workdir)
    finally:
        srv, err = stop_server(proc)
    rep["server"] = srv
    rep["pipeline"] = args.pipeline
    rep["backlog_chunks"] = args.backlog_chunks or "server default (4)"
    if args.devices:
        rep["devices"] = args.devices
        rep["server_stderr_tail"] = [ln for ln in err.splitlines() if "worker" in ln or "device" in ln][-12:]
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
