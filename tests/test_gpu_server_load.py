"""The product surface of the reference is its multi-stream server (src/nemo-server.cpp:192-271).  tests/server_load.py drives
nemo-server-amd --pipeline 4 with 16 live streams (real-time paced; Python client) and with a burst (native client, bin/nemo-load-amd), on a 24-layer Q8_0 GGUF of the speech
checkpoint: every stream's transcript must be the phone sequence of its audio (= the oracle's transcript,
tests/test_speech_checkpoint.py), three of them are checked against the F32 oracle directly, and the report carries aggregate
RTFx, token latency and the histogram of streams per engine call."""
import json
import os
from pathlib import Path

import pytest

from nemotron_asr_amd import gguf_io, synth
from oracle import binding as ob
from tests import server_load as sl

pytestmark = pytest.mark.gpu


def test_server_under_load_16_streams(tmp_path):
    model = sl.write_model(tmp_path / "speech-q8_0.gguf", 24, "q8_0")
    reports = {}
    for mode, seconds, R in (("realtime", 8.0, 0), ("burst", 12.0, 13)):
        sock = str(tmp_path / f"asr-{mode}.sock")
        proc = sl.start_server(model, sock, 16, 4, extra=("--prewarm", str(R)))
        try:
            rep = sl.run_load(sock, 16, seconds, R, mode, n_conns=4, client="native" if mode == "burst" else "python", workdir=str(tmp_path))
        finally:
            srv, err = sl.stop_server(proc)
        rep["server"] = srv
        reports[mode] = rep
        assert not rep["errors"], rep
        assert rep["transcripts_correct"] == 16, rep
        assert srv["engine_calls"] > 0 and sum(srv["b_histogram"].values()) == srv["engine_calls"]
        # the batch former hands whole chunks only: what runs eagerly is each session's sub-chunk tail before its STREAM_END
        # (one call for all sessions that end together) -- never the steady state
        c = srv["engine_counters"]
        assert c.get("graph_replays", 0) >= srv["engine_calls"] - srv["partial_chunk_calls"] - 2, srv       # every pipelined step is a graph replay too
        assert srv["partial_chunk_calls"] <= 16, srv
    assert reports["realtime"]["token_latency_ms"]["p99"] < 250.0, reports["realtime"]       # live streams: text within a quarter second of the audio
    assert reports["burst"]["aggregate_rtfx"] > 200.0, reports["burst"]
    d = os.environ.get("NASR_REPORT_DIR")
    if d:
        Path(d).mkdir(parents=True, exist_ok=True)
        (Path(d) / "server_load.json").write_text(json.dumps(reports, indent=1))
    # three streams against the F32 oracle on the dequantised weights (R = 13: the burst run's lookahead)
    W = synth.make_weights(24, margins="speech")
    _, deq = synth.quantize_weights(W, "q8_0")
    om = ob.OracleModel(deq, 24)
    vocab = gguf_io.synthetic_vocab()
    for b in range(3):
        pcm, ev = synth.make_speech_pcm(b, 12.0)
        ost = ob.OracleStream(om, 13)
        ref = ost.process(pcm) + ost.finalize()
        text = "".join((" " + vocab[t][1:]) if vocab[t].startswith("▁") else vocab[t] for t in ref)
        assert text == sl.expected_text(ev)[0]


def test_server_with_mixed_lookaheads_and_sessions_that_come_and_go(tmp_path):
    """The batch former under what a real population does: 12 live streams with lookahead 0, 1 and 13 mixed on 3 connections, starting 0 ... 2.2 s
    apart and 3 ... 7 s long, so that sessions join and leave while others are mid-stream and every call groups whoever holds a whole chunk
    of its lookahead.  Only the default lookahead is prewarmed (the other shapes are captured on the way).  Every transcript == the phone
    sequence of its audio; nothing runs eagerly but stream ends."""
    model = sl.write_model(tmp_path / "speech-q8_0.gguf", 24, "q8_0")
    sock = str(tmp_path / "asr-mixed.sock")
    streams = [sl.StreamState(40 + i, (0, 13, 1, 0, 13, 0)[i % 6], 3.0 + (i * 0.37) % 4.0, delay=0.2 * i) for i in range(12)]
    proc = sl.start_server(model, sock, 16, 4)
    try:
        rep = sl.run_load(sock, 12, 0.0, 0, "realtime", n_conns=3, client="python", streams=streams)
    finally:
        srv, err = sl.stop_server(proc)
    assert not rep["errors"], rep
    assert rep["transcripts_correct"] == 12, (rep, [(st.R, st.text[:40]) for st in streams])
    assert srv["eager_outside_stream_end"] <= 2, srv
    assert rep["token_latency_ms"]["p99"] < 400.0, rep          # shapes are captured on the way: a first call of a new batch size costs ~20 ms


def test_server_with_four_engines_spreads_the_streams(tmp_path):
    """`--devices` = one engine + FIFO + worker thread per entry, stream s on entry s mod count (src/nemo-server.cpp has one model and one worker; the
    N-GPU form of this server is what an 8-GPU node runs).  On the one GPU of a test box: `--devices 0,0,0,0`, 32 live streams (round 4 tested two
    entries with a handful of streams): every transcript correct, every engine serves calls, and no engine is handed more than its share."""
    model = sl.write_model(tmp_path / "speech-q8_0.gguf", 24, "q8_0")
    sock = str(tmp_path / "asr-4dev.sock")
    proc = sl.start_server(model, sock, 8, 4, extra=("--prewarm", "0", "--devices", "0,0,0,0"))
    try:
        rep = sl.run_load(sock, 32, 5.0, 0, "realtime", n_conns=4, client="native", workdir=str(tmp_path))
    finally:
        srv, err = sl.stop_server(proc)
    assert not rep["errors"], rep
    assert rep["transcripts_correct"] == 32, rep
    hists = [ln for ln in err.splitlines() if ln.startswith("worker: B histogram")]
    assert len(hists) == 4, err[-2000:]                      # four workers printed their own histogram
    for ln in hists:
        sizes = [int(tok.split(":")[0]) for tok in ln.split()[3:]]
        calls = sum(int(tok.split(":")[1]) for tok in ln.split()[3:])
        assert calls > 20 and max(sizes) <= 8, ln            # 32 streams mod 4: eight per engine, never more
    assert rep["token_latency_ms"]["p99"] < 250.0, rep
