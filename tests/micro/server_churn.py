#!/usr/bin/env python3
"""Sessions that start under load: N_LIVE long live streams (R = 0, 80 ms pushes) share the server with a steady
arrival of short sessions (2-3 s each, a new one every ARRIVAL s).  Reports token latency p50 / p99 / max of the LONG streams alone and of all
streams, and the transcripts.  A stream start used to queue 53 fills (32 MB) on the engine's stream in front of the live streams' next step;
it is one launch now (k_stream_reset).

    python tests/micro/server_churn.py [n_live] [n_short] [arrival_s]
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import server_load as sl  # noqa: E402
import numpy as np  # noqa: E402


def lat_of(streams):
    # the same accounting as server_load.run_load, for a subset of the streams
    rep = {}
    lat = []
    for st in streams:
        T = 1 + st.R
        want, ends = sl.expected_text(st.events)
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
    lat = np.array(lat) if lat else np.zeros(1)
    return dict(tokens=int(lat.size), p50=round(1e3 * float(np.percentile(lat, 50)), 1), p99=round(1e3 * float(np.percentile(lat, 99)), 1), max=round(1e3 * float(lat.max()), 1))


def main():
    n_live = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n_short = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    arrival = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
    work = Path("/tmp/nasr_churn")
    work.mkdir(exist_ok=True)
    model = sl.write_model(work / "speech-q8_0.gguf", 24, "q8_0")
    sock = str(work / "asr.sock")
    total = 4.0 + n_short * arrival + 3.0
    live = [sl.StreamState(i, 0, total, delay=0.02 * i) for i in range(n_live)]
    short = [sl.StreamState(100 + i, 0, 2.0 + (i * 0.31) % 1.0, delay=2.0 + arrival * i) for i in range(n_short)]
    out = {}
    for name, streams in (("live_only", [sl.StreamState(i, 0, total, delay=0.02 * i) for i in range(n_live)]), ("with_churn", live + short)):
        proc = sl.start_server(model, sock, n_live + n_short + 8, 4, extra=("--prewarm", "0"))      # the Python client opens every session up front; the short ones start PUSHING at their delay
        try:
            rep = sl.run_load(sock, len(streams), 0.0, 0, "realtime", n_conns=8, client="python", streams=streams)
        finally:
            srv, err = sl.stop_server(proc)
        out[name] = dict(streams=len(streams), transcripts_correct=rep["transcripts_correct"], errors=rep["errors"], all=rep["token_latency_ms"],
                         long_streams=lat_of([s for s in streams if s.idx < 100]), server=srv)
        print(name, json.dumps(out[name]), flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "r5_server_churn.json").write_text(json.dumps(dict(n_live=n_live, n_short=n_short, arrival_s=arrival, **out), indent=1))


if __name__ == "__main__":
    main()
