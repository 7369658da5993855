"""Debug driver: the mixed-lookahead server scenario of tests/test_gpu_server_load.py with a short client timeout; if the server is still
alive afterwards, all its thread stacks are dumped with rocgdb."""
import subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import server_load as sl
wd = Path("/tmp/nasr_mixed"); wd.mkdir(exist_ok=True)
model = wd / "speech-q8_0.gguf"
if not model.exists():
    sl.write_model(model, 24, "q8_0")
sock = str(wd / f"asr-{int(time.time()) % 100000}.sock")
streams = [sl.StreamState(40 + i, (0, 13, 1, 0, 13, 0)[i % 6], 3.0 + (i * 0.37) % 4.0, delay=0.2 * i) for i in range(12)]
proc = sl.start_server(model, sock, 16, 4)
rep = None
try:
    rep = sl.run_load(sock, 12, 0.0, 0, "realtime", n_conns=3, client="python", streams=streams, timeout=25.0)
    print({k: rep[k] for k in ("transcripts_correct", "errors", "token_latency_ms")})
    for st in streams:
        print(st.R, st.delay, st.pcm.size, len(st.send_times), repr(st.text[:50]), st.ended is not None)
except Exception as ex:
    print("client exception", repr(ex))
if proc.poll() is None:
    out = subprocess.run(["/opt/rocm/bin/rocgdb", "-p", str(proc.pid), "-batch", "-ex", "thread apply all bt 12"], capture_output=True, text=True, timeout=120)
    print(out.stdout[-6000:])
srv, err = sl.stop_server(proc)
print(srv)
print(err[-3000:])
