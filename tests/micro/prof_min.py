"""smallest engine run for bisecting profiler problems: usage prof_min.py [layers] [pipeline]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as ge
ge.load_package()
from nemotron_asr_amd import capi, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 0
host = len(sys.argv) > 3
W = synth.make_weights(n_layers=L)
print("weights", flush=True)
eng = capi.Engine(W, n_layers=L, dtype=capi.DTYPE_BF16, max_streams=1)
print("engine", flush=True)
eng.set_option("pipeline", pipe)
st = eng.stream(0)
print("stream", flush=True)
pcm = synth.make_pcm(1, 30.0)
dev = eng.upload(pcm)
print("upload", flush=True)
for k in range(int(sys.argv[4]) if len(sys.argv) > 4 else 12):
    if host:
        t = eng.step([st], [pcm[k * 1280:(k + 1) * 1280]])
    else:
        t = eng.step([st], [(dev + 2 * k * 1280, 1280)], flags=capi.FLAG_PCM_DEVICE)
    print("step", k, t, flush=True)
print(eng.finalize([st]))
eng.close()
print("done")
