"""Encoder features of the speech-like synthetic audio through the F32 oracle (CPU), for fitting the 'speech' joint.
usage: speech_feats.py n_layers R first_stream n_streams seconds out.npz [emulate_bf16]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge
ge.load_package()
from nemotron_asr_amd import synth
from oracle import binding as ob

nl, R, s0, ns, secs, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6]
emu = len(sys.argv) > 7 and sys.argv[7] == "1"
W = synth.make_weights(nl)
M = ob.OracleModel(W, nl, emulate_bf16=emu)
feats, labs, cover, sid = [], [], [], []
t0 = time.time()
for s in range(s0, s0 + ns):
    pcm, ev = synth.make_speech_pcm(s, secs)
    pp = ob.OraclePreproc(W["preprocessor.featurizer.fb"], W["preprocessor.featurizer.window"])
    mel = pp.process(pcm)
    st = ob.OracleStream(M, R)
    T, cm = st.T, st.chunk_mel
    # chunking as the stream driver does it (src/nemo-stream.cpp:1145-1206): first chunk has 9 zero-frames of left context
    buf = np.concatenate([np.zeros((9, 128), np.float32), mel])
    shift = 8 * T
    outs = []
    pos = 0
    while pos + cm <= buf.shape[0]:
        outs.append(st.encode_chunk(buf[pos:pos + cm]))
        pos += shift
    e = np.concatenate(outs)
    # frame f covers samples [f*1280, (f+1)*1280): label = phone covering it fully, -1 silence if no overlap, -2 partial
    lab = np.full(e.shape[0], -1, np.int32)
    cov = np.zeros(e.shape[0], np.float32)
    for k, a, b in ev:
        f0, f1 = a // 1280, (b - 1) // 1280
        for f in range(f0, min(f1 + 1, e.shape[0])):
            c = (min(b, (f + 1) * 1280) - max(a, f * 1280)) / 1280.0
            cov[f] = c
            lab[f] = k if c >= 0.999 else -2 - k
    feats.append(e); labs.append(lab); cover.append(cov); sid.append(np.full(e.shape[0], s, np.int32))
    print(f"stream {s}: {e.shape[0]} frames, {time.time() - t0:.1f}s", flush=True)
np.savez(out, e=np.concatenate(feats), lab=np.concatenate(labs), cov=np.concatenate(cover), sid=np.concatenate(sid))
