import sys,time; sys.path.insert(0,"/root/repo"); import __graft_entry__ as ge; ge.load_package()
from nemotron_asr_amd import capi, synth
W = synth.make_weights(n_layers=24)
for (B,k) in ((1,1),(2,1),(4,1),(1,4),(8,1)):
    eng = capi.Engine(W, n_layers=24, dtype=capi.DTYPE_BF16, max_streams=B)
    sts = [eng.stream(0) for _ in range(B)]
    piece=1280*k; n=piece*(300//k)
    devs = [eng.upload(synth.make_pcm(b, n/16000+0.1)[:n]) for b in range(B)]
    for o in range(0, piece*5, piece): eng.step(sts, [(d+2*o, piece) for d in devs], flags=capi.FLAG_PCM_DEVICE)
    eng.synchronize(); t0=time.perf_counter()
    for o in range(piece*5, n, piece): eng.step(sts, [(d+2*o, piece) for d in devs], flags=capi.FLAG_PCM_DEVICE)
    eng.synchronize(); dt=time.perf_counter()-t0
    print("B",B,"push_chunks",k,"RTFx", round(B*(n-piece*5)/16000/dt,1), "ms/push", round(dt/((n-piece*5)/piece)*1e3,3), flush=True)
    eng.close()
