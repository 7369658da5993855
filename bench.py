#!/usr/bin/env python3
"""bench.py -- RTFx (audio-seconds per wall-second) of the streaming forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B --right-context R --dtype bf16]

A "step" is one pass of the hot path over one batch: every stream of the batch is pushed
1280*(1+R) new PCM samples (80 ms * (1+R) of audio), which runs PCM->log-mel, one cached
encoder chunk (24 layers) and the RNN-T greedy decode for all B streams in one launch
sequence.  Default workload = BASELINE.json configs[1]: nemotron-0.6B, bf16, batch = 1 stream,
80 ms lookahead (R = 0), full-size seeded synthetic weights and PCM (no checkpoints/audio exist
on the box).  PCM is resident in HBM before the timed region starts (NASR_FLAG_PCM_DEVICE).

Multi-GPU (N > 1, launched by torch.distributed.run): streams are independent, so rank r owns
its own B streams on GPU r with replicated weights -- weak scaling, no data-path collective
(SURVEY.md §8e).  torch.distributed (RCCL) is used only for the barrier and the max-over-ranks
of the elapsed time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 (spec)
# measured-achievable on this pool (tests/micro/peaks.py): 1 GiB stream copy 5.0 TB/s, hipBLASLt bf16 8192^3 1261 TFLOP/s
HBM_MEASURED_GBS = 5020.0
MFMA_MEASURED_TFLOPS = 1261.0


# HIP-event category (engine) -> rocprofv3 kernel symbol, for the committed PMC traffic summary
_SYMBOL = {"k_fused_ln_gemm": "void nasr::k_fused_skinny<0>(nasr::FusedParams)",
           "k_fused_plain_gemm": "void nasr::k_fused_skinny<1>(nasr::FusedParams)",
           "k_fused_attn_gemm": "void nasr::k_fused_skinny<2>(nasr::FusedParams)",
           "k_fused_dwconv_gemm": "void nasr::k_fused_skinny<3>(nasr::FusedParams)"}


def pmc_traffic(category, B, R, dtype, layers):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes committed under
    profiles/ (FETCH_SIZE x2 correction + WRITE_SIZE, MI355X_MICROARCH.md).  PMC counters cannot be
    collected from inside this process; the number is reported only for the configuration it was
    measured on (batch 1, R 0, bf16, 24 layers) and is null otherwise."""
    if (B, R, dtype, layers, category) == (64, 13, "bf16", 24, "k_gemm_tiled"):       # BASELINE configs[2]/[3] shape
        f64 = ROOT / "profiles" / "r1f_pmc_batch64_R13.json"
        if not f64.exists():
            return None
        d = json.loads(f64.read_text())["derived"]["k_gemm_tiled2"]
        return int(d["fabric_read_bytes"] + d["write_bytes"])
    f = ROOT / "profiles" / "r1_pmc_traffic_batch1_R0.json"
    if (B, R, dtype, layers) != (1, 0, "bf16", 24) or not f.exists() or category not in _SYMBOL:
        return None
    k = json.loads(f.read_text())["kernels"].get(_SYMBOL[category])
    return k.get("hbm_bytes_per_launch_corrected") if k else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="streams per GPU")
    ap.add_argument("--right-context", type=int, default=0, choices=[0, 1, 6, 13])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--weights", default="f32", choices=["f32", "f16", "q8_0", "q4_0"],
                    help="GGUF tensor type of the encoder matrices handed to the engine (reference flavours, "
                         "scripts/convert_to_gguf.py:246-263); they are dequantised at upload (bf16 MFMA operands)")
    ap.add_argument("--chunks-per-step", type=int, default=1,
                    help="audio pushed per step, in chunks: > 1 = buffered / file transcription (several chunks of a stream "
                         "go through the layers as one launch sequence)")
    ap.add_argument("--sync-steps", action="store_true",
                    help="every step returns its own tokens before the next one starts (engine option pipeline = 0); default: "
                         "pipelined steps, the decode graph of step s runs on a second HIP stream beside the encoder graph of step s + 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=60.0,
                    help="audio seconds of the CPU-baseline sample (~10-20 s of CPU work on 16 host threads)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--no-buffered", action="store_true", help="skip the buffered-audio (file transcription) figure")
    ap.add_argument("--diarize", action="store_true",
                    help="BASELINE config 5: also time the diarization side-car on each step's audio (MarbleNet VAD on every "
                         "10 ms window + TitaNet-L embeddings of 1.5 s sub-segments at a 0.75 s shift, random-init weights)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1 or os.environ.get("NASR_BENCH_FORCE_DIST"):      # the knob exercises the RCCL plumbing on a 1-GPU box
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ge.load_package()
    from nemotron_asr_amd import capi, sharding, synth

    B, R = args.batch, args.right_context
    T = 1 + R
    n_step = synth.shift_samples(R) * args.chunks_per_step
    total_steps = args.warmup + args.steps
    audio_per_step = B * n_step / synth.SAMPLE_RATE

    t0 = time.time()
    W = synth.make_weights(n_layers=args.layers)
    t_weights = time.time() - t0
    dtype = capi.DTYPE_BF16 if args.dtype == "bf16" else capi.DTYPE_F32
    engW = W
    if args.weights != "f32":
        engW, W = synth.quantize_weights(W, args.weights)     # engine gets the packed blocks, the CPU baseline their values
    eng = capi.Engine(engW, n_layers=args.layers, dtype=dtype, max_streams=B, device=local_rank)
    del engW
    eng.set_option("pipeline", 0 if args.sync_steps else 1)
    streams = [eng.stream(R) for _ in range(B)]
    # PCM for every step, resident in HBM before timing starts.  Extra steps for the profile pass.
    prof_steps = 0 if args.no_profile_pass else min(args.steps, 50)
    n_total = (total_steps + prof_steps) * n_step
    secs = n_total / synth.SAMPLE_RATE
    pcm_host = [synth.make_pcm(sid, secs)[:n_total] for sid in sharding.stream_ids(rank, world, B)]
    pcm_dev = [eng.upload(p) for p in pcm_host]

    L = capi.lib()
    handles = (C.c_void_p * B)(*[s.h for s in streams])
    tok_cap = 16 * T * args.chunks_per_step
    tok_bufs = [np.zeros(tok_cap, np.int32) for _ in range(B)]
    tptrs = (C.c_void_p * B)(*[b.ctypes.data for b in tok_bufs])
    caps = (C.c_int32 * B)(*([tok_cap] * B))
    ntok = (C.c_int32 * B)()
    ns = (C.c_int32 * B)(*([n_step] * B))
    step_ptrs = [(C.c_void_p * B)(*[pcm_dev[s] + 2 * k * n_step for s in range(B)]) for k in range(total_steps + prof_steps)]

    tokens_total = 0

    def run_step(k):
        nonlocal tokens_total
        rc = L.nasr_engine_step(eng.h, handles, B, step_ptrs[k], ns, tptrs, caps, ntok, capi.FLAG_PCM_DEVICE)
        if rc < 0:
            raise RuntimeError(L.nasr_last_error().decode())
        tokens_total += sum(ntok[b] for b in range(B))

    def barrier():
        sharding.barrier(dist, eng.synchronize)
        if dist is not None:
            torch.cuda.synchronize()            # torch's own stream (RCCL barrier); the engine's stream is synchronised above

    for k in range(args.warmup):
        run_step(k)
    eng.collect(streams)                       # pipelined steps: the last warm-up step's tokens are not the timed region's
    tokens_total = 0
    barrier()
    t_start = time.perf_counter()
    for k in range(args.warmup, total_steps):
        run_step(k)
    barrier()
    elapsed = time.perf_counter() - t_start                # barrier() = engine synchronize: the last decode graph has finished
    elapsed = sharding.max_over_ranks(dist, elapsed, device="cuda")
    tokens_total += sum(len(t) for t in eng.collect(streams))       # tokens of the last pipelined step (host queue)
    tokens_timed = tokens_total
    chunks_timed = streams[0].stats().chunks

    value = sharding.aggregate_rtfx(world, audio_per_step * args.steps, elapsed)

    # ---- per-kernel HIP-event pass (same steps, events around every launch) ------------------
    roofline = None
    kernels = []
    if prof_steps and rank == 0:
        eng.profile(True)
        for k in range(total_steps, total_steps + prof_steps):
            run_step(k)
        kernels = eng.profile_read()
        eng.profile(False)
        dom = max((k for k in kernels if "gemm" in k["name"]), key=lambda k: k["total_ms"], default=None)
        if dom and dom["total_ms"] > 0:
            # Events bracket every launch of the (eager) profile pass, so each bracket also holds ~2 us of launch
            # latency that the timed region (graph replay) does not pay.  The kernel's duration inside the timed
            # region = its share of the bracketed time x the timed step; this is what rocprofv3 reports for the
            # same command (profiles/).  The raw bracket average is kept beside it.
            # pipelined steps: the decode graph is off the timed critical path, the step time is shared by the rest
            ev_total = sum(k["total_ms"] for k in kernels if args.sync_steps or k["name"] != "k_dec_iter")
            share = dom["total_ms"] / ev_total
            per_step = dom["launches"] / prof_steps
            avg_ms = share * (1e3 * elapsed / args.steps) / per_step
            avg_ev_us = 1e3 * dom["total_ms"] / dom["launches"]
            if dom["name"] == "k_gemm_tiled":
                ach = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
                roofline = dict(bound="mfma", kernel=dom["name"], achieved=round(ach, 2), peak=MFMA_PEAK_TFLOPS,
                                unit="TFLOP/s", frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                                traffic=pmc_traffic(dom["name"], B * args.chunks_per_step, R, args.dtype, args.layers),
                                peak_measured=MFMA_MEASURED_TFLOPS, frac_of_measured=round(ach / MFMA_MEASURED_TFLOPS, 4),
                                hbm_frac=round(dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                avg_launch_us=round(avg_ms * 1e3, 3), avg_launch_us_event_brackets=round(avg_ev_us, 3),
                                launches_per_step=per_step, share_of_step=round(share, 4))
            else:
                traffic = pmc_traffic(dom["name"], B * args.chunks_per_step, R, args.dtype, args.layers)
                ach = dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9
                roofline = dict(bound="hbm", kernel=dom["name"], achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                                unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                                peak_measured=HBM_MEASURED_GBS, frac_of_measured=round(ach / HBM_MEASURED_GBS, 4),
                                mfma_frac=round(dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 6),
                                avg_launch_us=round(avg_ms * 1e3, 3), avg_launch_us_event_brackets=round(avg_ev_us, 3),
                                launches_per_step=per_step, share_of_step=round(share, 4),
                                alg_bytes_per_launch=round(dom["bytes"] / dom["launches"]))

    # ---- the same stream fed buffered audio (file transcription): 256 chunks per push share one launch sequence ----
    buffered = None
    if rank == 0 and world == 1 and (B, args.chunks_per_step) == (1, 1) and not args.no_buffered:
        G, n_push = 256, 5
        nb = synth.shift_samples(R) * G
        pb = synth.make_pcm(1000, (n_push + 1) * nb / synth.SAMPLE_RATE + 0.01)[:(n_push + 1) * nb]
        db = eng.upload(pb)
        sb = streams[0]                         # the timed region is over: reuse its slot
        sb.reset()
        hb = (C.c_void_p * 1)(sb.h)
        tb = np.zeros(16 * T * G, np.int32)
        tpb = (C.c_void_p * 1)(tb.ctypes.data)
        cb = (C.c_int32 * 1)(tb.size)
        nbb = (C.c_int32 * 1)(nb)
        ntb = (C.c_int32 * 1)()
        tt = []
        for k in range(n_push + 1):
            ptr = (C.c_void_p * 1)(db + 2 * k * nb)
            eng.synchronize()
            tq = time.perf_counter()
            if L.nasr_engine_step(eng.h, hb, 1, ptr, nbb, tpb, cb, ntb, capi.FLAG_PCM_DEVICE) < 0:
                raise RuntimeError(L.nasr_last_error().decode())
            eng.synchronize()
            tt.append(time.perf_counter() - tq)
        tsum = sum(tt[1:])                      # the first push builds the graph
        buffered = dict(chunks_per_push=G, pushes=n_push, ms_per_push=round(1e3 * tsum / n_push, 3),
                        value=round(n_push * nb / synth.SAMPLE_RATE / tsum, 1), unit="audio-s/s",
                        note="same engine, same stream semantics (80 ms lookahead, chunk-by-chunk caches), the 256 chunks of "
                             "a push go through every layer as one launch sequence; not the headline value")

    # ---- diarization side-car on the same audio (BASELINE config 5) ---------------------------------------------------
    diar = None
    if args.diarize and rank == 0:
        dW = synth.make_diar_weights()
        deng = capi.Diar(dW, dtype=capi.DTYPE_BF16, max_segments=max(8, 2 * B), device=local_rank)
        hist = 10080 - 160                                   # samples of history a new 10 ms hop needs
        # the side-car reads the SAME s16 PCM the ASR streams were fed, already resident in HBM
        vad_ptrs = [pcm_dev[b] for b in range(B)]
        vad_n = [hist + n_step] * B
        n_seg = max(1, int(round(B * (n_step / synth.SAMPLE_RATE) / 0.75)))                          # sub-segment shift 0.75 s
        seg_ptrs = [pcm_dev[i % B] + 2 * 12000 * (i // B) for i in range(n_seg)]
        deng.vad_device_s16(vad_ptrs, vad_n); deng.embed_device_s16(seg_ptrs)
        reps = 10
        eng.synchronize()
        tq = time.perf_counter()
        for _ in range(reps):
            pv = deng.vad_device_s16(vad_ptrs, vad_n)
        t_vad = (time.perf_counter() - tq) / reps
        tq = time.perf_counter()
        for _ in range(reps):
            deng.embed_device_s16(seg_ptrs)
        t_spk = (time.perf_counter() - tq) / reps
        # the same work overlapped: the side-car runs on its own HIP stream from a second host thread while the ASR
        # step of the same audio runs on the engine's stream (ctypes releases the GIL during both calls)
        import threading
        n_ov = min(args.steps, 20)
        gate_go, gate_done = threading.Barrier(2), threading.Barrier(2)

        def side_car():
            for _ in range(n_ov):
                gate_go.wait()
                deng.vad_device_s16(vad_ptrs, vad_n)
                deng.embed_device_s16(seg_ptrs)
                gate_done.wait()

        th = threading.Thread(target=side_car)
        th.start()
        for s in streams:
            s.reset()
        run_step(0)
        eng.synchronize()
        tq = time.perf_counter()
        for k in range(n_ov):
            gate_go.wait()
            run_step(1 + k)
            gate_done.wait()
        t_ov = (time.perf_counter() - tq) / n_ov
        th.join()
        step_s = elapsed / args.steps
        diar = dict(overlapped_ms_per_step=round(1e3 * t_ov, 3), overlapped_rtfx=round(audio_per_step / t_ov, 1),
                    vad_windows_per_step=int(sum(x.size for x in pv)), vad_ms_per_step=round(1e3 * t_vad, 3),
                    embeddings_per_step=n_seg, embed_ms_per_step=round(1e3 * t_spk, 3),
                    asr_plus_diarization_rtfx=round(audio_per_step / (step_s + t_vad + t_spk), 1),
                    note="side-car on the streams' own s16 PCM, device-resident; *_ms_per_step: run alone after the ASR step, "
                         "overlapped_*: on its own HIP stream from a second host thread beside the ASR step")
        deng.close()

    # ---- CPU baseline: the oracle (a port of the reference's algorithm), bounded sample ------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        om = ob.OracleModel(W, args.layers)
        ost = ob.OracleStream(om, R)
        n_cpu_steps = max(2, int(args.cpu_seconds * synth.SAMPLE_RATE / n_step))
        p = synth.make_pcm(sharding.stream_ids(rank, world, B)[0], args.cpu_seconds + 3 * n_step / synth.SAMPLE_RATE + 1.0)
        ost.process(p[:2 * n_step])            # warm-up: fills the first chunk
        c0 = ost.total_chunks
        tc = time.perf_counter()
        for k in range(2, 2 + n_cpu_steps):
            ost.process(p[k * n_step:(k + 1) * n_step])
        tcpu = time.perf_counter() - tc
        cpu = dict(value=round(n_cpu_steps * n_step / synth.SAMPLE_RATE / tcpu, 3), unit="audio-s/s",
                   cores=ob.lib().orc_num_threads(), kind="port",
                   sample=f"{n_cpu_steps} steps ({n_cpu_steps * n_step / synth.SAMPLE_RATE:.2f} s of audio) of stream 0, "
                          f"same weights/PCM, f32 CPU restatement (oracle/nasr_oracle.c, OpenMP), "
                          f"{ost.total_chunks - c0} chunks in {tcpu:.2f} s")

    if rank == 0:
        out = {
            "metric": "RTFx (audio-sec/sec), nemotron-0.6B streaming forward path",
            "value": round(value, 2),
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"nemotron-speech-streaming-0.6B ({args.layers} layers) {args.dtype}"
                            + (f" from {args.weights.upper()} tensors" if args.weights != "f32" else "") + f", batch={B} stream(s)/GPU, "
                            f"{80 * T} ms lookahead (R={R}), {world}xMI355X"
                            + (f", {args.chunks_per_step} chunks pushed per step" if args.chunks_per_step > 1 else "")
                            + (" [BASELINE.json configs[1]]" if (B, R, args.dtype, args.layers, args.chunks_per_step) == (1, 0, "bf16", 24, 1) else ""),
                "streams_per_gpu": B, "right_context": R, "audio_s_per_step_per_gpu": audio_per_step,
                "parallelism": f"stream-sharded x{world}, no collectives",
                "pcm": "device-resident", "tokens_emitted": tokens_timed, "chunks": chunks_timed,
                "steps": "synchronous" if args.sync_steps else "pipelined: decode graph of step s on a second HIP stream beside the encoder graph of step s+1",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "buffered_audio": buffered,
            "diarization": diar,
            "kernels": [dict(name=k["name"], launches=k["launches"], ms=round(k["total_ms"], 3)) for k in kernels],
            "setup_s": {"weights": round(t_weights, 1)},
        }
        print(json.dumps(out), flush=True)
    for s in streams:
        s.destroy()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
