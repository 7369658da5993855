"""Seeded synthetic workload: full-size weights and PCM.

There are no model weights and no audio on either box (reference `.gitignore:16`
ignores `weights/`), so parity tests and `bench.py` run on deterministic synthetic
tensors with the reference's names, shapes and on-disk conventions
(reference `src/nemo-ggml.cpp:296-398`, `docs/TENSOR_SHAPES.md`,
`scripts/convert_to_gguf.py:399-413`: pointwise conv squeezed to 2-D, depthwise conv
stored as `(k, C)`).

Generator: counter-based splitmix64 (no state, vectorised, reproducible from
`(seed, tensor name, element index)` alone), values uniform in `[-a, a]` with
`a = sqrt(3 / fan_in)` (variance `1/fan_in`), computed in float64 and rounded once
to float32, so any implementation of splitmix64 reproduces the bytes.
"""
from __future__ import annotations

import numpy as np

D_MODEL = 1024
N_HEADS = 8
D_HEAD = 128
D_FF = 4096
N_MELS = 128
N_BINS = 257
N_FFT = 512
WIN = 400
HOP = 160
SUB_CH = 256
SUB_FLAT = 4352
VOCAB = 1025
BLANK = 1024
HIDDEN = 640
JOINT = 640
LEFT_CTX = 70
PRE_CACHE = 9
# synthetic decoder (see _make): token-specific refractory suppression through the prediction network
EMBED_GAIN = 7.0      # embed[v] = EMBED_GAIN * joint_out[v]
DEC_NOISE = 0.25      # amplitude of the dense random part of the LSTM / joint.pred matrices (x sqrt(3/fan_in))
SUPPRESS = 0.6        # joint.pred = -SUPPRESS * I + noise
GATE_OPEN = 3.0       # bias of the input / output gates
BLANK_GAIN = 4.5      # at 24 layers; shallow test models use BLANK_GAIN_SHALLOW (see default_blank_gain)
BLANK_GAIN_SHALLOW = 5.0
SAMPLE_RATE = 16000

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """z_i = mix(seed + (offset+i+1)*GOLDEN), i in [0, n): uint64."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """float64 in [0, 1) with 53 random bits."""
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def _tensor(seed: int, name: str, shape, amp: float, center: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    s = seed ^ _fnv1a64(name)
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    for o in range(0, n, step):
        k = min(step, n - o)
        u = uniform01(s, k, o)
        out[o:o + k] = (center + (2.0 * u - 1.0) * amp).astype(np.float32)
    return out.reshape(shape)


def mel_filterbank() -> np.ndarray:
    """Deterministic triangular mel filterbank [128][257] (HTK mel, area-normalised).

    Stands in for NeMo's `featurizer.fb`; any non-negative banded matrix exercises the
    same arithmetic (reference `src/preprocessor.cpp:374-383`)."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)

    freqs = np.linspace(0.0, SAMPLE_RATE / 2, N_BINS)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(SAMPLE_RATE / 2), N_MELS + 2))
    fb = np.zeros((N_MELS, N_BINS), dtype=np.float64)
    for m in range(N_MELS):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        up = (freqs - lo) / (ce - lo)
        dn = (hi - freqs) / (hi - ce)
        fb[m] = np.maximum(0.0, np.minimum(up, dn)) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


def hann_window() -> np.ndarray:
    """Symmetric Hann-400 (`torch.hann_window(400, periodic=False)`)."""
    n = np.arange(WIN, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / (WIN - 1))).astype(np.float32)


def layer_tensor_specs(kernel_size: int = 9):
    """(suffix, shape, kind, fan_in) for one encoder layer."""
    D, F = D_MODEL, D_FF
    return [
        ("norm_feed_forward1.weight", (D,), "ln_w", 0), ("norm_feed_forward1.bias", (D,), "bias", 0),
        ("feed_forward1.linear1.weight", (F, D), "mat", D), ("feed_forward1.linear2.weight", (D, F), "mat", F),
        ("norm_self_att.weight", (D,), "ln_w", 0), ("norm_self_att.bias", (D,), "bias", 0),
        ("self_attn.linear_q.weight", (D, D), "mat", D), ("self_attn.linear_k.weight", (D, D), "mat", D),
        ("self_attn.linear_v.weight", (D, D), "mat", D), ("self_attn.linear_pos.weight", (D, D), "mat", D),
        ("self_attn.linear_out.weight", (D, D), "mat", D),
        ("self_attn.pos_bias_u", (N_HEADS, D_HEAD), "posbias", 0),
        ("self_attn.pos_bias_v", (N_HEADS, D_HEAD), "posbias", 0),
        ("norm_conv.weight", (D,), "ln_w", 0), ("norm_conv.bias", (D,), "bias", 0),
        ("conv.pointwise_conv1.weight", (2 * D, D), "mat", D),
        ("conv.depthwise_conv.weight", (kernel_size, D), "mat", kernel_size),
        ("conv.batch_norm.weight", (D,), "ln_w", 0), ("conv.batch_norm.bias", (D,), "bias", 0),
        ("conv.pointwise_conv2.weight", (D, D), "mat", D),
        ("norm_feed_forward2.weight", (D,), "ln_w", 0), ("norm_feed_forward2.bias", (D,), "bias", 0),
        ("feed_forward2.linear1.weight", (F, D), "mat", D), ("feed_forward2.linear2.weight", (D, F), "mat", F),
        ("norm_out.weight", (D,), "ln_w", 0), ("norm_out.bias", (D,), "bias", 0),
    ]


def global_tensor_specs(num_prompts: int = 0):
    C = SUB_CH
    specs = [
        ("encoder.pre_encode.conv.0.weight", (C, 1, 3, 3), "mat", 9), ("encoder.pre_encode.conv.0.bias", (C,), "bias", 0),
        ("encoder.pre_encode.conv.2.weight", (C, 1, 3, 3), "mat", 9), ("encoder.pre_encode.conv.2.bias", (C,), "bias", 0),
        ("encoder.pre_encode.conv.3.weight", (C, C, 1, 1), "mat", C), ("encoder.pre_encode.conv.3.bias", (C,), "bias", 0),
        ("encoder.pre_encode.conv.5.weight", (C, 1, 3, 3), "mat", 9), ("encoder.pre_encode.conv.5.bias", (C,), "bias", 0),
        ("encoder.pre_encode.conv.6.weight", (C, C, 1, 1), "mat", C), ("encoder.pre_encode.conv.6.bias", (C,), "bias", 0),
        ("encoder.pre_encode.out.weight", (D_MODEL, SUB_FLAT), "mat", SUB_FLAT), ("encoder.pre_encode.out.bias", (D_MODEL,), "bias", 0),
        ("decoder.prediction.embed.weight", (VOCAB, HIDDEN), "embed", 1),
    ]
    for l in (0, 1):
        p = "decoder.prediction.dec_rnn.lstm."
        specs += [(p + f"weight_ih_l{l}", (4 * HIDDEN, HIDDEN), "lstm_ih", HIDDEN),
                  (p + f"weight_hh_l{l}", (4 * HIDDEN, HIDDEN), "lstm_hh", HIDDEN),
                  (p + f"bias_ih_l{l}", (4 * HIDDEN,), "lstm_bias", 0),
                  (p + f"bias_hh_l{l}", (4 * HIDDEN,), "lstm_bias", 0)]
    specs += [
        ("joint.enc.weight", (JOINT, D_MODEL), "mat", D_MODEL), ("joint.enc.bias", (JOINT,), "bias", 0),
        ("joint.pred.weight", (JOINT, HIDDEN), "jpred", HIDDEN), ("joint.pred.bias", (JOINT,), "bias", 0),
        ("joint.joint_net.2.weight", (VOCAB, JOINT), "jout", JOINT), ("joint.joint_net.2.bias", (VOCAB,), "bias", 0),
    ]
    if num_prompts > 0:
        specs += [
            ("prompt_kernel.0.weight", (2048, D_MODEL + num_prompts), "mat", D_MODEL + num_prompts),
            ("prompt_kernel.0.bias", (2048,), "bias", 0),
            ("prompt_kernel.2.weight", (D_MODEL, 2048), "mat", 2048),
            ("prompt_kernel.2.bias", (D_MODEL,), "bias", 0),
        ]
    return specs


def _jout_random(seed):
    return _tensor(seed, "joint.joint_net.2.weight", (VOCAB, JOINT), float(np.sqrt(3.0 / JOINT)))


def _make(seed, name, shape, kind, fan_in, blank_bias):
    if kind == "mat":
        return _tensor(seed, name, shape, float(np.sqrt(3.0 / fan_in)))
    if kind == "embed":
        w = (EMBED_GAIN * _jout_random(seed)).astype(np.float32)
        w[BLANK] = 0.0  # padding row (nn.Embedding padding_idx = blank)
        return w
    if kind == "ln_w":
        return _tensor(seed, name, shape, 0.05, center=1.0)
    if kind == "bias":
        return _tensor(seed, name, shape, 0.05)
    if kind == "posbias":
        return _tensor(seed, name, shape, 0.3)
    # ---- decoder: a hand-built stand-in for a TRAINED prediction network ---------------------------------
    # With fully random weights the greedy loop has no refractory behaviour: a frame whose best token
    # beats blank keeps beating it after the emission, so streams either stay silent or emit the
    # 10-symbol cap on every frame (measured: tests/micro/burst_stats.py), and a saturated random LSTM
    # is chaotic, which would turn token-for-token parity into a coin toss.  The structure below keeps
    # every matrix dense (a random part of amplitude DEC_NOISE rides on all of them, so the GEMM
    # arithmetic is exercised in full) but gives the recurrent path a meaning:
    #   embed[v]            = EMBED_GAIN * joint_out[v]          (the token's own output row)
    #   LSTM layer          : g-gate = identity on its input, input/output gates open (bias GATE_OPEN),
    #                         forget gate 0.5  ->  c' = 0.5 c + 0.95 tanh(x): a leaky sum of the last tokens' rows
    #   joint.pred          = -SUPPRESS * I                      ->  g = -SUPPRESS * h1'
    # so after emitting v the joint input is pushed against joint_out[v] (and, at half strength each, against
    # the tokens before it): v's logit drops by ~2 sigma, blank usually wins the next evaluation, and the same
    # token is not re-emitted on the following, similar frames.  The dynamics are contractive (forget 0.5).
    if kind in ("lstm_ih", "lstm_hh"):
        w = _tensor(seed, name, shape, float(np.sqrt(3.0 / fan_in)) * DEC_NOISE)
        if kind == "lstm_ih":
            w[2 * HIDDEN:3 * HIDDEN] += np.eye(HIDDEN, dtype=np.float32)          # gate order i, f, g, o
        return w
    if kind == "lstm_bias":
        b = _tensor(seed, name, shape, 0.05)
        if name.endswith(("bias_ih_l0", "bias_ih_l1")):
            b[0:HIDDEN] += np.float32(GATE_OPEN)
            b[3 * HIDDEN:4 * HIDDEN] += np.float32(GATE_OPEN)
        return b
    if kind == "jpred":
        w = _tensor(seed, name, shape, float(np.sqrt(3.0 / fan_in)) * DEC_NOISE)
        w -= np.float32(SUPPRESS) * np.eye(JOINT, HIDDEN, dtype=np.float32)
        return w
    if kind == "jout":
        # ~N(0, s^2) logits over 1024 tokens whose maximum scales with the activation level s; the blank row
        # is a constant g/640, i.e. blank logit = g * mean(relu(.)), which scales with s too, so the
        # blank/non-blank balance does not drift with the decoder state.  Measured on 64 streams x 39 s, R = 13
        # (tests/micro/burst_stats.py): g = BLANK_GAIN = 4.5 with SUPPRESS 0.6 -> 0.047 tokens per 80 ms frame,
        # the busiest of 64 streams emits 4.7 tokens per 1.12 s step on average and 18 at most; g = 4.4 -> 0.08
        # tokens/frame with a few 30-token steps; SUPPRESS = 0 -> 6 tokens/frame (the 10-symbol cap on most frames).
        w = _jout_random(seed)
        w[BLANK, :] = np.float32(blank_bias / fan_in)
        return w
    raise ValueError(kind)


def default_blank_gain(n_layers: int) -> float:
    """The emission rate of the synthetic joint depends on the statistics of the (random) encoder's output,
    which differ between the 1-4 layer models of the tests and the 24-layer model (measured with the CPU oracle
    and tests/micro/burst_stats.py): 4.5 gives ~0.05 tokens/frame at 24 layers but the 10-symbol cap on most frames
    at 1-4 layers, where 5.0 gives 0.1-0.7 tokens/frame."""
    return BLANK_GAIN if n_layers > 8 else BLANK_GAIN_SHALLOW


def make_weights(n_layers: int = 24, seed: int = 0xC0FFEE, kernel_size: int = 9,
                 num_prompts: int = 0, blank_bias: float = None, layers=None, margins: str = "random", readout=None) -> dict:
    """name -> float32 ndarray for the whole model (2.4 GB at 24 layers).

    `layers`: optional iterable of layer indices to materialise (default: all).
    `blank_bias`: blank gain of the synthetic joint (default: default_blank_gain(n_layers)).
    `margins`: "random" = the near-tie stress checkpoint (N(0, s^2) logits, top-2 margins of ~1e-3 everywhere);
               "speech" = same encoder bytes, decoder / joint with a fitted read-out for make_speech_pcm() audio
               (24 layers only: the fit belongs to the 24-layer encoder), see speech_decoder_tensors()."""
    if blank_bias is None:
        blank_bias = default_blank_gain(n_layers)
    w = {
        "preprocessor.featurizer.fb": mel_filterbank(),
        "preprocessor.featurizer.window": hann_window(),
    }
    for name, shape, kind, fan_in in global_tensor_specs(num_prompts):
        w[name] = _make(seed, name, shape, kind, fan_in, blank_bias)
    for l in (range(n_layers) if layers is None else layers):
        for suffix, shape, kind, fan_in in layer_tensor_specs(kernel_size):
            name = f"encoder.layers.{l}.{suffix}"
            w[name] = _make(seed, name, shape, kind, fan_in, blank_bias)
    if margins == "speech":
        scale_residual_branches(w, SPEECH_RESIDUAL_SCALE)
        w.update(speech_decoder_tensors(seed, readout))
    elif margins != "random":
        raise ValueError(margins)
    return w


def make_pcm(stream: int, seconds: float, seed: int = 0xA5A50000) -> np.ndarray:
    """int16 mono 16 kHz: 0.3*sin(2*pi*f_s*t)*env(t) + 0.05*N(0,1), f_s = 200 + 37*s Hz,
    4 Hz raised-cosine envelope."""
    n = int(round(seconds * SAMPLE_RATE))
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    f = 200.0 + 37.0 * (stream % 64)
    env = 0.5 - 0.5 * np.cos(2.0 * np.pi * 4.0 * t)
    u1 = uniform01(seed + stream, n, 0)
    u2 = uniform01(seed + stream, n, n)
    noise = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -53))) * np.cos(2.0 * np.pi * u2)
    x = 0.3 * np.sin(2.0 * np.pi * f * t) * env + 0.05 * noise
    return np.round(np.clip(x, -1.0, 1.0) * 32767.0).astype(np.int16)


def chunk_mel_frames(right_context: int) -> int:
    return PRE_CACHE + 8 * (1 + right_context)      # reference src/nemo-stream.h:65-72


def shift_samples(right_context: int) -> int:
    return HOP * 8 * (1 + right_context)            # reference src/nemo-stream.h:76-81


# ---- GGUF tensor flavours (layouts: reference scripts/convert_to_gguf.py:118-204) -----------------
QUANT_PATTERN = r"encoder\.layers\.\d+\.(feed_forward\d+|self_attn|conv)\.[^.]+\.weight$"   # :246-251
TYPE_F32, TYPE_F16, TYPE_Q4_0, TYPE_Q8_0 = 0, 1, 2, 8


def _blocks32(x: np.ndarray) -> np.ndarray:
    """flattened, zero padded to a multiple of 32 (the converter pads a ragged tail, scripts/convert_to_gguf.py:129-132)"""
    v = np.ascontiguousarray(x, np.float32).reshape(-1)
    if v.size % 32:
        v = np.concatenate([v, np.zeros(32 - v.size % 32, np.float32)])
    return v.reshape(-1, 32)


def pack_q8_0(x: np.ndarray) -> np.ndarray:
    """34-byte blocks: f16 scale (amax/127) + 32 int8 = round-half-even(x / f16 scale), no clip
    (byte-identical to the reference converter's quantize_q8_0: tests/test_gguf_reference_fixtures.py)."""
    b = _blocks32(x)
    sc = (np.abs(b).max(axis=1) / 127.0).astype(np.float16)
    s32 = sc.astype(np.float32)[:, None]
    q = np.where(s32 != 0, np.round(b / np.where(s32 != 0, s32, 1.0)), 0).astype(np.int8)
    out = np.empty(b.shape[0], dtype=np.dtype([("d", np.float16), ("q", np.int8, 32)]))
    out["d"], out["q"] = sc, q
    return out.view(np.uint8)


def unpack_q8_0(raw: np.ndarray, shape) -> np.ndarray:
    blk = np.ascontiguousarray(raw).view(np.dtype([("d", np.float16), ("q", np.int8, 32)]))
    return (blk["d"].astype(np.float32)[:, None] * blk["q"].astype(np.float32)).reshape(shape)


def pack_q4_0(x: np.ndarray) -> np.ndarray:
    """18-byte blocks: f16 scale (amax/7) + 16 bytes, low nibble = elements 0..15, high = 16..31; values
    round-half-even(x / f16 scale) clipped to [-8, 7] (byte-identical to the reference converter's quantize_q4_0)."""
    b = _blocks32(x)
    sc = (np.abs(b).max(axis=1) / 7.0).astype(np.float16)
    s32 = sc.astype(np.float32)[:, None]
    q = np.clip(np.where(s32 != 0, np.round(b / np.where(s32 != 0, s32, 1.0)), 0), -8, 7).astype(np.int8)
    u = (q + 8).astype(np.uint8)
    out = np.empty(b.shape[0], dtype=np.dtype([("d", np.float16), ("q", np.uint8, 16)]))
    out["d"], out["q"] = sc, (u[:, :16] & 0xF) | ((u[:, 16:] & 0xF) << 4)
    return out.view(np.uint8)


def unpack_q4_0(raw: np.ndarray, shape) -> np.ndarray:
    blk = np.ascontiguousarray(raw).view(np.dtype([("d", np.float16), ("q", np.uint8, 16)]))
    lo = (blk["q"] & 0xF).astype(np.int32) - 8
    hi = (blk["q"] >> 4).astype(np.int32) - 8
    q = np.concatenate([lo, hi], axis=1).astype(np.float32)
    return (blk["d"].astype(np.float32)[:, None] * q).reshape(shape)


def quantize_weights(W: dict, kind: str):
    """Returns (engine_weights, dequantised_f32_weights): the tensors the reference converter would store as `kind` in
    {"f16","q8_0","q4_0"} -- QUANT_PATTERN, never the depthwise conv (convert_to_gguf.py:237-243), at least 256 elements
    and at least 2 dimensions (:413-419: conv.batch_norm.weight matches the pattern but is 1-D and stays F32), i.e. the
    11 matrices of a layer: FFN 2 x 2, attention q/k/v/pos/out, pointwise conv 1/2."""
    import re
    eng, deq = {}, {}
    for name, a in W.items():
        if re.search(QUANT_PATTERN, name) and "depthwise_conv" not in name and a.size >= 256 and a.ndim >= 2 and a.size % 32 == 0:
            if kind == "f16":
                h = a.astype(np.float16)
                eng[name], deq[name] = (TYPE_F16, h, a.shape), h.astype(np.float32)
            elif kind == "q8_0":
                raw = pack_q8_0(a)
                eng[name], deq[name] = (TYPE_Q8_0, raw, a.shape), unpack_q8_0(raw, a.shape)
            elif kind == "q4_0":
                raw = pack_q4_0(a)
                eng[name], deq[name] = (TYPE_Q4_0, raw, a.shape), unpack_q4_0(raw, a.shape)
            else:
                raise ValueError(kind)
        else:
            eng[name], deq[name] = a, a
    return eng, deq


# ---- diarization side-car: MarbleNet VAD + TitaNet-L, tensor names and layouts of
# scripts/convert_diarize_to_gguf.py (depthwise (k, ch), pointwise (out, in)) ---------------------------------------
VAD_TOPO = [(11, 1, 1, 80, 128, False, True, False), (13, 1, 2, 128, 64, True, True, False), (15, 1, 2, 64, 64, True, True, False),
            (17, 1, 2, 64, 64, True, True, False), (29, 2, 1, 64, 128, False, True, False), (1, 1, 1, 128, 128, False, False, False)]
SPK_TOPO = [(3, 1, 1, 80, 1024, False, True, True), (7, 1, 3, 1024, 1024, True, True, True), (11, 1, 3, 1024, 1024, True, True, True),
            (15, 1, 3, 1024, 1024, True, True, True), (1, 1, 1, 1024, 3072, False, True, True)]
DIAR_N_MELS = 80


def mel_filterbank_n(n_mels: int) -> np.ndarray:
    """Triangular HTK filterbank [n_mels][257] (same construction as mel_filterbank())."""
    def hz2mel(f):
        return 2595.0 * np.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)

    freqs = np.linspace(0.0, SAMPLE_RATE / 2, N_BINS)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(SAMPLE_RATE / 2), n_mels + 2))
    fb = np.zeros((n_mels, N_BINS), dtype=np.float64)
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        fb[m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce))) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


def slaney_filterbank(n_mels: int = 80, fmax: float = SAMPLE_RATE / 2) -> np.ndarray:
    """The filterbank NeMo's featurizer really holds (`featurizer.fb` of a converted checkpoint):
    librosa.filters.mel(sr=16000, n_fft=512, n_mels, fmin=0, fmax, htk=False, norm='slaney') -- linear below 1 kHz,
    logarithmic above, area-normalised triangles.  With it the NeMo-generated fixtures the reference commits
    (tests/diarize/{vad,spk}_ref/{input_audio,mel}.f32) pin the 80-mel front end without any checkpoint."""
    f_sp, min_log_hz, logstep = 200.0 / 3.0, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz2mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    freqs = np.linspace(0.0, SAMPLE_RATE / 2, N_BINS)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(fmax), n_mels + 2))
    fb = np.zeros((n_mels, N_BINS), dtype=np.float64)
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        fb[m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce))) * (2.0 / (hi - lo))
    return fb.astype(np.float32)


def _bn(seed, prefix, C, w):
    w[prefix + ".weight"] = _tensor(seed, prefix + ".weight", (C,), 0.05, center=1.0)
    w[prefix + ".bias"] = _tensor(seed, prefix + ".bias", (C,), 0.05)
    w[prefix + ".running_mean"] = _tensor(seed, prefix + ".running_mean", (C,), 0.05)
    w[prefix + ".running_var"] = _tensor(seed, prefix + ".running_var", (C,), 0.3, center=1.0)


def _jasper(seed, ns, topo, w):
    for b, (kernel, _dil, repeat, cin, cout, residual, separable, has_se) in enumerate(topo):
        pre = f"{ns}.encoder.encoder.{b}"
        for s in range(repeat):
            ci = cin if s == 0 else cout
            dw_i, pw_i, bn_i = 5 * s, 5 * s + 1, 5 * s + 2
            if not separable:
                pw_i, bn_i = 0, 1
            else:
                n = f"{pre}.mconv.{dw_i}.conv.weight"
                w[n] = _tensor(seed, n, (kernel, ci), float(np.sqrt(3.0 / kernel)))
            n = f"{pre}.mconv.{pw_i}.conv.weight"
            w[n] = _tensor(seed, n, (cout, ci), float(np.sqrt(3.0 / ci)))
            _bn(seed, f"{pre}.mconv.{bn_i}", cout, w)
        if residual:
            n = f"{pre}.res.0.0.conv.weight"
            w[n] = _tensor(seed, n, (cout, cin), float(np.sqrt(3.0 / cin)))
            _bn(seed, f"{pre}.res.0.1", cout, w)
        if has_se:
            se = f"{pre}.mconv.{5 * (repeat - 1) + 3}"
            w[se + ".fc.0.weight"] = _tensor(seed, se + ".fc.0.weight", (cout // 8, cout), float(np.sqrt(3.0 / cout)))
            w[se + ".fc.2.weight"] = _tensor(seed, se + ".fc.2.weight", (cout, cout // 8), float(np.sqrt(3.0 / (cout // 8))))


def make_diar_weights(seed: int = 0xD1A12, vad: bool = True, spk: bool = True) -> dict:
    """name -> float32 ndarray for diarize.gguf's two namespaces (random-init MarbleNet 6 blocks / TitaNet-L 5 blocks)."""
    w = {}
    for ns, on in (("vad", vad), ("spk", spk)):
        if not on:
            continue
        w[f"{ns}.preprocessor.featurizer.fb"] = mel_filterbank_n(DIAR_N_MELS)
        w[f"{ns}.preprocessor.featurizer.window"] = hann_window()
    if vad:
        _jasper(seed, "vad", VAD_TOPO, w)
        w["vad.decoder.decoder_layers.0.weight"] = _tensor(seed, "vad.decoder.decoder_layers.0.weight", (2, 128), float(np.sqrt(3.0 / 128)) * 4)
        w["vad.decoder.decoder_layers.0.bias"] = _tensor(seed, "vad.decoder.decoder_layers.0.bias", (2,), 0.05)
    if spk:
        _jasper(seed, "spk", SPK_TOPO, w)
        C, A = 3072, 128
        d = "spk.decoder"
        w[d + "._pooling.attention_layer.0.conv_layer.weight"] = _tensor(seed, d + ".a1w", (A, 3 * C), float(np.sqrt(3.0 / (3 * C))))
        w[d + "._pooling.attention_layer.0.conv_layer.bias"] = _tensor(seed, d + ".a1b", (A,), 0.05)
        _bn(seed, d + "._pooling.attention_layer.0.bn", A, w)
        w[d + "._pooling.attention_layer.2.weight"] = _tensor(seed, d + ".a2w", (C, A), float(np.sqrt(3.0 / A)))
        w[d + "._pooling.attention_layer.2.bias"] = _tensor(seed, d + ".a2b", (C,), 0.05)
        _bn(seed, d + ".emb_layers.0.0", 2 * C, w)
        w[d + ".emb_layers.0.1.weight"] = _tensor(seed, d + ".embw", (192, 2 * C), float(np.sqrt(3.0 / (2 * C))))
        w[d + ".emb_layers.0.1.bias"] = _tensor(seed, d + ".embb", (192,), 0.05)
    return w


# ---- "speech-like" synthetic workload (round 3): audio with a discrete phone inventory ----------------------------------
# A trained RNN-T decides with wide top-2 margins because its encoder maps a DISCRETE inventory (phones) to separated
# clusters.  make_pcm() above is a tone under noise: a random encoder's output for it is a continuum, every decision
# boundary of any joint network is crossed with the same density, and a 1 % activation error flips 1-3 % of the greedy
# decisions whatever the logits are scaled to.  The audio below has that discrete structure: a random sequence of N_PHONES
# "phones" (three formant clusters each, out of 12 candidate centre frequencies, any two phones share at most one)
# separated by near-silence, with onsets at arbitrary sample positions.
N_PHONES = 16
_FORMANTS = np.geomspace(260.0, 6200.0, 12)
# 16 triples over 12 points, pairwise intersection <= 1 (rows/columns/diagonals of a 4x3 torus arrangement + shifts)
_PHONE_FORMANTS = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (9, 10, 11), (0, 3, 6), (1, 4, 7), (2, 5, 8), (0, 4, 8),
                   (1, 5, 6), (2, 3, 7), (0, 5, 9), (1, 3, 10), (2, 4, 11), (0, 7, 10), (1, 8, 11), (2, 6, 9)]
PHONE_TOKEN_STRIDE, PHONE_TOKEN_BASE = 61, 37


def phone_token(k: int) -> int:
    """vocabulary id of phone k (spread over the 1024 tokens so the arg-max scan is exercised across the row range)"""
    return (PHONE_TOKEN_BASE + PHONE_TOKEN_STRIDE * k) % BLANK


def speech_events(stream: int, seconds: float, seed: int = 0x5BEEC400):
    """[(phone, onset_sample, end_sample)]: phones of 240-480 ms, gaps of 80-320 ms, never the same phone twice in a row."""
    n = int(round(seconds * SAMPLE_RATE))
    u = uniform01(seed + 7919 * stream, 3 * (int(seconds * 4) + 8), 0)
    ev, pos, prev, i = [], int(0.1 * SAMPLE_RATE + u[0] * 0.2 * SAMPLE_RATE), -1, 1
    while True:
        dur = int((0.24 + 0.24 * u[i]) * SAMPLE_RATE)
        gap = int((0.08 + 0.24 * u[i + 1]) * SAMPLE_RATE)
        k = int(u[i + 2] * (N_PHONES - 1))
        if k >= prev >= 0:
            k += 1                      # uniform over the phones other than the previous one
        if pos + dur > n:
            break
        ev.append((k, pos, pos + dur))
        pos, prev, i = pos + dur + gap, k, i + 3
    return ev


def make_speech_pcm(stream: int, seconds: float, seed: int = 0x5BEEC400):
    """int16 mono 16 kHz + its event list.  Phone k = 3 formant clusters (3 partials each, fixed phases) at amplitude 0.25
    with 10 ms raised-cosine edges; background = white noise of sigma 0.004 everywhere (so log-mel never sits at its floor),
    plus noise of sigma 0.01 inside phones."""
    n = int(round(seconds * SAMPLE_RATE))
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    u1 = uniform01(seed + 7919 * stream + 1, n, 0)
    u2 = uniform01(seed + 7919 * stream + 1, n, n)
    noise = np.sqrt(-2.0 * np.log(np.maximum(u1, 2.0 ** -53))) * np.cos(2.0 * np.pi * u2)
    x = 0.004 * noise
    ev = speech_events(stream, seconds, seed)
    edge = int(0.010 * SAMPLE_RATE)
    for k, a, b in ev:
        seg = np.zeros(b - a)
        tt = t[a:b]
        for j, fi in enumerate(_PHONE_FORMANTS[k]):
            fc = _FORMANTS[fi]
            for m, det in enumerate((0.965, 1.0, 1.04)):
                seg += np.sin(2.0 * np.pi * fc * det * tt + 0.7 * (3 * j + m) + 0.37 * k) / 9.0
        env = np.ones(b - a)
        r = 0.5 - 0.5 * np.cos(np.pi * np.arange(edge) / edge)
        env[:edge], env[-edge:] = r, r[::-1]
        x[a:b] += env * (0.75 * seg + 0.01 * noise[a:b])
    return np.round(np.clip(x, -1.0, 1.0) * 32767.0).astype(np.int16), ev


# ---- the "speech" checkpoint: a joint with the margins of a TRAINED one -------------------------------------------------------
# make_weights(..., margins="speech") keeps every tensor of the default checkpoint's encoder (same seed) except that the
# output matrices of the residual branches are scaled by SPEECH_RESIDUAL_SCALE (scale_residual_branches), and replaces the
# decoder / joint by a structured one whose acoustic read-out was FITTED (ridge regression, tests/golden/gen_speech_joint.py)
# to read the phone of make_speech_pcm() audio off that frozen synthetic encoder:
#   joint hidden units [0, 17): phone detectors -- row k of joint.enc is the fitted direction w_k (target: 1 on frames of
#       phone k, 0 elsewhere; unit 16 = "silence"), hidden = relu(w_k . e + b_k - SPEECH_DETECT_FLOOR); the output layer
#       maps detector k to token k (detector 16 to blank, + SPEECH_BLANK_BIAS) with SPEECH_LOGIT_SCALE logits per target unit;
#   joint hidden units [17, SPEECH_NA): dense random rows as in the default checkpoint; only the never-emitted tokens' output
#       rows read them (small random weights under a negative bias);
#   joint hidden units [SPEECH_NA, 640): refractory -- phone k owns 8 units (its "code"); embed[token k] drives them through
#       the two LSTM layers (forget gate ~0.003: the state is the LAST token), joint.pred = -SPEECH_PRED_GAIN on those units,
#       so after emitting k its own logit drops by SPEECH_SUPPRESS target units: one token per phone, as a trained
#       prediction network would do it.
# The LSTM / joint.pred matrices keep a dense random part, so their arithmetic is exercised in full.
SPEECH_NA = 512
SPEECH_CODE = (JOINT - SPEECH_NA) // N_PHONES          # 8 units per phone
SPEECH_LOGIT_SCALE = 8.0
SPEECH_BLANK_BIAS = 0.5
SPEECH_DETECT_FLOOR = 0.0
SPEECH_SUPPRESS = 1.0
SPEECH_PRED_GAIN = 3.0
SPEECH_EMBED = 3.0
SPEECH_FORGET_BIAS = -6.0
SPEECH_DEC_NOISE = 0.05     # amplitude of the dense random part of the LSTM / joint.pred matrices (x sqrt(3/fan_in)); the embedding gain
                            # and the prediction gain multiply it, so it is 5x smaller than DEC_NOISE of the default checkpoint
SPEECH_READOUT_FILE = "speech_readout_v1.npz"


def load_speech_readout():
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "data" / SPEECH_READOUT_FILE)
    return z["w"].astype(np.float32), z["b"].astype(np.float32)


def speech_decoder_tensors(seed: int, readout=None) -> dict:
    """decoder + joint tensors of the speech checkpoint.  readout = (w [N_PHONES + 1][1024], b [N_PHONES + 1]) in target
    units (row N_PHONES = silence); default: the committed fit."""
    w_fit, b_fit = load_speech_readout() if readout is None else readout
    ND = N_PHONES + 1
    assert w_fit.shape == (ND, D_MODEL) and b_fit.shape == (ND,)
    NA, A = SPEECH_NA, np.float32(SPEECH_LOGIT_SCALE)
    out = {}
    code = np.zeros((N_PHONES, JOINT), np.float32)
    for k in range(N_PHONES):
        code[k, NA + k * SPEECH_CODE:NA + (k + 1) * SPEECH_CODE] = 1.0
    toks = [phone_token(k) for k in range(N_PHONES)]
    # joint.enc: detector rows fitted, the other acoustic rows as in the default checkpoint, refractory rows 20x smaller with bias 1
    we = _tensor(seed, "joint.enc.weight", (JOINT, D_MODEL), float(np.sqrt(3.0 / D_MODEL)))
    be = _tensor(seed, "joint.enc.bias", (JOINT,), 0.05)
    we[:ND] = w_fit
    be[:ND] = b_fit - np.float32(SPEECH_DETECT_FLOOR)
    we[NA:] *= np.float32(0.05)
    be[NA:] += np.float32(1.0)
    out["joint.enc.weight"], out["joint.enc.bias"] = we, be
    # joint.pred: dense random part (not on the detector units) + -gain * I on the refractory units
    wp = _tensor(seed, "joint.pred.weight", (JOINT, HIDDEN), float(np.sqrt(3.0 / HIDDEN)) * SPEECH_DEC_NOISE)
    wp[:ND] = 0.0
    wp[NA:, NA:] -= np.float32(SPEECH_PRED_GAIN) * np.eye(JOINT - NA, dtype=np.float32)
    bp = _tensor(seed, "joint.pred.bias", (JOINT,), 0.05)
    bp[:ND] = 0.0
    out["joint.pred.weight"], out["joint.pred.bias"] = wp, bp
    # output layer: the never-emitted tokens read the random acoustic units (small weights, negative bias); phone k reads
    # detector k and its refractory code, blank reads the silence detector
    wo = (0.25 * _jout_random(seed)).astype(np.float32)
    wo[:, :ND] = 0.0
    bo = _tensor(seed, "joint.joint_net.2.bias", (VOCAB,), 0.05) - np.float32(1.5) * A
    gamma = np.float32(SPEECH_SUPPRESS) * A / np.float32(SPEECH_CODE)
    for k, v in enumerate(toks):
        wo[v] = 0.0
        wo[v, k] = A
        wo[v, NA:] = gamma * code[k, NA:]
        bo[v] = A * np.float32(SPEECH_DETECT_FLOOR) - gamma * np.float32(SPEECH_CODE)     # un-suppressed: the refractory units add gamma * 8 back
    wo[BLANK] = 0.0
    wo[BLANK, N_PHONES] = A
    bo[BLANK] = A * np.float32(SPEECH_DETECT_FLOOR + SPEECH_BLANK_BIAS)
    out["joint.joint_net.2.weight"], out["joint.joint_net.2.bias"] = wo, bo
    # embedding: the token's code on the refractory dims, small random elsewhere; blank = padding row
    em = _tensor(seed, "decoder.prediction.embed.weight", (VOCAB, HIDDEN), 0.05)
    for k, v in enumerate(toks):
        em[v] += np.float32(SPEECH_EMBED) * code[k]
    em[BLANK] = 0.0
    out["decoder.prediction.embed.weight"] = em
    # LSTM x 2: g-gate = identity on the input, input / output gates open, forget gate nearly closed
    for l in (0, 1):
        p = "decoder.prediction.dec_rnn.lstm."
        for kind in ("ih", "hh"):
            name = p + f"weight_{kind}_l{l}"
            w = _tensor(seed, name, (4 * HIDDEN, HIDDEN), float(np.sqrt(3.0 / HIDDEN)) * SPEECH_DEC_NOISE)
            if kind == "ih":
                w[2 * HIDDEN:3 * HIDDEN] += np.eye(HIDDEN, dtype=np.float32)
            out[name] = w
        for kind in ("ih", "hh"):
            name = p + f"bias_{kind}_l{l}"
            b = _tensor(seed, name, (4 * HIDDEN,), 0.01)     # (the prediction gain multiplies what the g-gate bias leaks into h)
            if kind == "ih":
                b[0:HIDDEN] += np.float32(GATE_OPEN)
                b[HIDDEN:2 * HIDDEN] += np.float32(SPEECH_FORGET_BIAS)
                b[3 * HIDDEN:4 * HIDDEN] += np.float32(GATE_OPEN)
            out[name] = b
    return out


RESIDUAL_BRANCH_OUT = ("feed_forward1.linear2.weight", "self_attn.linear_out.weight", "conv.pointwise_conv2.weight",
                       "feed_forward2.linear2.weight")
SPEECH_RESIDUAL_SCALE = 0.1


def scale_residual_branches(W: dict, alpha: float) -> dict:
    """multiplies the output matrix of every residual branch (FFN x 2, attention, conv module) of every layer by alpha, in
    place.  With variance-1/fan_in matrices every branch adds as much as the stream carries, each layer re-mixes the frame
    with its 70-frame context at full strength and after 24 layers nothing of the frame's own content is linearly readable
    (measured, held-out frame accuracy of a ridge read-out of the phone at 24 layers: alpha 1: 30 %, 0.5: 34 %, 0.3: 92 %,
    0.2: 98 %, 0.1: 99.6 %; gpurun_out/speech/speech_alpha_scan.json of round 3).  A trained
    network's branches are small updates of the stream; alpha < 1 gives the synthetic encoder that property."""
    a = np.float32(alpha)
    for name in W:
        if name.startswith("encoder.layers.") and name.endswith(RESIDUAL_BRANCH_OUT):
            W[name] = (W[name] * a).astype(np.float32)
    return W


def apply_speech_decoder(W: dict, seed: int = 0xC0FFEE, readout=None) -> dict:
    """a copy of the checkpoint dict with the decoder / joint tensors of the speech checkpoint (encoder tensors shared)"""
    W2 = dict(W)
    W2.update(speech_decoder_tensors(seed, readout))
    return W2
