"""Word error rate of the engine's transcript against a labelled reference -- the accuracy half of BASELINE.json's metric
("RTFx ...; WER vs ref").  No checkpoint or transcript ships with the reference, so the parity suite
measures token agreement with the oracle instead; with a real model this tool gives the WER:

    python -m nemotron_asr_amd.wer --gguf nemotron-speech-streaming-0.6b.gguf --pcm audio.pcm --ref transcript.txt [--right-context 13]
    NEMOTRON_GGUF=... python nemotron-asr.cpp_amd/wer.py --pcm a.pcm --ref a.txt

It runs `bin/nemotron-asr-amd` (the CLI of the reference, same argv) and scores its stdout.  Normalisation = lower case, punctuation
stripped, whitespace collapsed (what the reference's scripts/compare_transcripts.py-style checks and NeMo's WER use by default).
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent


def normalise(text: str) -> list:
    text = text.lower()
    text = re.sub(r"[^\w\s']", " ", text, flags=re.UNICODE)       # keep letters, digits, apostrophes
    return text.split()


def edit_counts(ref: list, hyp: list):
    """(substitutions, deletions, insertions) of a minimal word alignment (Levenshtein, ties resolved towards substitutions)"""
    n, m = len(ref), len(hyp)
    cost = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        cost[i][0] = i
    for j in range(1, m + 1):
        cost[0][j] = j
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            cost[i][j] = min(cost[i - 1][j - 1] + (ref[i - 1] != hyp[j - 1]), cost[i - 1][j] + 1, cost[i][j - 1] + 1)
    i, j, sub, dele, ins = n, m, 0, 0, 0
    while i > 0 or j > 0:
        if i > 0 and j > 0 and cost[i][j] == cost[i - 1][j - 1] + (ref[i - 1] != hyp[j - 1]):
            sub += ref[i - 1] != hyp[j - 1]
            i, j = i - 1, j - 1
        elif i > 0 and cost[i][j] == cost[i - 1][j] + 1:
            dele += 1
            i -= 1
        else:
            ins += 1
            j -= 1
    return sub, dele, ins


def wer(ref_text: str, hyp_text: str) -> dict:
    ref, hyp = normalise(ref_text), normalise(hyp_text)
    sub, dele, ins = edit_counts(ref, hyp)
    n = max(len(ref), 1)
    return dict(wer=(sub + dele + ins) / n, substitutions=sub, deletions=dele, insertions=ins, ref_words=len(ref), hyp_words=len(hyp))


def transcribe(gguf: str, pcm: str, right_context: int = 0, extra=()) -> str:
    cli = HERE / "bin" / "nemotron-asr-amd"
    if not cli.exists():
        raise RuntimeError(f"{cli} not built: python __graft_entry__.py")
    r = subprocess.run([str(cli), gguf, pcm, "80", str(right_context), *extra], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return r.stdout.splitlines()[0] if r.stdout else ""


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gguf", default=os.environ.get("NEMOTRON_GGUF"))
    ap.add_argument("--pcm", required=True, help="raw s16le 16 kHz mono")
    ap.add_argument("--ref", required=True, help="text file with the reference transcript")
    ap.add_argument("--right-context", type=int, default=0, choices=[0, 1, 6, 13])
    ap.add_argument("--read-chunks", type=int, default=64, help="chunks per read (file mode: same transcript, faster)")
    args = ap.parse_args(argv)
    if not args.gguf:
        ap.error("--gguf or $NEMOTRON_GGUF is required")
    hyp = transcribe(args.gguf, args.pcm, args.right_context, ["--read-chunks", str(args.read_chunks)])
    res = wer(Path(args.ref).read_text(), hyp)
    print(f"WER {100 * res['wer']:.2f} %  ({res['substitutions']} sub, {res['deletions']} del, {res['insertions']} ins over {res['ref_words']} words)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
