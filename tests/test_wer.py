"""The WER scorer (nemotron-asr.cpp_amd/wer.py): BASELINE.json's metric names "WER vs ref"; no labelled audio or checkpoint is
in the tree, so only the scoring itself can be pinned here."""
from nemotron_asr_amd import wer as W


def test_wer_counts():
    assert W.wer("the cat sat on the mat", "the cat sat on the mat")["wer"] == 0.0
    r = W.wer("the cat sat on the mat", "the cat sit on mat")
    assert (r["substitutions"], r["deletions"], r["insertions"]) == (1, 1, 0) and abs(r["wer"] - 2 / 6) < 1e-12
    r = W.wer("hello world", "hello brave new world")
    assert (r["substitutions"], r["deletions"], r["insertions"]) == (0, 0, 2) and r["wer"] == 1.0
    r = W.wer("a b c d", "")
    assert r["deletions"] == 4 and r["wer"] == 1.0
    assert W.wer("", "x y")["insertions"] == 2


def test_normalisation():
    assert W.normalise(" Hello,   WORLD! It's 9 o'clock.") == ["hello", "world", "it's", "9", "o'clock"]
    # the engine's transcripts start with a space (U+2581 rule, src/nemo-ggml.cpp:1562-1581): irrelevant to the score
    assert W.wer("Good morning.", " good morning")["wer"] == 0.0
