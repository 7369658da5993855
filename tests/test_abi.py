"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the
header declares (no compute calls without a GPU), and fails loudly when no GPU is visible."""
import re
from pathlib import Path

import numpy as np
import pytest

from nemotron_asr_amd import capi, synth

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "nemotron_asr_amd.h").read_text()
    declared = set(re.findall(r"\b(nasr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS)
    assert capi.check_exports()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    W = {"preprocessor.featurizer.window": synth.hann_window()}
    with pytest.raises(capi.NasrError, match="HIP|device|hip"):
        capi.Engine(W, n_layers=1)


def test_product_does_not_import_oracle():
    """Only smoke.py (the checker entry) may reference the oracle; the product path never does."""
    pkg = ROOT / "nemotron-asr.cpp_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        if f.name == "smoke.py":
            continue
        txt = f.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt and "nasr_oracle" not in txt, f
