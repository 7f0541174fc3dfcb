"""A bounded run of the randomised GPU-vs-oracle sweep (tools/fuzz_parity.py): random geometry (BlockSize 256..8192, 1-3
channels, 22-96 kHz), rate-control mode and parameter, signal shape (synthetic mix, white noise, DC offset, impulse
trains, silence, clipping, beyond full scale), loudness down to digital silence, and calls of varying length.  Every
configuration must be bit-exact in encoded bytes, sizes, WindowCtrl, BlockComplexity and decoded PCM."""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [20261003, 7])
def test_randomised_parity_sweep(seed):
    import fuzz_parity
    n, nblk = fuzz_parity.run(budget=30.0, seed=seed)
    assert n >= 10 and nblk >= 200, f"the sweep covered too little in its time budget: {n} configurations, {nblk} blocks"
