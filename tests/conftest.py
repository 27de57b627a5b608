"""pytest configuration: the `gpu` marker and shared golden-fixture helpers."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names(prefix="G", exclude=("G8_", "G9_chirp_input", "G15_")):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz"))):
        n = os.path.basename(p)[:-4]
        if not any(n.startswith(e) for e in exclude):
            out.append(n)
    return out


def load_golden(name):
    """Returns the fixture as a dict with `x` as the float64 signal the reference analysed."""
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if name.startswith("G9_chirp_n"):
        src = np.load(os.path.join(GOLDEN, "G9_chirp_input.npz"))
        g["x"] = src["x"][: int(g["x_len"])].astype(np.float64)
    elif "x_scale" in g:                      # int16 WAV payload (examples/WavResynth.py:18)
        g["x_raw"] = g["x"]
        g["x"] = g["x"] / float(np.iinfo(g["x"].dtype).max)
    else:
        g["x"] = g["x"].astype(np.float64)
    for k in ("nfft", "hop", "npks", "nframes"):
        g[k] = int(g[k])
    g["sr"] = float(g["sr"])
    if "pkthresh" in g:
        g["pkthresh"] = float(g["pkthresh"])
    if "fmin" in g:
        g["fmin"] = float(g["fmin"])
    return g


@pytest.fixture(scope="session")
def oracle():
    from oracle import pvoracle
    pvoracle.build()
    return pvoracle


@pytest.fixture
def witness():
    """fft modes 1 and 3 (k_fused.hip, k_fused_ring.hip: the witness kernels of the bit-identity tests) are not in the product
    library: for the duration of a test every call goes through tests/libpvx_witness.so, the product's objects plus those two
    (`make -C pypevoc_amd/csrc witness`, which __graft_entry__.build() runs)."""
    from pypevoc_amd import _lib
    path = os.path.join(ROOT, "tests", "libpvx_witness.so")
    if not os.path.exists(path):
        pytest.fail("tests/libpvx_witness.so is missing: make -C pypevoc_amd/csrc witness")
    old = _lib.swap_library(path)
    _lib.init()
    try:
        yield path
    finally:
        _lib.swap_library(old)
        _lib.init()
