"""pypevoc_amd -- MI355X-native phase-vocoder analysis / resynthesis (drop-in for the
PV.run_pv -> toSinSum -> SinSum.synth path of goiosunsw/PyPeVoc).

    from pypevoc_amd import PV
    p = PV(sig, sr, nfft=2048, npks=3); p.run_pv(); w = p.toSinSum().synth(sr, p.hop)

Everything heavy runs in libpvx_hip.so (hand-written HIP kernels for gfx950 + rocFFT), loaded with
ctypes; see include/pvx.h, DESIGN.md and INTEGRATION.md.
"""
from .PVAnalysis import PV, SinSum, RegPartial, PVHarmonic  # noqa: F401  (pypevoc/__init__.py:1 exports PV, SinSum)
from .PeakFinder import PeakFinder  # noqa: F401
from .batch import PVBatch, PVMany  # noqa: F401
from ._lib import PvxError  # noqa: F401

__all__ = ["PV", "PVHarmonic", "SinSum", "RegPartial", "PeakFinder", "PVBatch", "PVMany", "PvxError"]
