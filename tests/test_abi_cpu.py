"""CPU-side checks: the C-ABI library loads and exports every symbol include/pvx.h declares, host
logic that needs no device, and the loud failure without a GPU.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from .conftest import ROOT, golden_names, load_golden


def header_functions():
    src = open(os.path.join(ROOT, "include", "pvx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pvx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pypevoc_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libpvx_hip.so lacks %s declared in include/pvx.h" % n
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    # the other direction: the library exports NOTHING but the declared functions (-fvisibility=hidden, default visibility on
    # the header's declarations, csrc/pvx.map for the weak template instantiations of the standard headers) -- also the
    # witness build of the bit-identity tests
    import os
    import subprocess
    for path in (_lib.LIB_PATH, os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpvx_witness.so")):
        if not os.path.exists(path):
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
        assert exported == names, (path, sorted(set(exported) ^ set(names)))


def test_nframes_and_synth_len_host_logic():
    from pypevoc_amd import _lib
    lib = _lib.load()
    for name in golden_names():
        g = load_golden(name)
        assert lib.pvx_nframes(len(g["x"]), g["nfft"], g["hop"]) == g["nframes"]
        for k in g:
            if k.startswith("w_hop"):
                h = int(k[5:])
                maxend = int((g["part_start"].astype(np.int64) + g["part_len"] - 1).max())
                assert lib.pvx_synth_len(maxend, g["nfft"], g["hop"], h, 1.0) == len(g[k])
    assert lib.pvx_nframes(1024, 1024, 512) == 0
    assert lib.pvx_nframes(1025, 1024, 512) == 1
    # frame counts of the BASELINE configs
    assert lib.pvx_nframes(26460000, 2048, 512) == 51676
    assert lib.pvx_nframes(1440000, 2048, 512) == 2809


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pypevoc_amd
    p = pypevoc_amd.PV(np.zeros(4096), 44100, nfft=1024, hop=512, npks=4, progress=False)
    with pytest.raises(pypevoc_amd.PvxError) as e:
        p.run_pv()
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(pypevoc_amd.PvxError):
        pypevoc_amd.PeakFinder(np.arange(10.0))
    with pytest.raises(pypevoc_amd.PvxError) as e:
        pypevoc_amd.PVMany(44100, nfft=1024, hop=512, npks=4).run([np.zeros(4096), np.zeros(9000)])
    assert "no CPU fallback" in str(e.value)


def test_batch_item_layout_is_the_headers():
    """pvx_batch_item (include/pvx.h): two 8-byte inputs, seven pointers, nframes, device, reserved -- 88 bytes, the ctypes
    mirror field for field."""
    import ctypes
    from pypevoc_amd import _lib
    assert ctypes.sizeof(_lib.BatchItem) == 88
    names = [f[0] for f in _lib.BatchItem._fields_]
    src = open(os.path.join(ROOT, "include", "pvx.h")).read()
    body = src[src.index("typedef struct pvx_batch_item {"):src.index("} pvx_batch_item;")]
    import re
    decl = re.findall(r"[*\s]([a-z_0-9]+)\s*[,;]", body)
    assert decl == names, (decl, names)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pypevoc_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "pvoracle" not in txt and "oracle/" not in txt and "import oracle" not in txt, fn


def test_pv_constructor_constants_match_reference_formulas():
    """PV.__init__ (PVAnalysis.py:84-121) is host arithmetic; check the values the kernels consume."""
    import pypevoc_amd
    p = pypevoc_amd.PV(np.zeros(5000), 44100, nfft=2048, npks=3, progress=False)
    assert p.hop == 1024 and p.nfft2 == 1024 and p.nsamp == 5000
    win = np.hanning(2048)
    assert np.array_equal(p.win, win)
    assert p.wfact == np.sqrt(sum(win ** 2) * 2048) / 2.0
    assert p.fstep == 44100.0 / 2048 and p.dt == 1024 / 44100.0
    # round-half-even wrap table for hop = nfft/2: 0,0,1,2,2,2,3,4
    assert list((p.wfbin[:8] / (2 * np.pi)).round().astype(int)) == [0, 0, 1, 2, 2, 2, 3, 4]
    assert p.oldfft.shape == (1024,) and not p.oldfft.any()


def test_shard_range_partitions():
    from pypevoc_amd.batch import shard_range
    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_frame_descriptors_match_reference():
    """Host-side consumers of the (F, K) arrays, no GPU involved: calc_f0, fundamental_*,
    partial_* and calc_harmonic_power (incl. its row-indexing quirk, PV.py:278) against values captured
    from the reference (tests/golden/make_golden_harmonic.py, D1)."""
    import pypevoc_amd
    g = np.load(os.path.join(ROOT, "tests", "golden", "D1_descriptors.npz"))
    p = pypevoc_amd.PV(np.zeros(4096), 44100, nfft=2048, hop=512, npks=8, progress=False)
    p.f, p.mag, p.totalmag = g["f"], g["mag"], list(g["totalmag"])
    f0 = p.calc_f0()
    assert np.array_equal(f0, g["f0"]) and np.array_equal(p.fundamental_idx, g["fundamental_idx"])
    assert np.array_equal(p.fundamental_frequency, g["fundamental_frequency"])
    assert np.array_equal(p.fundamental_magnitude, g["fundamental_magnitude"])
    assert np.allclose(p.partial_sum_magnitude, g["partial_sum_magnitude"], rtol=1e-14, atol=0)
    assert np.allclose(p.partial_magnitude_ratio, g["partial_magnitude_ratio"], rtol=1e-14, atol=0)
    f0b = p.calc_f0(fmin=300, fmax=2000, thr=0.3)
    assert np.array_equal(f0b, g["f0_b"]) and np.array_equal(p.fundamental_idx, g["fundamental_idx_b"])
    p.calc_harmonic_power()
    assert np.array_equal(p.nharmonics, g["nharmonics"])
    assert np.allclose(p.hpower, g["hpower"], rtol=1e-13, atol=0)
    p.calc_harmonic_power(f_threshold=0.002)
    assert np.array_equal(p.nharmonics, g["nharmonics_b"])
    assert np.allclose(p.hpower, g["hpower_b"], rtol=1e-13, atol=0)


def test_generated_isa_has_no_asm_to_dpp_hazard():
    """The packed complex primitives of pvx_cplx.h are inline asm; the compiler's hazard recogniser does
    not see inside them, so a DPP move reading one of their results too early would silently read a stale
    register.  tools/check_dpp_hazard.py compiles the fused kernels to gfx950 assembly and scans for it."""
    import shutil
    import subprocess
    import sys
    if shutil.which(os.environ.get("HIPCC", "hipcc")) is None:
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazard.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 asm -> DPP hazard candidates" in r.stdout


def test_every_environment_switch_of_the_library_is_documented():
    """INTEGRATION.md's table of environment switches names every PVX_* variable the native sources read."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for path in glob.glob(os.path.join(root, "pypevoc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "pypevoc_amd", "csrc", "*.h")):
        read |= set(re.findall(r'getenv\("(PVX_[A-Z0-9_]+)"\)', open(path).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(v for v in read if v not in doc)
    assert read and not missing, missing


def test_design_document_stays_a_design_document():
    """DESIGN.md is the design as it stands in at most 40 KB (the rounds' narratives live in HISTORY.md and profiles/), and its
    numbers are the generated block: `tools/design_numbers.py` rewrites what sits between the markers from the committed profiles."""
    path = os.path.join(ROOT, "DESIGN.md")
    raw = open(path, "rb").read()
    assert len(raw) <= 40 * 1024, len(raw)
    text = raw.decode()
    assert text.count("<!-- NUMBERS:BEGIN") == 1 and text.count("<!-- NUMBERS:END -->") == 1
    assert text.index("<!-- NUMBERS:BEGIN") < text.index("<!-- NUMBERS:END -->")
