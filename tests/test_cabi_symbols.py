"""The C-ABI library must load on a CPU-only box and export every symbol include/orbfe.h declares;
without a GPU every compute entry point must fail loudly (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "orbfe.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from orb_slam2_annotate_amd import _lib
    L = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"liborbfe.so does not export {n}"
    assert sorted(_lib.EXPORTS) == names, "orb_slam2_annotate_amd/_lib.py EXPORTS out of sync with include/orbfe.h"


def test_structs_match_reference_layouts():
    from orb_slam2_annotate_amd import _lib
    assert _lib.KP_DTYPE.itemsize == 28  # cv::KeyPoint: 5 floats + 2 ints
    assert [_lib.KP_DTYPE.fields[n][1] for n in _lib.KP_DTYPE.names] == [0, 4, 8, 12, 16, 20, 24]
    assert C.sizeof(_lib.FeatVecC) == 32


def test_no_cpu_fallback_without_gpu():
    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import _lib
    if _lib.load().orbfe_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(amd.OrbfeError) as ei:
        amd.ORBextractor(1000, 1.2, 8, 20, 7)
    assert ei.value.code == _lib.ERR_HIP
    with pytest.raises(amd.OrbfeError):
        amd.ORBmatcher.DescriptorDistance(np.zeros(32, np.uint8), np.zeros(32, np.uint8))
    with pytest.raises(amd.OrbfeError):
        amd.resize_linear(np.zeros((10, 10), np.uint8), 5, 5)


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = ROOT / "orb_slam2_annotate_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.h")) + \
            [ROOT / "include" / "orbfe.h"]:
        if "build" in p.parts:
            continue
        text = p.read_text(errors="replace")
        assert "orb_oracle" not in text and "oracle_lib" not in text and "liborb_oracle" not in text, p
