"""The no-GPU developer tools keep working: tools/isa_blocks.py cross-compiles a kernel file for gfx950 and reads its ISA."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/opt/rocm/bin/hipcc").exists() and shutil.which("hipcc") is None, reason="no hipcc")
def test_isa_blocks_reads_the_stereo_kernel():
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "isa_blocks.py"), str(ROOT / "orb_slam2_annotate_amd" / "csrc" / "k_match.hip"),
                        "k_stereo_match_batch", "--mem"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout
    assert "k_stereo_match_batch" in out.splitlines()[0]
    assert "static: valu" in out and "Occupancy" in out and "ScratchSize: 0" in out
    # the SAD block is one basic block of several hundred VALU instructions behind a wave-uniform liveness test
    import re
    big = [int(m.group(1)) for m in re.finditer(r"valu\s+(\d+)\s+salu", out)]
    assert max(big) > 300
    assert "global_load" in out  # --mem lists the loads and the vmcnt waits in program order
