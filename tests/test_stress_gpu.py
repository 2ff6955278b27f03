"""-m gpu: a few seconds of the randomised parity stress tools (tools/stress_*.py): random sizes,
label counts, kernels, shared / per-edge positions and tie-heavy integer data against the oracle
(TRW-S) and the reference's own QPBO library where it travelled (QPBO)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed", [("stress_trws.py", 11), ("stress_rd.py", 12), ("stress_improve.py", 13),
                                       ("stress_certificate.py", 14)])
def test_randomised_parity(tool, seed, hip, oracle):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "8", str(seed)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
