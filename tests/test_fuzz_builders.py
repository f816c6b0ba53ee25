"""GPU: a short run of tools/fuzz_device_build.py -- random scenes of many sizes and shapes (soups, architectural,
grid-aligned with ties everywhere, clustered with far outliers), device-built against host-built accels: the
reference's arrays byte for byte, the reference rules identical, the default rule at the parity bar (exact ties and
the stated ill-conditioned cases aside)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_and_host_builders_agree_on_random_scenes():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_device_build.py"), "14", "5"],
                         capture_output=True, text=True, timeout=280)
    lines = [l for l in out.stdout.splitlines() if l.startswith("round") or l.startswith("mismatches")]
    assert out.returncode == 0, "\n".join(lines[-20:]) + out.stderr[-2000:]
    assert lines and lines[-1] == "mismatches: 0"
    assert sum(" ok" in l for l in lines) >= 8
