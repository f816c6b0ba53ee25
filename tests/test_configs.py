"""BASELINE configs[1], [2] at a size that runs in seconds: primary rays and
per-bounce path-tracing batches into the Sponza-class stand-in, every batch sampled against the
oracle at the parity bar.  The full-size numbers are in profiles/r01_configs_c2c3.json."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c2_c3_batches_within_bar():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "configs_c2c3.py"), "--tris", "60000",
                        "--width", "320", "--height", "180", "--spp", "2", "--depth", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    c2, c3 = recs[0], recs[1]
    p = c2["parity_sample_vs_oracle"]
    assert p["hit_miss_equal"] and p["same_prim_fraction"] >= 0.9999 and p["max_rel_t_err"] <= 1e-5
    checked = 0
    for b in c3["bounces"]:
        if "closest_parity" in b:
            q = b["closest_parity"]
            assert q["hit_miss_equal"] and q["same_prim_fraction"] >= 0.9999 and q["max_rel_t_err"] <= 1e-5
            checked += 1
        if "shadow_parity" in b:
            assert b["shadow_parity"]["any_hit_equal"]
            checked += 1
    assert checked >= 4 and len(c3["bounces"]) >= 3
