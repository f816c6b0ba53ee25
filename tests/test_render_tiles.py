"""BASELINE configs[4] / SURVEY.md 8e on the GPU box: image tiles dealt to ranks, scene replicated, every
rank's share rendered by the device wavefront driver, framebuffer gathered with one exchange -- every pixel
identical to the unsharded render, and every rank's pixels checked against the per-pixel CPU loop over the
oracle (a seeded sample re-rendered by oracle/diffuse_ref.py; its closest and shadow rays re-traced at the
parity bar).  Two ranks share the box's one GPU here (gloo); with a GPU per rank the same script gathers
with yart_frame_gather over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SCRIPT = os.path.join(ROOT, "tools", "render_tiles.py")
ARGS = ["--tris", "30000", "--width", "200", "--height", "120", "--spp", "2", "--depth", "4", "--slots", "8192", "--verify"]


def _last_json(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_single_rank_frame():
    r = subprocess.run([sys.executable, SCRIPT] + ARGS, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["n_ranks"] == 1 and out["sharded_equals_single_rank_bitwise"] and out["lit_fraction"] > 0.01
    assert out["oracle_check"]["all_ranks_within_parity_bar"] and out["oracle_check"]["pixels_per_rank_rerendered_on_cpu"] >= 256


def test_two_ranks_gather_equals_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), SCRIPT] + ARGS,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["n_ranks"] == 2 and out["sharded_equals_single_rank_bitwise"]
    assert out["oracle_check"]["all_ranks_within_parity_bar"] and out["oracle_check"]["pixels_differing"] <= 2
