"""examples/trace_obj.c drives the whole C-ABI from plain C (the header must be valid C99, not only
C++): OBJ -> accel -> closest hits -> normals on the device -> BMP.  CPU: it compiles with
-std=c99 -pedantic and fails loudly without a device.  GPU: it renders the terrain fixture."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def build(tmp_path):
    exe = str(tmp_path / "trace_obj")
    lib_dir = os.path.join(ROOT, "yart_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "trace_obj.c"), "-o", exe, "-L" + lib_dir, "-lyart_amd",
                        "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_compiles_as_c99_and_needs_a_device(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present; covered by the gpu test")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "obj") + os.sep, "terrain.obj", str(tmp_path / "o.bmp")],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_example_renders_the_terrain(tmp_path):
    exe = build(tmp_path)
    out = tmp_path / "o.bmp"
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "obj") + os.sep, "terrain.obj", str(out), "320", "200"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = out.read_bytes()
    w, h = struct.unpack_from("<ii", raw, 18)
    assert (w, h) == (320, 200) and len(raw) == 54 + 960 * 200
    px = np.frombuffer(raw[54:], np.uint8)
    assert (px > 0).mean() > 0.2                    # the terrain fills a good part of the frame
    assert "2714 triangles, 1 spheres, 64000 rays" in r.stdout
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "example_terrain.bmp"), "wb") as f:
            f.write(raw)
