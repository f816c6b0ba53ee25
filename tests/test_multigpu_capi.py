"""GPU: image mode behind the C-ABI -- a rank's tile share rendered by the device wavefront
driver into a tile-major buffer, gathered into the frame by yart_frame_gather over an RCCL communicator.
One GPU here, so the communicator has one rank (the exchange is the local scatter); the frame must equal
the unsharded render bit for bit, and the scatter of a several-rank layout is checked part by part."""
import numpy as np
import pytest

from yart_amd import api, scenes

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_tile_major_render_gathers_to_the_unsharded_frame():
    v, t, s = scenes.architectural(20000, seed=5)
    acc = api.Accel(v, t, s)
    acc.set_surface_data(v, t, np.zeros_like(v), s)
    W, H = 200, 120                           # not a multiple of the tile size: strips and a corner tile
    cam = api.camera_basis([-10.0, 5.0, 0.0], [1.0, -0.1, 0.05], W, H, 0.785398)
    args = (cam, W, H, 2, 4, [0.0, 9.0, 0.0], [60.0, 55.0, 50.0], [0.5, 0.5, 0.5])
    wf = api.Wavefront(acc, 1 << 14)
    whole, _, _ = wf.render_diffuse(*args, seed=3)
    comm = api.Comm(api.Comm.unique_id(), 0, 1, device=0)
    pix = torch.from_numpy(api.tiles_rank_pixels(W, H, 0, 1).astype(np.int32)).cuda()
    local, _, _ = wf.render_diffuse(*args, seed=3, pixels=pix)
    frame = comm.gather_frame(local, W, H)
    torch.cuda.synchronize()
    assert torch.equal(frame.view(-1, 3).view(torch.int32), whole.view(torch.int32))
    assert float(frame.max()) > 0
    # the parts of a 3-rank layout, rendered one after the other and scattered by the same tables
    # (what rank 0 does with the received parts)
    frame3 = torch.zeros_like(frame)
    for rank in range(3):
        pix = api.tiles_rank_pixels(W, H, rank, 3)
        part, _, _ = wf.render_diffuse(*args, seed=3, pixels=torch.from_numpy(pix.astype(np.int32)).cuda())
        frame3.view(-1, 3)[torch.from_numpy(pix.astype(np.int64)).cuda()] = part
    assert torch.equal(frame3.view(torch.int32), frame.view(torch.int32))
    # the counters / checksum exchange: over one rank the sum is the value; two gathers in a row on different streams
    # and frame shapes (the tables and the staging block are replaced: the second call waits for the first)
    vals = torch.tensor([7, 2**40 + 5, 2**63 - 1], dtype=torch.int64, device="cuda")
    assert comm.allreduce_u64(vals.clone()).tolist() == vals.tolist()
    other = torch.cuda.Stream()
    pix2 = torch.from_numpy(api.tiles_rank_pixels(W, H, 0, 1, tile=16).astype(np.int32)).cuda()
    local2, _, _ = wf.render_diffuse(*args, seed=3, pixels=pix2)
    torch.cuda.synchronize()
    f1 = comm.gather_frame(local, W, H)
    f2 = comm.gather_frame(local2, W, H, tile=16, stream=other)
    torch.cuda.synchronize()
    assert torch.equal(f1.view(torch.int32), frame.view(torch.int32)) and torch.equal(f2.view(torch.int32), frame.view(torch.int32))
    comm.close()
    wf.close()
    acc.close()
