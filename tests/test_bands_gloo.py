"""CPU, world_size 2, 3 and 4 over gloo (incl. heights that are no multiple of band_rows * world): the N>1 path of bench.py — band partition (rt_config.band_*), one
gather to rank 0, de-interleave (uob_raytracer_amd/bands.py) — with the CPU oracle standing in for the
HIP kernel as the per-rank renderer.  The rebuilt frame must equal the whole-frame render bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, band_rows, height, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyref
    from uob_raytracer_amd import abi, bands, runtime as rt
    kw = dict(width=64, height=height, shadow_samples=4)
    scene = rt.Scene.cornell_box()           # host code only
    v, n, c = scene.packed()
    cfg = abi.make_config(band_rows=band_rows, band_index=rank, band_count=world, **kw)
    rot = rt.rotation_matrix(0.2, -0.1)
    cam, light = [0.0, 0.1, -3.0], [0.1, -0.5, -0.6]
    focal = 1100.0 * 64 / 1024.0 * 2
    argb, _ = pyref.Oracle().render(cfg, v, n, c, rot, cam, light, focal, nthreads=2)
    rows = bands.band_rows_of(rank, world, kw["height"], band_rows)
    assert len(rows) == rt.lib().rt_config_owned_rows(cfg)
    stripe = torch.full((bands.padded_rows(height, world, band_rows), kw["width"]), -1, dtype=torch.int32)
    stripe[:len(rows)] = torch.from_numpy(argb.view(np.int32).reshape(len(rows), kw["width"]).copy())
    frame = bands.gather_frame(stripe, world, rank, band_rows, height=height)
    if rank == 0:
        whole, _ = pyref.Oracle().render(abi.make_config(**kw), v, n, c, rot, cam, light, focal, nthreads=2)
        ok = np.array_equal(frame.numpy().view(np.uint32).ravel(), whole)
        open(out_path, "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


# (4, 8, 70): 9 bands, the last one of 6 rows: ranks own 22 / 16 / 16 / 16 rows; (3, 4, 10): rank 2 owns a 2-row band
@pytest.mark.parametrize("world,band_rows,height", [(2, 8, 32), (3, 4, 24), (4, 8, 70), (3, 4, 10), (4, 16, 20)])
def test_band_gather_rebuilds_the_frame(world, band_rows, height, tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), band_rows, height, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_partition_helpers():
    from uob_raytracer_amd import bands
    assert bands.band_rows_of(1, 2, 16, 4) == [4, 5, 6, 7, 12, 13, 14, 15]
    assert bands.padded_rows(4096, 8, 32) == 512 and bands.padded_rows(100, 8, 32) == 32
    assert bands.padded_rows(70, 4, 8) == 24 and bands.padded_rows(1080, 8, 32) == 160
