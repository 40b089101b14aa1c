"""Throughput of ONE rank's bands of an N-rank job when consecutive frames come from two contexts on two streams (the
tail of frame k overlaps the start of frame k+1) against one context on one stream.  usage: band_pipeline.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uob_raytracer_amd import abi, runtime as rt
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
for bc in [int(v) for v in (sys.argv[1:] or ["8"])]:
    cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=32 if bc > 1 else 0, band_index=0, band_count=bc)
    trs = [rt.RayTracer(cfg, rt.Scene.cornell_box()) for _ in range(2)]
    bufs = [torch.empty((trs[0].rows, 4096), dtype=torch.int32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for mode in ("one context, one stream", "two contexts, two streams"):
        def frame(k):
            i = k % 2 if mode.startswith("two") else 0
            trs[i].render_device(rot, cam, light, 17600.0, bufs[i].data_ptr(), None, streams[i].cuda_stream)
        for k in range(6):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for k in range(K):
            frame(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        print("band_count %d rows %d  %-28s %.3f ms/frame" % (bc, trs[0].rows, mode, dt), flush=True)
