"""Throughput of ONE rank's bands of an N-rank job when consecutive frames come from M contexts on M streams (the tail of
frame k overlaps the start of frame k+1) against one context on one stream — the render alone, no gather.
usage: band_pipeline.py [N ...]      env: BANDS (band rows, default 16), RANK (default 0), CONTEXTS (default "1 2")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uob_raytracer_amd import abi, runtime as rt
rot, cam, light = rt.rotation_matrix(0, 0), [0, 0, -3.2], [0, -0.5, -0.7]
br, rank = int(os.environ.get("BANDS", "16")), int(os.environ.get("RANK", "0"))
for bc in [int(v) for v in (sys.argv[1:] or ["8"])]:
    cfg = abi.make_config(width=4096, height=4096, aa_x=4, aa_y=2, shadow_samples=64, band_rows=br if bc > 1 else 0, band_index=rank % bc, band_count=bc)
    for M in [int(v) for v in os.environ.get("CONTEXTS", "1 2").split()]:
        trs = [rt.RayTracer(cfg, rt.Scene.cornell_box()) for _ in range(M)]
        bufs = [torch.empty((trs[0].rows, 4096), dtype=torch.int32, device="cuda") for _ in range(M)]
        streams = [torch.cuda.Stream() for _ in range(M)]
        def frame(k):
            i = k % M
            trs[i].render_device(rot, cam, light, 17600.0, bufs[i].data_ptr(), None, streams[i].cuda_stream)
        for k in range(60):
            frame(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 300
        for k in range(K):
            frame(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        print("ranks %d rows %d  %d context(s): %.4f ms/frame  -> x%d = %.3f ms per whole frame" % (bc, trs[0].rows, M, dt, bc, dt * bc), flush=True)
        for t in trs:
            t.close()
