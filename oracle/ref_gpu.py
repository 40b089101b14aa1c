"""TEST INFRASTRUCTURE ONLY — the reference's own kernel on the box's OpenCL device.

`oracle/_ref/ref_<variant>_gfx950.co` is /root/reference/Source/kernels.cl compiled where it lies by oracle/build_ref.py
for gfx950 against AMD's own OpenCL builtin library with the reference's own build options (skeleton.cpp:407) — no
stand-in for any builtin.  `oracle/ref_cl_host` (oracle/ref_cl_host.c) loads it through the OpenCL runtime and launches
`draw` as skeleton.cpp does.  It runs in a process of its own: the OpenCL runtime and the HIP runtime of the product
are never in one address space.

Only tests/, tests/golden/make_ref_gpu_golden.py and bench.py's baseline leg may import this module.
"""
import json
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
HOST = os.path.join(HERE, "ref_cl_host")
ARCH = "gfx950"


def code_object(variant):
    return os.path.join(REF_DIR, "ref_%s_%s.co" % (variant, ARCH))


def have(variant="default"):
    return os.path.exists(HOST) and os.path.exists(code_object(variant))


def probe(timeout=120):
    """Platforms / devices the OpenCL ICD loader finds on this machine (dict), or {"error": ...}."""
    if not os.path.exists(HOST):
        return {"error": "oracle/ref_cl_host not built"}
    try:
        out = subprocess.run([HOST, "probe"], capture_output=True, timeout=timeout, text=True)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001 - a probe reports, it does not fail the caller
        return {"error": "%s: %s" % (type(e).__name__, e)}


def gpu_available():
    p = probe()
    return p.get("opencl_gpu_devices", 0) > 0


def run(variant, W, H, verts4, normals4, colors4, rot12, cam3, light3, focal, reps=1, timeout=600):
    """One frame of the reference kernel `variant` (its W, H, AA, samples are compile-time constants of the code object;
    W and H here are the NDRange and must be the ones it was built for).  Returns (argb[H*W] uint32, info dict)."""
    v = np.ascontiguousarray(verts4, np.float32).reshape(-1, 4)
    nr = np.ascontiguousarray(normals4, np.float32).reshape(-1, 4)
    c = np.ascontiguousarray(colors4, np.float32).reshape(-1, 4)
    n = c.shape[0]
    assert v.shape[0] == 3 * n and nr.shape[0] == n
    rot = np.ascontiguousarray(rot12, np.float32).reshape(12)
    cam = np.zeros(4, np.float32); cam[:3] = cam3; cam[3] = 1.0          # camera_position is a vec4 (.., 1), skeleton.cpp:62
    li = np.zeros(4, np.float32); li[:3] = light3; li[3] = 1.0
    with tempfile.TemporaryDirectory() as d:
        job, out = os.path.join(d, "job.bin"), os.path.join(d, "out.bin")
        with open(job, "wb") as f:
            f.write(struct.pack("<iiiif", n, W, H, reps, float(np.float32(focal))))
            f.write(rot.tobytes()); f.write(cam.tobytes()); f.write(li.tobytes())
            f.write(v.tobytes()); f.write(nr.tobytes()); f.write(c.tobytes())
        p = subprocess.run([HOST, "run", code_object(variant), job, out], capture_output=True, timeout=timeout, text=True)
        if p.returncode != 0:
            raise RuntimeError("ref_cl_host failed (%d): %s" % (p.returncode, p.stderr.strip()))
        info = json.loads(p.stdout.strip().splitlines()[-1])
        argb = np.fromfile(out, np.uint32)
    assert argb.size == W * H
    return argb, info
