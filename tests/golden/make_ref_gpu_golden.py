#!/usr/bin/env python3
"""Generates tests/golden/ref_gfx950.npz / .json: frames rendered by THE REFERENCE ITSELF on a real OpenCL device.

Runs on the GPU box (`gpurun -- python tests/golden/make_ref_gpu_golden.py`, output lands in gpurun_out/ and is copied
to tests/golden/ by hand; `--mesh`: the box + OBJ-mesh frames, ref_gfx950_mesh.npz / .json).  What renders the frames: /root/reference/Source/kernels.cl compiled where it lies for gfx950
against AMD's own OpenCL builtin library with the reference's own build options (oracle/build_ref.py, code objects
under oracle/_ref/), loaded through the OpenCL runtime and launched as skeleton.cpp launches it (oracle/ref_cl_host.c).
No builtin, header or library is replaced by anything of ours.  The scene is the committed output of the reference's
own LoadTestModel (tests/golden/scene_cornell_aos.npy).  What is stored is data: ARGB frames (the kernel's only
output) and the run's metadata — never reference source.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyref as R, ref_gpu  # noqa: E402

POSES = [  # (yaw, pitch, cam, light) — the poses of make_golden.py
    (0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]),
    (0.3, -0.2, [0.2, 0.1, -2.9], [-0.3, -0.5, -0.7]),
]
# code object (oracle/build_ref.py GPU_VARIANTS) -> (rt_config keywords expressing the same constants, scene, poses)
FRAMES = {
    "default":    (dict(width=1024, height=1024), "box", (0, 1)),           # the reference exactly as shipped
    "default256": (dict(width=256, height=256), "box", (0, 1)),
    "cfg1":       (dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=()), "box", (0, 1)),
    "cfg2":       (dict(width=1024, height=1024, shadow_samples=16, spheres=()), "box", (0,)),
    "cfg3":       (dict(width=1920, height=1080, max_bounces=5), "mirrorwall", (0, 1)),
    "s64_512":    (dict(width=512, height=512, shadow_samples=64), "box", (0,)),
    "aa3_256":    (dict(width=256, height=256, aa_x=3, aa_y=3), "box", (0,)),
}
# the headline's sample count at the headline's size (2x2 AA: the reference cannot express 4x2): crops only
CROPS_4096 = [(1536, 2560, 512, 256), (600, 2900, 512, 192), (2304, 1100, 384, 256)]     # (x0, y0, w, h)


# Box + OBJ mesh, the shape of the reference's main() (skeleton.cpp:102-103 appends "bunny_200.obj", a file the reference does
# not hold: the meshes are uob_raytracer_amd/meshgen.py's, read by the product's Loader.cpp counterpart, which is pinned against
# the reference's own loader in tests/test_scene.py).  The reference stages the whole scene in local memory (kernels.cl:374-376,
# 80 bytes per triangle of the device's 64 KB): ~800 triangles is the most its kernel can render at all.
#   name -> (code object, rt_config keywords, (n_lon, n_lat, wound outward?) of the mesh, poses)
MESH_FRAMES = {
    "mesh224_default":    ("default", dict(width=1024, height=1024), (16, 8, 1), (0,)),      # the reference as shipped + a 224-triangle mesh
    "mesh224_default256": ("default256", dict(width=256, height=256), (16, 8, 1), (0, 1)),
    "mesh624_cfg1":       ("cfg1", dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=()), (24, 14, 1), (0, 1)),
    "mesh624_inside_out": ("cfg1", dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=()), (24, 14, 0), (0,)),
}


def mesh_scene(lon, lat, outward=1):
    """AoS [n,5,4] of the Cornell Box + the synthetic OBJ mesh, built by the product's host code (no GPU involved)."""
    import tempfile
    from uob_raytracer_amd import meshgen, runtime as rt
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.obj")
        meshgen.write_sphere_obj(path, lon, lat, outward=bool(outward))
        return (rt.Scene(np.load(os.path.join(HERE, "scene_cornell_aos.npy"))) + rt.Scene.load_obj(path)).aos


def main_mesh():
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    probe = ref_gpu.probe()
    print(json.dumps(probe), flush=True)
    if probe.get("opencl_gpu_devices", 0) < 1:
        sys.exit("no OpenCL GPU device on this machine")
    arrays, meta = {}, {"probe": probe, "poses": POSES, "frames": {}}
    for name, (variant, kw, (lon, lat, outw), poses) in MESH_FRAMES.items():
        aos = mesh_scene(lon, lat, outw)
        v, n, c = R.pack_scene(aos)
        meta["frames"][name] = {"variant": variant, "config": kw, "mesh": [lon, lat, outw], "triangles": int(aos.shape[0]),
                                "scene_fnv": "%016x" % R.fnv1a64_words(np.ascontiguousarray(aos).view(np.uint32).ravel()),
                                "poses": list(poses), "runs": {}}
        for pi in poses:
            yaw, pitch, cam, light = POSES[pi]
            argb, info = ref_gpu.run(variant, kw["width"], kw["height"], v, n, c, R.rot_matrix(yaw, pitch), cam, light,
                                     focal_for(kw), reps=2)
            arrays["%s_p%d" % (name, pi)] = argb.reshape(kw["height"], kw["width"])
            info["fnv_words"] = "%016x" % R.fnv1a64_words(argb)
            meta["frames"][name]["runs"]["p%d" % pi] = info
            print(name, pi, json.dumps(info), flush=True)
    np.savez_compressed(os.path.join(out_dir, "ref_gfx950_mesh.npz"), **arrays)
    with open(os.path.join(out_dir, "ref_gfx950_mesh.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("written to", out_dir)


def focal_for(kw):
    return 1100.0 * min(kw["width"], kw["height"]) / 1024.0 * kw.get("aa_x", 2)


def main():
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    probe = ref_gpu.probe()
    print(json.dumps(probe), flush=True)
    if probe.get("opencl_gpu_devices", 0) < 1:
        sys.exit("no OpenCL GPU device on this machine")
    aos = np.load(os.path.join(HERE, "scene_cornell_aos.npy"))
    scenes = {"box": R.pack_scene(aos)}
    aos_m = aos.copy()
    aos_m[[8, 9], 4, :] = (1.0, 1.0, 1.0, 0.0)          # back wall -> mirror (TestModelH.h:58)
    scenes["mirrorwall"] = R.pack_scene(aos_m)

    arrays, meta = {}, {"probe": probe, "poses": POSES, "frames": {}, "crops_4096": {}}
    for name, (kw, sc, poses) in FRAMES.items():
        v, n, c = scenes[sc]
        meta["frames"][name] = {"config": kw, "scene": sc, "poses": list(poses), "runs": {}}
        for pi in poses:
            yaw, pitch, cam, light = POSES[pi]
            argb, info = ref_gpu.run(name, kw["width"], kw["height"], v, n, c, R.rot_matrix(yaw, pitch), cam, light,
                                     focal_for(kw), reps=5)
            arrays["%s_p%d" % (name, pi)] = argb.reshape(kw["height"], kw["width"])
            info["fnv_words"] = "%016x" % R.fnv1a64_words(argb)
            info["black_pixels"] = int((argb == 0xFF000000).sum())
            meta["frames"][name]["runs"]["p%d" % pi] = info
            print(name, pi, json.dumps(info), flush=True)
    kw = dict(width=4096, height=4096, shadow_samples=64)
    v, n, c = scenes["box"]
    yaw, pitch, cam, light = POSES[0]
    argb, info = ref_gpu.run("s64_4096", 4096, 4096, v, n, c, R.rot_matrix(yaw, pitch), cam, light, focal_for(kw), reps=2)
    frame = argb.reshape(4096, 4096)
    for i, (x0, y0, w, h) in enumerate(CROPS_4096):
        arrays["s64_4096_crop%d" % i] = frame[y0:y0 + h, x0:x0 + w].copy()
    info["black_pixels"] = int((argb == 0xFF000000).sum())
    info["checksum_u64"] = int(argb.astype(np.uint64).sum())
    meta["crops_4096"] = {"config": kw, "crops": CROPS_4096, "run": info}
    print("s64_4096", json.dumps(info), flush=True)
    np.savez_compressed(os.path.join(out_dir, "ref_gfx950.npz"), **arrays)
    with open(os.path.join(out_dir, "ref_gfx950.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("written to", out_dir)


if __name__ == "__main__":
    main_mesh() if "--mesh" in sys.argv[1:] else main()
