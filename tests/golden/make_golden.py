#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the REAL reference code.

Runs only in the build container (needs /root/reference and the libraries oracle/build_ref.py builds from
it): the reference kernel Source/kernels.cl, compiled for x86-64 where it lies, renders the frames; the
reference's own LoadTestModel / load_obj produce the scene vectors.  What is stored is data — inputs and
expected outputs — never reference source.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyref as R  # noqa: E402

POSES = [  # (yaw, pitch, cam, light)
    (0.0, 0.0, [0.0, 0.0, -3.2], [0.0, -0.5, -0.7]),
    (0.3, -0.2, [0.2, 0.1, -2.9], [-0.3, -0.5, -0.7]),
]

# variant (oracle/build_ref.py) -> rt_config keyword arguments that express the same constants
FRAMES = {
    "default256": dict(width=256, height=256),
    "cfg1": dict(width=256, height=256, aa_x=1, aa_y=1, shadow_samples=1, light_spread=0.0, spheres=()),
    "cfg2_256": dict(width=256, height=256, shadow_samples=16, spheres=()),
    "cfg3_480": dict(width=480, height=270, max_bounces=5),
    "aa3_256": dict(width=256, height=256, aa_x=3, aa_y=3),
}
# large frames: FNV hash of the whole frame + a seeded pixel subset with values
BIG = {
    "default": dict(width=1024, height=1024),
    "default_fast": dict(width=1024, height=1024),
    "cfg2": dict(width=1024, height=1024, shadow_samples=16, spheres=()),
    "cfg3": dict(width=1920, height=1080, max_bounces=5),
    "s64_512": dict(width=512, height=512, shadow_samples=64),
}
SUBSET_ONLY = {  # too slow to render whole on the CPU: seeded pixel subset only
    "s64_4096": dict(width=4096, height=4096, shadow_samples=64),
}


def focal_for(kw):
    return 1100.0 * min(kw["width"], kw["height"]) / 1024.0 * kw.get("aa_x", 2)


def main():
    aos = R.ref_load_test_model()
    v, n, c = R.pack_scene(aos)
    meta = {"poses": POSES, "frames": {}, "big": {}, "subset": {}, "scene": {}}
    np.save(os.path.join(HERE, "scene_cornell_aos.npy"), aos)
    meta["scene"] = {"n": int(aos.shape[0]), "fnv_vertices": "%016x" % R.fnv1a64_bytes(v.tobytes()),
                     "fnv_normals": "%016x" % R.fnv1a64_bytes(n.tobytes()),
                     "fnv_colors": "%016x" % R.fnv1a64_bytes(c.tobytes())}
    # the configs[2] scene: back wall -> mirror (TestModelH.h:58 `mirror`), used with cfg3
    aos_m = aos.copy()
    aos_m[[8, 9], 4, :] = (1.0, 1.0, 1.0, 0.0)
    vm, nm, cm = R.pack_scene(aos_m)

    arrays = {}
    for name, kw in FRAMES.items():
        k = R.RefKernel(name)
        for pi, (yaw, pitch, cam, light) in enumerate(POSES):
            argb, rgb = k.render(v, n, c, R.rot_matrix(yaw, pitch), cam, light, focal_for(kw))
            arrays["%s_p%d_argb" % (name, pi)] = argb.reshape(kw["height"], kw["width"])
            arrays["%s_p%d_tap" % (name, pi)] = rgb.reshape(kw["height"], kw["width"], 3)
        meta["frames"][name] = kw
    # mirror-wall scene on the cfg3 kernel
    k = R.RefKernel("cfg3_480")
    argb, rgb = k.render(vm, nm, cm, R.rot_matrix(0.3, -0.2), POSES[1][2], POSES[1][3], focal_for(FRAMES["cfg3_480"]))
    arrays["cfg3_480_mirrorwall_argb"] = argb.reshape(270, 480)
    arrays["cfg3_480_mirrorwall_tap"] = rgb.reshape(270, 480, 3)
    np.savez_compressed(os.path.join(HERE, "frames_small.npz"), **arrays)

    rng = np.random.default_rng(20261004)
    sub = {}
    for name, kw in BIG.items():
        k = R.RefKernel(name)
        yaw, pitch, cam, light = POSES[0]
        argb, rgb = k.render(v, n, c, R.rot_matrix(yaw, pitch), cam, light, focal_for(kw))
        pix = np.sort(rng.choice(kw["width"] * kw["height"], size=4096, replace=False)).astype(np.int32)
        sub[name + "_pix"] = pix
        sub[name + "_argb"] = argb[pix]
        sub[name + "_tap"] = rgb[pix]
        meta["big"][name] = dict(config=kw, fnv_words="%016x" % R.fnv1a64_words(argb),
                                 black_pixels=int((argb == 0xFF000000).sum()))
    for name, kw in SUBSET_ONLY.items():
        k = R.RefKernel(name)
        yaw, pitch, cam, light = POSES[0]
        pix = np.sort(rng.choice(kw["width"] * kw["height"], size=20000, replace=False)).astype(np.int32)
        argb, rgb = k.render(v, n, c, R.rot_matrix(yaw, pitch), cam, light, focal_for(kw), pix=pix)
        sub[name + "_pix"] = pix
        sub[name + "_argb"] = argb
        sub[name + "_tap"] = rgb
        meta["subset"][name] = kw
    np.savez_compressed(os.path.join(HERE, "frames_subsets.npz"), **sub)

    # function-level vectors: in_shadow (kernels.cl:243) and single_ray_intersections (:168), default kernel
    k = R.RefKernel("default")
    nray = 20000
    starts = rng.uniform(-1.0, 1.0, (nray, 3)).astype(np.float32)
    target = np.array([0.0, -0.5, -0.7], np.float32) + rng.uniform(-0.3, 0.3, (nray, 3)).astype(np.float32)
    dirs = (target - starts).astype(np.float32)
    rays = np.concatenate([starts, dirs], axis=1).astype(np.float32)
    r2 = (dirs * dirs).sum(axis=1).astype(np.float32)
    shadow = k.in_shadow(v, c, rays, r2)
    dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    rays_n = np.concatenate([starts, dn.astype(np.float32)], axis=1).astype(np.float32)
    tri, hit = k.closest_hit(v, n, c, rays_n)
    np.savez_compressed(os.path.join(HERE, "function_vectors.npz"), rays=rays, radius_sq=r2, in_shadow=shadow,
                        rays_unit=rays_n, hit_tri=tri, hit_out=hit)

    # OBJ loader vector (Loader.cpp:11): synthetic mesh in the only syntax the reference's parser accepts
    obj = os.path.join(HERE, "mesh_small.obj")
    vv = rng.uniform(-0.3, 0.3, (40, 3))
    ff = rng.integers(1, 41, (60, 3))
    with open(obj, "w") as f:
        f.write("# synthetic test mesh: v/f records only (Loader.cpp accepts nothing else)\n")
        for a in vv:
            f.write("v %.6f %.6f %.6f\n" % tuple(a))
        f.write("vn 0 0 1\n")     # ignored by the reference's parser
        for a in ff:
            f.write("f %d %d %d\n" % tuple(a))
    np.save(os.path.join(HERE, "mesh_small_aos.npy"), R.ref_load_obj(obj))

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
