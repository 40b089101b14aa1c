#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — builds the *real* reference kernel for x86-64 into oracle/_ref/.

Recipe: compile /root/reference/Source/kernels.cl where it lies with
ROCm clang `-x cl -target x86_64-unknown-linux-gnu`, link it with oracle/ref_shim.cpp (OpenCL builtins
+ per-pixel driver).  Every knob of the reference kernel is an unconditional #define/const in the file
(kernels.cl:7-17, :316-317, :343), so variant configurations need a textual patch: the patched text is
piped to the compiler on stdin — it never exists as a file in this repository or under oracle/_ref/,
so no reference source is committed or travels to the GPU box; only the built .so files do.

Usage: python oracle/build_ref.py [variant ...]     (default: all variants)
Needs /root/reference (container only).  The product never loads anything built here.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("UOB_REFERENCE", "/root/reference")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"

STRICT = ["-O2", "-ffp-contract=off"]
FAST = ["-O2", "-cl-fast-relaxed-math", "-cl-mad-enable"]   # the reference's own options, skeleton.cpp:407

# name -> dict(W,H,aa (square grid side),S,spread,bounces,spheres(bool),flags)
VARIANTS = {
    # the reference exactly as shipped
    "default":      dict(W=1024, H=1024, aa=2, S=10, spread=0.05, bounces=10, spheres=True, flags=STRICT),
    "default_fast": dict(W=1024, H=1024, aa=2, S=10, spread=0.05, bounces=10, spheres=True, flags=FAST),
    # shipped constants on a small frame (committed full-frame fixture)
    "default256":   dict(W=256, H=256, aa=2, S=10, spread=0.05, bounces=10, spheres=True, flags=STRICT),
    # BASELINE.json configs[0]: 256x256, 1 spp, hard shadow, no spheres
    "cfg1":         dict(W=256, H=256, aa=1, S=1, spread=0.0, bounces=10, spheres=False, flags=STRICT),
    # configs[1]: 1024x1024, 4xAA, 16 shadow rays, diffuse only
    "cfg2":         dict(W=1024, H=1024, aa=2, S=16, spread=0.05, bounces=10, spheres=False, flags=STRICT),
    "cfg2_256":     dict(W=256, H=256, aa=2, S=16, spread=0.05, bounces=10, spheres=False, flags=STRICT),
    # configs[2]: 1920x1080, 4xAA, recursion depth 5, spheres on (mirror wall comes from the scene colours)
    "cfg3":         dict(W=1920, H=1080, aa=2, S=10, spread=0.05, bounces=5, spheres=True, flags=STRICT),
    "cfg3_480":     dict(W=480, H=270, aa=2, S=10, spread=0.05, bounces=5, spheres=True, flags=STRICT),
    # 64 shadow samples (headline sample count) on grids the reference can express
    "s64_512":      dict(W=512, H=512, aa=2, S=64, spread=0.05, bounces=10, spheres=True, flags=STRICT),
    "s64_4096":     dict(W=4096, H=4096, aa=2, S=64, spread=0.05, bounces=10, spheres=True, flags=STRICT),
    # 3x3 supersampling
    "aa3_256":      dict(W=256, H=256, aa=3, S=10, spread=0.05, bounces=10, spheres=True, flags=STRICT),
}


def _sub(text, pattern, repl):
    new, k = re.subn(pattern, repl, text, count=1)
    if k != 1:
        raise RuntimeError("patch point not found: %s" % pattern)
    return new


def patched_source(v):
    with open(os.path.join(REF, "Source", "kernels.cl")) as f:
        t = f.read()
    t = _sub(t, r"#define SCREEN_WIDTH [0-9.]+f", "#define SCREEN_WIDTH %d.0f" % v["W"])
    t = _sub(t, r"#define SCREEN_HEIGHT [0-9.]+f", "#define SCREEN_HEIGHT %d.0f" % v["H"])
    t = _sub(t, r"constant char rays_x = \d+;", "constant char rays_x = %d;" % v["aa"])
    t = _sub(t, r"constant char rays_y = \d+;", "constant char rays_y = %d;" % v["aa"])
    t = _sub(t, r"#define aa_rays \d+", "#define aa_rays %d" % (v["aa"] * v["aa"]))
    t = _sub(t, r"const short light_sources = \d+;", "const short light_sources = %d;" % v["S"])
    t = _sub(t, r"const float light_spread = [0-9.]+f;", "const float light_spread = %sf;" % repr(float(v["spread"])))
    t = _sub(t, r"const int bounces = \d+;", "const int bounces = %d;" % v["bounces"])
    if not v["spheres"]:   # radius^2 = -1 disables a sphere with no other edit (discriminant < 0 always)
        t = _sub(t, r"sphere_radius_sqs\[SPHERES\] = \{[^}]*\};", "sphere_radius_sqs[SPHERES] = {-1.0f, -1.0f, 0.1f};")
    return t


def build_variant(name):
    v = VARIANTS[name]
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "k_%s.o" % name)
    so = os.path.join(OUT, "libref_%s.so" % name)
    src = patched_source(v)
    subprocess.run([CLANG, "-x", "cl", "-cl-std=CL1.2", "-target", "x86_64-unknown-linux-gnu", *v["flags"],
                    "-Wno-incompatible-pointer-types", "-Wno-excess-initializers", "-fPIC",
                    "-c", "-", "-o", obj], input=src.encode(), check=True)
    subprocess.run([CLANGXX, "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                    "-DREF_W=%d" % v["W"], "-DREF_H=%d" % v["H"],
                    os.path.join(HERE, "ref_shim.cpp"), obj, "-o", so], check=True)
    os.remove(obj)
    return so


# ---- the same kernel text for the GPU itself ----------------------------------------------------------------------
# clang -x cl for amdgcn-amd-amdhsa links AMD's OpenCL builtin library (opencl.bc / ocml.bc / ockl.bc from the ROCm
# installation) by itself: every builtin the kernel calls is the one a clBuildProgram on this GPU would use, and the
# options are the reference's own (skeleton.cpp:407).  Nothing of ours is linked in: this IS the reference built here,
# for the OpenCL device of the GPU box.  oracle/ref_cl_host.c (our stand-in for the SDL-bound host) loads the code object
# with clCreateProgramWithBinary and runs it; oracle/ref_gpu.py drives that.
GPU_ARCH = "gfx950"
GPU_FAST = ["-O3", "-cl-fast-relaxed-math", "-cl-mad-enable"]      # -O3: the OpenCL runtime's default level
GPU_PLAIN = ["-O3"]                                                 # no build options at all (OpenCL defaults)
GPU_VARIANTS = {
    # name -> (constants of VARIANTS[...], flags)
    "default":    ("default", GPU_FAST),        # the reference exactly as it builds and runs itself
    "default_plain": ("default", GPU_PLAIN),
    "default256": ("default256", GPU_FAST),
    "cfg1":       ("cfg1", GPU_FAST),
    "cfg2":       ("cfg2", GPU_FAST),
    "cfg3":       ("cfg3", GPU_FAST),           # 1920 x 1080: width is a multiple of 128, height of 4
    "s64_512":    ("s64_512", GPU_FAST),
    "s64_4096":   ("s64_4096", GPU_FAST),
    "aa3_256":    ("aa3_256", GPU_FAST),
}


def build_gpu_variant(name):
    base, flags = GPU_VARIANTS[name]
    os.makedirs(OUT, exist_ok=True)
    co = os.path.join(OUT, "ref_%s_%s.co" % (name, GPU_ARCH))
    src = patched_source(VARIANTS[base])
    subprocess.run([CLANG, "-x", "cl", "-cl-std=CL1.2", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + GPU_ARCH, *flags,
                    "-Wno-incompatible-pointer-types", "-Wno-excess-initializers", "-", "-o", co],
                   input=src.encode(), check=True)
    return co


def build_scene():
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libref_scene.so")
    subprocess.run(["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w",
                    "-I" + os.path.join(REF, "Source"), "-I" + os.path.join(REF, "glm"),
                    os.path.join(HERE, "ref_scene.cpp"), "-o", so], check=True)
    return so


def main(argv):
    if not os.path.isdir(REF):
        print("reference not present at %s: nothing to build (prebuilt oracle/_ref/ files are used)" % REF)
        return 0
    names = argv or list(VARIANTS)
    build_scene()
    for n in names:
        if n.startswith("gpu:"):
            print("built", build_gpu_variant(n[4:]))
        else:
            print("built", build_variant(n))
    if not argv:
        for n in GPU_VARIANTS:
            print("built", build_gpu_variant(n))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
