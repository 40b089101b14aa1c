"""ctypes binding of libuob_rt.so (include/uob_rt.h) for tests and bench.py.

The library is the product; this module is plumbing.  There is no CPU fallback: if the shared library
cannot be loaded, or no HIP device is present when a context is created, the call raises.
"""
import ctypes as C
import os

import numpy as np

# Two HIP runtimes can end up in one process: torch ships its own libamdhip64, libuob_rt.so links the system's.  Loaded in
# the order torch -> libuob_rt.so both work; the other order ends in "No HIP GPUs are available" from torch.  So: if torch
# is installed, it is imported here, before lib() can load the library (torch is plumbing for tests and bench.py; the
# library itself needs none of it).
try:
    import torch as _torch  # noqa: F401
except ImportError:          # a host without torch: nothing to order
    _torch = None

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UOB_RT_LIB", os.path.join(_HERE, "libuob_rt.so"))   # override: kernel experiments only

EXPORTS = (
    "rt_abi_version", "rt_last_error", "rt_config_default", "rt_config_owned_rows", "rt_init", "rt_render",
    "rt_render_device", "rt_count_work", "rt_count_executed", "rt_last_kernel_ms", "rt_destroy", "rt_scene_cornell_box",
    "rt_scene_load_obj", "rt_scene_load_obj_ex", "rt_triangle_compute_normal", "rt_scene_pack", "rt_rotation_matrix",
    "rt_selftest_rcp", "rt_selftest_normalize", "rt_debug_trace_rays", "rt_debug_block_costs", "rt_debug_world_masks", "rt_debug_wave_timeline", "rt_register_output", "rt_unregister_output",
    "rt_debug_band_copy_plan",
)

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libuob_rt: error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libuob_rt.so (built by __graft_entry__.build() / make -C uob_raytracer_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        fp, vp = C.POINTER(C.c_float), C.c_void_p
        L.rt_last_error.restype = C.c_char_p
        L.rt_config_default.argtypes = [C.POINTER(abi.RtConfig)]
        L.rt_config_default.restype = None
        L.rt_config_owned_rows.argtypes = [C.POINTER(abi.RtConfig)]
        L.rt_init.argtypes = [C.POINTER(abi.RtConfig), fp, fp, fp, C.c_int32, C.POINTER(vp)]
        L.rt_render.argtypes = [vp, fp, fp, fp, C.c_float, C.POINTER(C.c_uint32), fp]
        L.rt_render_device.argtypes = [vp, fp, fp, fp, C.c_float, vp, vp, vp]
        L.rt_count_work.argtypes = [vp, fp, fp, fp, C.c_float, C.POINTER(abi.RtWork)]
        L.rt_count_executed.argtypes = [vp, fp, fp, fp, C.c_float, C.POINTER(C.c_uint64)]
        L.rt_last_kernel_ms.argtypes = [vp, fp]
        L.rt_destroy.argtypes = [vp]
        L.rt_destroy.restype = None
        L.rt_scene_cornell_box.argtypes = [C.POINTER(abi.RtTriangle), C.c_int32]
        L.rt_scene_load_obj.argtypes = [C.c_char_p, C.POINTER(abi.RtTriangle), C.c_int32]
        L.rt_scene_load_obj_ex.argtypes = [C.c_char_p, fp, C.c_float, fp, C.POINTER(abi.RtTriangle), C.c_int32]
        L.rt_debug_trace_rays.argtypes = [vp, C.c_int32, fp, fp, C.c_int64, C.POINTER(C.c_int32), fp]
        L.rt_debug_block_costs.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int32]
        L.rt_debug_world_masks.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.rt_debug_wave_timeline.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.rt_debug_band_copy_plan.argtypes = [C.c_int32] * 9 + [C.POINTER(abi.RtBandCopy), C.c_int32]
        L.rt_register_output.argtypes = [vp, vp, C.c_size_t]
        L.rt_unregister_output.argtypes = [vp]
        L.rt_triangle_compute_normal.argtypes = [C.POINTER(abi.RtTriangle)]
        L.rt_triangle_compute_normal.restype = None
        L.rt_scene_pack.argtypes = [C.POINTER(abi.RtTriangle), C.c_int32, fp, fp, fp]
        L.rt_scene_pack.restype = None
        L.rt_rotation_matrix.argtypes = [C.c_float, C.c_float, fp]
        L.rt_rotation_matrix.restype = None
        if L.rt_abi_version() != abi.RT_ABI_VERSION:
            raise ImportError("libuob_rt.so ABI %d != binding %d" % (L.rt_abi_version(), abi.RT_ABI_VERSION))
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))
    return rc


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def band_copy_plan(num_devices, k, device_band_rows, width, height, elem_bytes, dev_to_dev, peer_ok, same_device):
    """The copies that deliver device k's bands of a multi-device context (rt_debug_band_copy_plan): list of dicts."""
    n = _check(lib().rt_debug_band_copy_plan(num_devices, k, device_band_rows, width, height, elem_bytes, int(dev_to_dev),
                                             int(peer_ok), int(same_device), None, 0))
    buf = (abi.RtBandCopy * max(n, 1))()
    _check(lib().rt_debug_band_copy_plan(num_devices, k, device_band_rows, width, height, elem_bytes, int(dev_to_dev),
                                         int(peer_ok), int(same_device), buf, n))
    return [{f: int(getattr(buf[i], f)) for f, _ in abi.RtBandCopy._fields_ if f != "reserved"} for i in range(n)]


def selftest_rcp():
    """All-2^32-patterns check of the exact reciprocal (include/uob_rt.h rt_selftest_rcp)."""
    out = (C.c_uint64 * 64)()
    _check(lib().rt_selftest_rcp(out))
    v = list(out)
    return {"safe_mismatch_1step": v[0], "safe_mismatch_2step": v[1], "edge_mismatch_1step": v[2],
            "edge_mismatch_2step": v[3], "examples": [int(b) & 0xFFFFFFFF for b in v[8:8 + min(v[4], 56)]]}


def selftest_normalize(b_stride=64):
    """normalize()'s square root over every FP32 pattern and its quotients over every significand of a and every
    b_stride-th significand of b (include/uob_rt.h rt_selftest_normalize)."""
    out = (C.c_uint64 * 8)()
    _check(lib().rt_selftest_normalize(out, C.c_uint32(b_stride)))
    v = list(out)
    return {"sqrt_mismatches": v[0], "div_mismatches": v[1], "div_pairs": v[2], "sqrt_example": v[3], "div_example": v[4]}


def default_config():
    cfg = abi.RtConfig()
    lib().rt_config_default(C.byref(cfg))
    return cfg


def rotation_matrix(yaw, pitch):
    rot = np.zeros(12, np.float32)
    lib().rt_rotation_matrix(C.c_float(yaw), C.c_float(pitch), _fp(rot))
    return rot


class Scene:
    """Triangle list in the reference's AoS format (TestModelH.h:11-38): array [n,5,4] = v0,v1,v2,normal,color."""

    def __init__(self, aos):
        self.aos = np.ascontiguousarray(aos, np.float32).reshape(-1, 5, 4)

    def __len__(self):
        return self.aos.shape[0]

    @classmethod
    def cornell_box(cls):
        """LoadTestModel (TestModelH.h:44)."""
        buf = (abi.RtTriangle * 64)()
        n = _check(lib().rt_scene_cornell_box(buf, 64))
        return cls(np.frombuffer(buf, np.float32, n * 20).copy())

    @classmethod
    def load_obj(cls, path, color=None, scale=1.5, translate=None):
        """load_obj (Loader.cpp:11); colour / scale / translation default to the reference's constants (:20,:42,:48)."""
        col = None if color is None else _fp(np.ascontiguousarray(color, np.float32))
        mv = None if translate is None else _fp(np.ascontiguousarray(translate, np.float32))
        n = _check(lib().rt_scene_load_obj_ex(os.fsencode(path), col, C.c_float(scale), mv, None, 0))
        buf = (abi.RtTriangle * max(n, 1))()
        _check(lib().rt_scene_load_obj_ex(os.fsencode(path), col, C.c_float(scale), mv, buf, n))
        return cls(np.frombuffer(buf, np.float32, n * 20).copy())

    def __add__(self, other):
        """triangles.insert(end, ...) as at skeleton.cpp:103."""
        return Scene(np.concatenate([self.aos, other.aos], axis=0))

    def with_color(self, indices, rgba):
        aos = self.aos.copy()
        aos[list(indices), 4, :] = np.asarray(rgba, np.float32)
        return Scene(aos)

    def packed(self):
        """The three float4 arrays uploaded at skeleton.cpp:474-496."""
        n = len(self)
        v = np.zeros((3 * max(n, 1), 4), np.float32)
        nr = np.zeros((max(n, 1), 4), np.float32)
        c = np.zeros((max(n, 1), 4), np.float32)
        tris = self.aos.ctypes.data_as(C.POINTER(abi.RtTriangle))
        lib().rt_scene_pack(tris, n, _fp(v), _fp(nr), _fp(c))
        return v[:3 * n], nr[:n], c[:n]


class RayTracer:
    """One rt_ctx: the scene uploaded once (opencl_initialise), frames rendered on demand (offload_rendering)."""

    def __init__(self, cfg, scene):
        self.cfg = cfg
        self.scene = scene
        v, nr, c = scene.packed()
        self._keep = (v, nr, c)
        h = C.c_void_p()
        _check(lib().rt_init(C.byref(cfg), _fp(v), _fp(nr), _fp(c), len(scene), C.byref(h)))
        self._h = h
        self.width = cfg.width
        self.n_triangles = len(scene)
        self.rows = lib().rt_config_owned_rows(C.byref(cfg))

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _args(rot, cam, light):
        return (np.ascontiguousarray(rot, np.float32), np.ascontiguousarray(cam, np.float32)[:3].copy(),
                np.ascontiguousarray(light, np.float32)[:3].copy())

    def render(self, rot, cam, light, focal, want_rgb=False, out=None):
        """Blocking render + readback.  Returns ARGB [rows,W] (and the float4 tap [rows,W,4]).
        `out`: optional C-contiguous uint32 [rows,W] array to receive the frame (screen->buffer)."""
        rot, cam, light = self._args(rot, cam, light)
        if out is not None and (out.dtype != np.uint32 or out.shape != (self.rows, self.width) or not out.flags.c_contiguous):
            raise ValueError("out must be a C-contiguous uint32 array of shape (%d, %d)" % (self.rows, self.width))
        argb = out if out is not None else np.zeros((self.rows, self.width), np.uint32)
        rgb = np.zeros((self.rows, self.width, 4), np.float32) if want_rgb else None
        _check(lib().rt_render(self._h, _fp(rot), _fp(cam), _fp(light), C.c_float(focal),
                               argb.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(rgb) if want_rgb else None))
        return (argb, rgb) if want_rgb else argb

    def register_output(self, arr):
        """Pin and map a host array (the caller's framebuffer): later render(out=arr) calls have the kernel write the pixels
        straight into it over PCIe (rt_register_output).  Keep `arr` alive until unregister_output() / close()."""
        _check(lib().rt_register_output(self._h, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)))
        self._registered = arr

    def unregister_output(self):
        _check(lib().rt_unregister_output(self._h))
        self._registered = None

    def render_device(self, rot, cam, light, focal, d_argb_ptr, d_rgb_ptr=None, stream=None):
        """Enqueue a frame into caller-owned device memory (raw pointers, e.g. torch .data_ptr())."""
        rot, cam, light = self._args(rot, cam, light)
        _check(lib().rt_render_device(self._h, _fp(rot), _fp(cam), _fp(light), C.c_float(focal),
                                      C.c_void_p(d_argb_ptr), C.c_void_p(d_rgb_ptr or 0), C.c_void_p(stream or 0)))

    def count_work(self, rot, cam, light, focal):
        rot, cam, light = self._args(rot, cam, light)
        w = abi.RtWork()
        _check(lib().rt_count_work(self._h, _fp(rot), _fp(cam), _fp(light), C.c_float(focal), C.byref(w)))
        return w.as_dict()

    def count_executed(self, rot, cam, light, focal):
        """Executed work of the wave kernel (include/uob_rt.h rt_count_executed)."""
        rot, cam, light = self._args(rot, cam, light)
        out = (C.c_uint64 * 8)()
        _check(lib().rt_count_executed(self._h, _fp(rot), _fp(cam), _fp(light), C.c_float(focal), out))
        if self.n_triangles > 64:      # tiled mesh kernel
            keys = ("primary_tile_visits", "primary_bound_survivors", "shadow_tile_visits", "level1_survivors",
                    "level3_pair_calls", "level3_stage1_iterations", "longest_block_ticks", "wave_task_rounds")
        else:
            keys = ("surface_points", "stage1_wave_iterations", "stage2_wave_iterations", "sphere_wave_evaluations",
                    "culled_pairs", "tasks_resolved_whole", "sampled_points_fully_lit", "sampled_points_fully_blocked")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def trace_in_shadow(self, rays, radius_sq):
        """Device in_shadow (kernels.cl:243) on caller rays [k,6] = start, direction -> uint8 [k]."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        r2 = np.ascontiguousarray(radius_sq, np.float32)
        out = np.zeros(rays.shape[0], np.int32)
        _check(lib().rt_debug_trace_rays(self._h, abi.RT_TRACE_IN_SHADOW, _fp(rays), _fp(r2), rays.shape[0],
                                         out.ctypes.data_as(C.POINTER(C.c_int32)), None))
        return out.astype(np.uint8)

    def trace_closest_hit(self, rays):
        """Device single_ray_intersections (kernels.cl:168) on caller rays -> (tri [k], out [k,10])."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        tri = np.zeros(rays.shape[0], np.int32)
        out = np.zeros((rays.shape[0], 10), np.float32)
        _check(lib().rt_debug_trace_rays(self._h, abi.RT_TRACE_CLOSEST_HIT, _fp(rays), None, rays.shape[0],
                                         tri.ctypes.data_as(C.POINTER(C.c_int32)), _fp(out)))
        return tri, out

    def wave_timeline(self):
        """Wave kernel, context created with UOB_RT_TIMELINE=1: start / end statistics of the last frame's persistent waves
        in 100 MHz ticks (rt_debug_wave_timeline)."""
        out = (C.c_uint64 * 8)()
        _check(lib().rt_debug_wave_timeline(self._h, out))
        n = max(int(out[0]), 1)
        first, last = int(out[1]), int(out[2])
        return {"waves": int(out[0]), "span_us": (last - first) / 100.0, "mean_start_us": (int(out[3]) / n - first) / 100.0,
                "mean_idle_tail_us": (last - int(out[4]) / n) / 100.0, "jobs": int(out[5]), "max_jobs_per_wave": int(out[6]), "listed_jobs": int(out[7])}

    def world_masks(self):
        """Mesh kernel: the last frame's shadow-ray tile masks, uint64 [G, G, G, words] indexed [z, y, x] (rt_debug_world_masks)."""
        g, w = C.c_int32(), C.c_int32()
        n = _check(lib().rt_debug_world_masks(self._h, None, 0, C.byref(g), C.byref(w)))
        out = np.zeros(n, np.uint64)
        _check(lib().rt_debug_world_masks(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(g), C.byref(w)))
        return out.reshape(g.value, g.value, g.value, w.value)

    def block_costs(self):
        """Mesh kernel: s_memtime ticks of every 16x16-pixel block of the last frame, [rows/16, W/16] (rt_debug_block_costs)."""
        n = _check(lib().rt_debug_block_costs(self._h, None, 0))
        out = np.zeros(n, np.uint32)
        _check(lib().rt_debug_block_costs(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out.reshape((self.rows + 15) // 16, (self.width + 15) // 16)

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().rt_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)
