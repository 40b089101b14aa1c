"""ctypes mirror of include/uob_rt.h (struct layouts and constants only; loads nothing)."""
import ctypes as C

RT_ABI_VERSION = 2
RT_MAX_SPHERES = 4
RT_MAX_DEVICES = 8
RT_OK, RT_E_INVALID, RT_E_DEVICE, RT_E_NOMEM, RT_E_IO, RT_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
RT_FLAG_GENERIC_KERNEL = 2
RT_FLAG_NO_CULL = 4
RT_FLAG_NO_TILE_BINS = 8
RT_FLAG_PLAIN_ORDER = 16
RT_FLAG_STAGED_GATHER = 32
RT_TRACE_IN_SHADOW, RT_TRACE_CLOSEST_HIT = 0, 1


class RtSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius_sq", C.c_float), ("color", C.c_float * 4)]


class RtConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("aa_x", C.c_int32), ("aa_y", C.c_int32),
        ("shadow_samples", C.c_int32), ("light_spread", C.c_float),
        ("max_bounces", C.c_int32), ("num_spheres", C.c_int32),
        ("spheres", RtSphere * RT_MAX_SPHERES),
        ("band_rows", C.c_int32), ("band_index", C.c_int32), ("band_count", C.c_int32),
        ("device", C.c_int32), ("flags", C.c_int32),
        ("num_devices", C.c_int32), ("devices", C.c_int32 * RT_MAX_DEVICES), ("device_band_rows", C.c_int32),
    ]


class RtWork(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in (
        "primary_rays", "bounce_rays", "shadow_rays", "closest_tri_tests", "closest_sphere_tests",
        "shadow_tri_tests", "shadow_sphere_tests", "lit_hits")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class RtBandCopy(C.Structure):
    _fields_ = [("op", C.c_int32), ("reserved", C.c_int32)] + [(k, C.c_uint64) for k in (
        "dst_offset", "dst_pitch", "src_offset", "src_pitch", "width_bytes", "rows")]


RT_COPY_2D, RT_COPY_PEER, RT_COPY_LINEAR = 0, 1, 2


class RtTriangle(C.Structure):
    _fields_ = [(k, C.c_float * 4) for k in ("v0", "v1", "v2", "normal", "color")]


# kernels.cl:8-10 — the two live spheres of the reference (the third initialiser is dropped by SPHERES 2)
REFERENCE_SPHERES = (
    ((0.3, 0.1, -0.5), 0.075, (0.0, 0.0, 0.0, -1.0)),   # glass
    ((-0.4, 0.8, -0.5), 0.05, (0.0, 0.0, 0.0, 0.0)),    # mirror
)


def make_config(width=1024, height=1024, aa_x=2, aa_y=2, shadow_samples=10, light_spread=0.05,
                max_bounces=10, spheres=REFERENCE_SPHERES, band_rows=0, band_index=0, band_count=1,
                device=-1, flags=0, devices=(), device_band_rows=0):
    """Reference constants by default (kernels.cl:7-17, :316-317, :343); same as rt_config_default()."""
    cfg = RtConfig()
    cfg.width, cfg.height, cfg.aa_x, cfg.aa_y = width, height, aa_x, aa_y
    cfg.shadow_samples, cfg.light_spread, cfg.max_bounces = shadow_samples, light_spread, max_bounces
    spheres = tuple(spheres or ())
    if len(spheres) > RT_MAX_SPHERES:
        raise ValueError("at most %d spheres" % RT_MAX_SPHERES)
    cfg.num_spheres = len(spheres)
    for i, (c, r2, col) in enumerate(spheres):
        cfg.spheres[i].center[:] = c
        cfg.spheres[i].radius_sq = r2
        cfg.spheres[i].color[:] = col
    cfg.band_rows = band_rows if band_rows > 0 else height
    cfg.band_index, cfg.band_count = band_index, band_count
    cfg.device, cfg.flags = device, flags
    devices = tuple(devices or ())
    if len(devices) > RT_MAX_DEVICES:
        raise ValueError("at most %d devices" % RT_MAX_DEVICES)
    cfg.num_devices = len(devices)
    for i, d in enumerate(devices):
        cfg.devices[i] = d
    cfg.device_band_rows = device_band_rows
    return cfg
