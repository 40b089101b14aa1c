"""CPU: the C-ABI library loads, exports every symbol include/uob_rt.h declares, agrees with the ctypes
struct layouts, validates its arguments and fails loudly without a GPU (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from uob_raytracer_amd import abi, runtime as rt

HEADER = os.path.join(ROOT, "include", "uob_rt.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(rt.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), "libuob_rt.so does not export %s" % name
    assert sorted(rt.EXPORTS) == names       # the binding covers exactly the header


def test_struct_layouts_match_the_header(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "uob_rt.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(rt_config),sizeof(rt_sphere),sizeof(rt_work),sizeof(rt_triangle),'
                    'offsetof(rt_config,spheres),offsetof(rt_config,band_rows),offsetof(rt_config,devices),'
                    'offsetof(rt_config,device_band_rows));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(abi.RtConfig), C.sizeof(abi.RtSphere), C.sizeof(abi.RtWork),
                                     C.sizeof(abi.RtTriangle), abi.RtConfig.spheres.offset, abi.RtConfig.band_rows.offset,
                                     abi.RtConfig.devices.offset, abi.RtConfig.device_band_rows.offset]


def test_default_config_is_the_reference_constants():
    cfg = rt.default_config()
    assert (cfg.width, cfg.height, cfg.aa_x, cfg.aa_y) == (1024, 1024, 2, 2)          # kernels.cl:12-17
    assert (cfg.shadow_samples, cfg.max_bounces, cfg.num_spheres) == (10, 10, 2)      # :316, :343, :7
    assert abs(cfg.light_spread - 0.05) < 1e-9
    assert bytes(cfg) == bytes(abi.make_config())
    assert rt.lib().rt_config_owned_rows(C.byref(cfg)) == 1024
    half = abi.make_config(band_rows=32, band_index=1, band_count=2)
    assert rt.lib().rt_config_owned_rows(C.byref(half)) == 512


def _init(cfg, scene):
    v, n, c = scene.packed()
    h = C.c_void_p()
    rc = rt.lib().rt_init(C.byref(cfg), rt._fp(v), rt._fp(n), rt._fp(c), len(scene), C.byref(h))
    return rc, rt.lib().rt_last_error().decode()


@pytest.mark.parametrize("kw", [dict(width=0), dict(aa_x=0), dict(shadow_samples=0), dict(max_bounces=-1),
                                dict(band_index=2, band_count=2, band_rows=4), dict(width=8192, height=8192),
                                dict(light_spread=-1.0), dict(flags=1), dict(devices=(0, 0), band_rows=8, band_count=2),
                                dict(device_band_rows=-1)])
def test_invalid_configurations_are_rejected_before_touching_the_device(kw, scene):
    rc, msg = _init(abi.make_config(**kw), scene)
    assert rc in (abi.RT_E_INVALID, abi.RT_E_UNSUPPORTED) and msg


def test_scene_coordinate_bound(scene):
    big = rt.Scene(scene.aos.copy())
    big.aos[0, 0, 0] = 3.0e6
    rc, msg = _init(abi.make_config(), big)
    assert rc == abi.RT_E_INVALID and "2^16" in msg
    nan = rt.Scene(scene.aos.copy())
    nan.aos[3, 1, 2] = np.nan
    assert _init(abi.make_config(), nan)[0] == abi.RT_E_INVALID


def test_no_gpu_means_a_loud_failure_not_a_fallback(scene):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.RayTracer(abi.make_config(width=64, height=64), scene)
    assert e.value.code == abi.RT_E_DEVICE


def test_null_arguments():
    L = rt.lib()
    assert L.rt_init(None, None, None, None, 0, None) == abi.RT_E_INVALID
    assert L.rt_render(None, None, None, None, C.c_float(1.0), None, None) == abi.RT_E_INVALID
    assert L.rt_scene_cornell_box(None, 0) == abi.RT_E_INVALID
    L.rt_destroy(None)      # must be a no-op


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under uob_raytracer_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "uob_raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "rt_oracle" not in text and "pyref" not in text and "oracle/" not in text, os.path.join(dirpath, f)
