"""CPU: the sanitizer build (make -C oracle asan: -fsanitize=address,undefined, halt on the first report) of the oracle
restatement and of the product's host-side scene reader (uob_raytracer_amd/csrc/scene.cpp, which parses untrusted OBJ
files): a 64x48 frame through every function of the oracle, and an OBJ with a 5000-character `f` record, a 70-gon, slash
tokens, relative indices, plus four malformed files.  The sanitized frame must be the frame of the ordinary build."""
import os
import subprocess

import numpy as np

from conftest import ROOT
from oracle import pyref
from uob_raytracer_amd import abi, runtime as rt


def test_sanitizer_build_is_clean_and_agrees(tmp_path, scene, oracle):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="halt_on_error=1:detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([os.path.join(ROOT, "oracle", "asan_check"), str(tmp_path)], capture_output=True, text=True, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr
    assert "143 triangles from the hostile file, 4 malformed files rejected" in res.stdout
    # the same 64x48 frame from the ordinary build of the oracle
    cfg = abi.make_config(width=64, height=48)
    v, n, c = scene.packed()
    argb, _ = oracle.render(cfg, v, n, c, rt.rotation_matrix(0.2, -0.1), [0.1, 0.0, -3.0], [-0.2, -0.5, -0.7], 1100.0 * 48 / 1024 * 2)
    assert ("fnv %016x" % pyref.fnv1a64_words(argb)) in res.stdout


def test_obj_reader_long_records(tmp_path):
    """The product's reader through the C ABI: an `f` record far beyond any fixed line buffer, and an n-gon beyond 64 corners."""
    path = tmp_path / "big.obj"
    with open(path, "w") as f:
        for i in range(300):
            f.write("v %d %d %d\n" % (i % 11, i % 7, i % 3))
        f.write("f " + " ".join("%d/%d/%d" % (i, 10 ** 12 + i, 10 ** 12 + i) for i in range(1, 301)) + "\n")     # ~9000 characters, 300-gon
    tris = rt.Scene.load_obj(str(path))
    assert len(tris) == 298
    # fan triangulation in order: triangle j = (1, j+2, j+3) of the scaled, negated, translated vertices
    v = lambda i: -1.5 * np.array([i % 11, i % 7, i % 3], np.float32) + np.array([-0.4, 1.15, -0.7], np.float32)
    assert np.allclose(tris.aos[0, 0, :3], v(0)) and np.allclose(tris.aos[297, 2, :3], v(299)) and np.allclose(tris.aos[100, 1, :3], v(101))
