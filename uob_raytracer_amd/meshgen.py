"""Synthetic Wavefront OBJ meshes in the only syntax the reference's loader accepts (`v x y z`, `f a b c`,
Loader.cpp:39-46).  The reference's call site loads "Source/bunny_200.obj" (skeleton.cpp:102), a file that
is not in the reference repository, so tests and benchmarks write their own meshes with this module."""
import math


def write_sphere_obj(path, n_lon=32, n_lat=24, radius=0.13, center=(0.0, 0.25, 0.0), bumps=0.15, outward=True):
    """A bumpy UV sphere of 2*n_lon*(n_lat-1) triangles.  In OBJ space; load_obj scales by 1.5, negates and
    translates by (-0.4, 1.15, -0.7) (Loader.cpp:42,48-52), which puts this default on the floor of the box.
    outward=True: faces wound counter-clockwise seen from outside, the OBJ convention — the loader's normals
    (cross(e2, e1) of the negated vertices, TestModelH.h:33-37) then point out of the mesh and its light-facing side
    is lit.  outward=False: the opposite winding (normals into the mesh, every visible point faces away from the
    light or is in the mesh's own shadow): the orientation of this generator before round 3, kept for the tests."""
    verts = []
    for j in range(n_lat + 1):
        th = math.pi * j / n_lat
        for i in range(n_lon):
            ph = 2.0 * math.pi * i / n_lon
            r = radius * (1.0 + bumps * math.sin(5 * ph) * math.sin(4 * th))
            verts.append((center[0] + r * math.sin(th) * math.cos(ph), center[1] + r * math.cos(th),
                          center[2] + r * math.sin(th) * math.sin(ph)))
    faces = []
    for j in range(n_lat):
        for i in range(n_lon):
            a = j * n_lon + i
            b = j * n_lon + (i + 1) % n_lon
            c = a + n_lon
            d = b + n_lon
            if j > 0:
                faces.append((a + 1, b + 1, c + 1) if outward else (a + 1, c + 1, b + 1))
            if j < n_lat - 1:
                faces.append((b + 1, d + 1, c + 1) if outward else (b + 1, c + 1, d + 1))
    with open(path, "w") as f:
        f.write("# synthetic bumpy sphere: %d vertices, %d faces\n" % (len(verts), len(faces)))
        for v in verts:
            f.write("v %.7f %.7f %.7f\n" % v)
        for t in faces:
            f.write("f %d %d %d\n" % t)
    return len(faces)


def write_cubesphere_obj(path, n=91, radius=0.13, center=(0.0, 0.25, 0.0), bumps=0.15):
    """The same bumpy sphere from a cube's six n x n grids pushed out to it: 12 n^2 triangles of similar size and shape — no
    polar slivers (a UV sphere's triangles next to its poles have an aspect of 40:1 at 100 000 triangles).  Wound outward."""
    verts, index, faces = [], {}, []

    def vid(face, i, j):
        # cube-surface point of grid node (i, j) of a face, shared along the cube's edges through its rounded coordinates
        a, b = 2.0 * i / n - 1.0, 2.0 * j / n - 1.0
        p = [(1.0, a, b), (-1.0, b, a), (b, 1.0, a), (a, -1.0, b), (a, b, 1.0), (b, a, -1.0)][face]
        key = (round(p[0] * n), round(p[1] * n), round(p[2] * n))
        if key not in index:
            ln = math.sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2])
            d = (p[0] / ln, p[1] / ln, p[2] / ln)
            th, ph = math.acos(max(-1.0, min(1.0, d[1]))), math.atan2(d[2], d[0])
            r = radius * (1.0 + bumps * math.sin(5 * ph) * math.sin(4 * th))
            verts.append((center[0] + r * d[0], center[1] + r * d[1], center[2] + r * d[2]))
            index[key] = len(verts)
        return index[key]

    for face in range(6):
        for i in range(n):
            for j in range(n):
                a, b, c, d = vid(face, i, j), vid(face, i + 1, j), vid(face, i, j + 1), vid(face, i + 1, j + 1)
                faces.append((a, b, d))          # (a, b, d), (a, d, c): counter-clockwise seen from outside on every face
                faces.append((a, d, c))
    with open(path, "w") as f:
        f.write("# synthetic bumpy cube-sphere: %d vertices, %d faces\n" % (len(verts), len(faces)))
        for v in verts:
            f.write("v %.7f %.7f %.7f\n" % v)
        for t in faces:
            f.write("f %d %d %d\n" % t)
    return len(faces)
