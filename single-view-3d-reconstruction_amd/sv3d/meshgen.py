"""Synthetic mesh inputs: a subdivided icosahedron on the unit sphere.

Same vertex / face counts as the reference's `utils/mesh.create_sphere`
(utils/mesh.py:12-16: 3 subdivisions -> 642 vertices / 1280 faces, 4 -> 2562 / 5120);
vertex ORDER is this generator's own, which is all the benchmarks need.
"""
import numpy as np


def icosphere(subdivide=3):
    """Returns (verts [V,3] float32 on the unit sphere, faces [F,3] int32, outward winding)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
             (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [np.asarray(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    for _ in range(int(subdivide)):
        cache = {}

        def midpoint(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nxt = []
        for a, b, c in faces:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            nxt += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nxt
    return np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.int32)
