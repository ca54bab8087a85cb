"""Mesh asset processing for the model compiler.

Restates what MuJoCo's (absent, closed-source 2.0-era) compiler does to an STL asset
before the physics ever sees it [EXT: recalled from the public >=2.1 source, legacy
(non-"exactmeshinertia") path, which is what the reference ran]:

  1. read binary STL, weld repeated vertices;
  2. area-weighted centroid of the triangle centres -> pyramid apex;
  3. volume / centre of mass from |pyramid| volumes (legacy: absolute value);
  4. recentre on the CoM, inertia tensor from the same pyramids, rotate the vertex
     set into the principal frame; the mesh then carries (pos, quat) that is composed
     into every geom that instantiates it;
  5. convex hull (qhull) -- collision uses only the hull vertices
     (``convexhull="true"``, jaco2_curtain_torque.xml:3).
"""
import struct

import numpy as np
from scipy.spatial import ConvexHull

from . import rot


def load_stl(path):
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    if 84 + 50 * n != len(b):
        raise ValueError("only binary STL supported: %s" % path)
    rec = np.frombuffer(b, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), offset=84, count=n)
    tri = rec["v"].astype(np.float64)  # [n,3,3]
    flat = tri.reshape(-1, 3)
    uniq, inv = np.unique(flat.astype(np.float32), axis=0, return_inverse=True)
    faces = inv.reshape(-1, 3)
    return uniq.astype(np.float64), faces


def _pyramid_props(vert, faces):
    v0, v1, v2 = vert[faces[:, 0]], vert[faces[:, 1]], vert[faces[:, 2]]
    nrm = np.cross(v1 - v0, v2 - v0)
    area2 = np.linalg.norm(nrm, axis=1)
    ok = area2 > 1e-16
    v0, v1, v2, nrm, area2 = v0[ok], v1[ok], v2[ok], nrm[ok], area2[ok]
    area = area2 / 2
    nrm = nrm / area2[:, None]
    cen = (v0 + v1 + v2) / 3
    return v0, v1, v2, nrm, area, cen


def process_mesh(path, scale=(1, 1, 1)):
    vert, faces = load_stl(path)
    vert = vert * np.asarray(scale, dtype=np.float64)
    v0, v1, v2, nrm, area, cen = _pyramid_props(vert, faces)
    facecen = (cen * area[:, None]).sum(0) / area.sum()
    vol = np.abs(((cen - facecen) * nrm).sum(1) * area / 3)  # legacy: |volume|
    volume = vol.sum()
    com = (vol[:, None] * (0.75 * cen + 0.25 * facecen)).sum(0) / volume
    # recentre, inertia products from pyramids with apex at the CoM
    vert = vert - com
    v0, v1, v2, nrm, area, cen = _pyramid_props(vert, faces)
    vol = np.abs((cen * nrm).sum(1) * area / 3)
    P = np.zeros((3, 3))
    for a in range(3):
        for b in range(3):
            P[a, b] = (vol / 20 * (
                2 * (v0[:, a] * v0[:, b] + v1[:, a] * v1[:, b] + v2[:, a] * v2[:, b])
                + v0[:, a] * v1[:, b] + v0[:, b] * v1[:, a]
                + v0[:, a] * v2[:, b] + v0[:, b] * v2[:, a]
                + v1[:, a] * v2[:, b] + v1[:, b] * v2[:, a])).sum()
    inertia = np.eye(3) * np.trace(P) - P  # unit density, about CoM
    w, V = np.linalg.eigh(inertia)
    order = np.argsort(-w)  # principal moments, descending
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    quat = rot.mat_to_quat(V)
    R = rot.quat_to_mat(quat)
    vert = vert @ R  # into the principal frame
    hull = ConvexHull(vert)
    hv = vert[np.sort(hull.vertices)]
    return {
        "pos": com, "quat": quat, "volume": volume, "inertia": w,  # unit density
        "nvert": len(vert), "nface": len(faces),
        "hull_vert": hv, "rbound": float(np.linalg.norm(hv, axis=1).max()),
        "aabb": np.abs(hv).max(0),
    }
