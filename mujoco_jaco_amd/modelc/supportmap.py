"""Direction-indexed support tables for the convex hulls (model compile time).

A support query `argmax_v v.d` over a hull only ever returns a vertex whose normal cone contains d.  The directions are
binned into the cells of a cube map (6 faces x R x R); for every cell the table lists, in vertex-index order, every
vertex that can be the maximiser for some direction of the cell (bounding-cone test, then dominance pruning; both keep a
superset), padded to 64 entries of
(x, y, z, index).  The kernels then answer a query with ONE 16-byte load per lane instead of scanning the whole hull
(300-1500 vertices), and the result is identical to the full scan, ties included (lowest index wins: entries are in
index order).  `verify` checks exactly that on random and cell-boundary directions in float32.
"""
import numpy as np
from scipy.spatial import ConvexHull

SLOTS = 64
SLACK = 2e-3  # rad: covers float32 rounding of the direction -> cell classification and of near-tied dot products


def cube_cell(d, R):
    """Cell index of direction(s) d [..., 3]; the kernels use the same rule (collision.h: cube_cell)."""
    d = np.asarray(d, dtype=np.float64)
    a = np.abs(d)
    ax, ay, az = a[..., 0], a[..., 1], a[..., 2]
    fx = (ax >= ay) & (ax >= az)
    fy = ~fx & (ay >= az)
    face = np.where(fx, np.where(d[..., 0] > 0, 0, 1), np.where(fy, np.where(d[..., 1] > 0, 2, 3), np.where(d[..., 2] > 0, 4, 5)))
    m = np.where(fx, ax, np.where(fy, ay, az))
    m = np.where(m < 1e-30, 1.0, m)
    u = np.where(fx, d[..., 1], d[..., 0]) / m
    v = np.where(fx | fy, d[..., 2], d[..., 1]) / m
    iu = np.clip(np.floor((u + 1) * 0.5 * R).astype(int), 0, R - 1)
    iv = np.clip(np.floor((v + 1) * 0.5 * R).astype(int), 0, R - 1)
    return (face * R + iu) * R + iv


def _cell_cones(R):
    """Bounding cone (axis, half-angle) of every cell."""
    axes, angs = [], []
    for face in range(6):
        for iu in range(R):
            for iv in range(R):
                def point(u, v):
                    s = 1.0 if face % 2 == 0 else -1.0
                    if face < 2:
                        return np.array([s, u, v])
                    if face < 4:
                        return np.array([u, s, v])
                    return np.array([u, v, s])
                u0, u1 = -1 + 2.0 * iu / R, -1 + 2.0 * (iu + 1) / R
                v0, v1 = -1 + 2.0 * iv / R, -1 + 2.0 * (iv + 1) / R
                c = point(0.5 * (u0 + u1), 0.5 * (v0 + v1)); c /= np.linalg.norm(c)
                corners = [point(u, v) for u in (u0, u1) for v in (v0, v1)]
                ang = max(np.arccos(np.clip(np.dot(c, p / np.linalg.norm(p)), -1, 1)) for p in corners)
                axes.append(c); angs.append(ang)
    return np.array(axes), np.array(angs)


def _vertex_cones(V):
    """Bounding cone of every vertex's normal cone (hull of the outward normals of its incident facets)."""
    hull = ConvexHull(V)
    assert len(hull.vertices) == len(V), "input must be the hull's own vertex set"
    nrm = hull.equations[:, :3]
    inc = [[] for _ in range(len(V))]
    for f, simplex in enumerate(hull.simplices):
        for v in simplex:
            inc[v].append(f)
    axes, angs = np.zeros((len(V), 3)), np.zeros(len(V))
    for v in range(len(V)):
        N = nrm[inc[v]]
        a = N.sum(0)
        na = np.linalg.norm(a)
        if na < 1e-9:
            axes[v], angs[v] = [1, 0, 0], np.pi
            continue
        a /= na
        ang = np.arccos(np.clip(N @ a, -1, 1)).max()
        axes[v], angs[v] = a, (ang if ang < 0.5 * np.pi - 1e-6 else np.pi)   # caps >= 90 deg are not convex: "everywhere"
    return axes, angs


def build(V):
    """V: hull vertices [n, 3] (float64).  Returns (R, table float32 [6*R*R*SLOTS, 4]) or (0, None) if no R fits."""
    V = np.asarray(V, dtype=np.float64)
    vax, vang = _vertex_cones(V)
    V32 = V.astype(np.float32)
    for R in (4, 6, 8, 12):
        cax, cang = _cell_cones(R)
        ang = np.arccos(np.clip(cax @ vax.T, -1, 1))             # [cells, n]
        member = ang <= (cang[:, None] + vang[None, :] + SLACK)
        # dominance pruning: v can never be the maximiser inside the cell if some w beats it for EVERY direction of the
        # cell's bounding cone, i.e. (w - v) makes an angle < 90 deg - theta_C with the cone axis (strict, so ties stay)
        for c in range(len(cax)):
            idx = np.nonzero(member[c])[0]
            if len(idx) <= 1:
                continue
            P = V[idx]
            diff = P[None, :, :] - P[:, None, :]                   # diff[v, w] = w - v
            along = diff @ cax[c]
            dist = np.linalg.norm(diff, axis=2)
            beaten = (along > dist * np.sin(min(cang[c] + SLACK, 0.5 * np.pi)) + 1e-12).any(1)
            member[c, idx[beaten]] = False
        if member.sum(1).max() > SLOTS:
            continue
        table = np.zeros((len(cax) * SLOTS, 4), np.float32)
        ids = np.full(len(cax) * SLOTS, -1, np.int32)
        for c in range(len(cax)):
            idx = np.nonzero(member[c])[0]                        # ascending vertex index
            table[c * SLOTS:c * SLOTS + len(idx), :3] = V32[idx]
            ids[c * SLOTS:c * SLOTS + len(idx)] = idx
        table[:, 3] = ids.view(np.float32)                         # index as raw bits; -1 marks an empty slot
        return R, table
    return 0, None


def verify(V, R, table, ndir=20000, seed=0):
    """float32 support query through the table == full scan (first maximum), on random and cell-boundary directions."""
    V32 = np.asarray(V, dtype=np.float64).astype(np.float32)
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(ndir, 3))
    # directions hugging the cell boundaries and the cube-face seams
    k = rng.integers(0, R + 1, size=(ndir // 2, 2)) * (2.0 / R) - 1 + rng.normal(scale=1e-6, size=(ndir // 2, 2))
    b = np.concatenate([np.ones((ndir // 2, 1)), k], axis=1)
    perm = [rng.permutation(3) for _ in range(ndir // 2)]
    b = np.array([row[p] for row, p in zip(b, perm)]) * rng.choice([-1, 1], size=(ndir // 2, 3))
    d = np.concatenate([d, b]).astype(np.float32)
    cells = cube_cell(d.astype(np.float64), R)
    ids = table[:, 3].copy().view(np.int32).reshape(-1, SLOTS)
    T = table[:, :3].reshape(-1, SLOTS, 3)
    full = ((V32[None, :, 0] * d[:, None, 0] + V32[None, :, 1] * d[:, None, 1]).astype(np.float32) + V32[None, :, 2] * d[:, None, 2]).astype(np.float32)
    ref = full.argmax(1)                                           # first maximum
    tv = T[cells]
    t = ((tv[:, :, 0] * d[:, None, 0] + tv[:, :, 1] * d[:, None, 1]).astype(np.float32) + tv[:, :, 2] * d[:, None, 2]).astype(np.float32)
    t = np.where(ids[cells] >= 0, t, -np.inf)
    got = ids[cells][np.arange(len(d)), t.argmax(1)]
    bad = np.nonzero(got != ref)[0]
    return len(bad), (d[bad[:3]], got[bad[:3]], ref[bad[:3]]) if len(bad) else None
