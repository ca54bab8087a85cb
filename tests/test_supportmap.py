"""CPU tier: the hulls' direction-indexed support tables (modelc/supportmap.py) as stored in the compiled model answer
every support query exactly like a scan of the whole hull (float32, first maximum wins), including directions on the
cube-map cell boundaries."""
import numpy as np

from mujoco_jaco_amd.modelc import supportmap


def test_tables_in_model_match_full_scan(model_arrays):
    M = model_arrays
    cnt, ids = M["f_hullmap_cnt"], M["f_hullmap_ids"]
    start = np.concatenate([[0], np.cumsum(cnt)])
    mv = M["mesh_vert"].reshape(-1, 3)
    seen, with_table = set(), 0
    for g in range(int(M["f_ngeom"][0])):
        R, c0, adr, n = int(M["f_geom_cellR"][g]), int(M["f_geom_cell0"][g]), int(M["f_geom_vertadr"][g]), int(M["f_geom_vertnum"][g])
        if n == 0 or (adr, R) in seen:
            continue
        seen.add((adr, R))
        if R == 0:
            continue
        with_table += 1
        V = mv[adr:adr + n]
        table = np.zeros((6 * R * R * supportmap.SLOTS, 4), np.float32)
        tid = np.full(6 * R * R * supportmap.SLOTS, -1, np.int32)
        for c in range(6 * R * R):
            row = ids[start[c0 + c]:start[c0 + c + 1]]
            assert len(row) <= supportmap.SLOTS and np.all(np.diff(row) > 0)           # index order: ties resolve like a scan
            table[c * supportmap.SLOTS:c * supportmap.SLOTS + len(row), :3] = V[row].astype(np.float32)
            tid[c * supportmap.SLOTS:c * supportmap.SLOTS + len(row)] = row
        table[:, 3] = tid.view(np.float32)
        nbad, example = supportmap.verify(V, R, table, ndir=4000, seed=g)
        assert nbad == 0, (g, example)
    assert with_table >= 8            # every link / finger hull that can collide has one


def test_cube_cell_covers_all_cells():
    for R in (4, 6, 8):
        d = np.random.default_rng(R).normal(size=(200000, 3))
        cells = supportmap.cube_cell(d, R)
        assert cells.min() == 0 and cells.max() == 6 * R * R - 1 and len(np.unique(cells)) == 6 * R * R
