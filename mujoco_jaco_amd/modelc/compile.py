"""Model compiler: reference MJCF + STL  ->  committed model-constant blobs.

  python -m mujoco_jaco_amd.modelc.compile            # needs /root/reference (not on the GPU box)

Writes mujoco_jaco_amd/assets/<model>.jacomdl holding two views of the same model:

* raw view (MuJoCo-style arrays, all 56 bodies) -- consumed by the fp64 oracle;
* fused view (prefix ``f_``): welded bodies merged into 11 moving bodies + static
  world, per-pair contact parameters pre-mixed, collision-pair whitelist -- consumed
  by the HIP library.

Collision filter and parameter mixing follow SURVEY.md App. C / D.1 step 3 [EXT].
"""
import os
import sys

import numpy as np

from . import supportmap, blob, kin, mjcf, rot

REF_ASSETS = "/root/reference/env_script/assets/jaco2"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def collision_pairs(M):
    ng = int(M["ngeom"][0])
    weld, parent = M["body_weldid"], M["body_parentid"]
    gb, ct, ca = M["geom_bodyid"], M["geom_contype"], M["geom_conaffinity"]
    pairs = []
    for g1 in range(ng):
        for g2 in range(g1 + 1, ng):
            w1, w2 = weld[gb[g1]], weld[gb[g2]]
            if w1 == w2:
                continue
            if w1 != 0 and w2 != 0:
                if weld[parent[w1]] == w2 or weld[parent[w2]] == w1:
                    continue
            if not ((ct[g1] & ca[g2]) or (ct[g2] & ca[g1])):
                continue
            pairs.append((g1, g2))
    return np.array(pairs, dtype=np.int32).reshape(-1, 2)


def _geom_points(M, g, xpos, xquat):
    """World-frame extreme points of a convex geom at the pose given by its body frame (mesh: hull vertices)."""
    b = M["geom_bodyid"][g]
    R = rot.quat_to_mat(xquat[b]) @ rot.quat_to_mat(M["geom_quat"].reshape(-1, 4)[g])
    p = xpos[b] + rot.quat_to_mat(xquat[b]) @ M["geom_pos"].reshape(-1, 3)[g]
    t, sz = M["geom_type"][g], M["geom_size"].reshape(-1, 3)[g]
    if t == mjcf.GEOM_MESH:
        md = M["geom_dataid"][g]
        v = M["mesh_vert"].reshape(-1, 3)[M["mesh_vertadr"][md]:M["mesh_vertadr"][md] + M["mesh_vertnum"][md]]
        return p + v @ R.T, 0.0
    if t == mjcf.GEOM_BOX:
        c = np.array([[sx, sy, sz_] for sx in (-1, 1) for sy in (-1, 1) for sz_ in (-1, 1)]) * sz
        return p + c @ R.T, 0.0
    if t == mjcf.GEOM_SPHERE:
        return p[None, :], float(sz[0])
    if t == mjcf.GEOM_CYLINDER:   # the corners of the enclosing box (r, r, h): a superset, which is what a "can never touch" proof may use
        c = np.array([[sx, sy, sz_] for sx in (-1, 1) for sy in (-1, 1) for sz_ in (-1, 1)]) * np.array([sz[0], sz[0], sz[1]])
        return p + c @ R.T, 0.0
    return None, 0.0


def prune_static_pairs(M, pairs):
    """Drop pairs that provably can never touch: a geom whose body moves only by rotation about one fixed world
    axis sweeps a solid of revolution; if that solid is separated from a *static* geom along the axis, or radially,
    the pair is dead for every configuration.  Exact (collision results are unchanged); it removes the permanent
    0.2 mm near-miss link0<->link1 hull pair and its siblings from the per-step narrowphase."""
    xpos, xquat, xanchor, xaxis = kin.fk(M, M["qpos0"])
    weld, parent = M["body_weldid"], M["body_parentid"]
    keep, dropped = [], []
    for g1, g2 in pairs:
        b1, b2 = M["geom_bodyid"][g1], M["geom_bodyid"][g2]
        w1, w2 = weld[b1], weld[b2]
        if (w1 == 0) == (w2 == 0):
            keep.append((g1, g2)); continue
        gs, gm, wm = (g1, g2, w2) if w1 == 0 else (g2, g1, w1)
        if M["body_mocapid"][M["geom_bodyid"][gs]] >= 0:
            keep.append((g1, g2)); continue   # mocap geoms move at run time
        # moving body must hang off the static world through exactly one hinge
        if weld[parent[wm]] != 0 or M["body_jntnum"][wm] != 1 or M["jnt_type"][M["body_jntadr"][wm]] != mjcf.JNT_HINGE:
            keep.append((g1, g2)); continue
        j = M["body_jntadr"][wm]
        a, c = xaxis[j], xanchor[j]
        pm, rm = _geom_points(M, gm, xpos, xquat)
        if pm is None:
            keep.append((g1, g2)); continue
        hm = (pm - c) @ a
        radm = np.linalg.norm((pm - c) - np.outer(hm, a), axis=1).max() + rm
        hm_lo, hm_hi = hm.min() - rm, hm.max() + rm
        if M["geom_type"][gs] == mjcf.GEOM_PLANE:
            b = M["geom_bodyid"][gs]
            n = (rot.quat_to_mat(xquat[b]) @ rot.quat_to_mat(M["geom_quat"].reshape(-1, 4)[gs]))[:, 2]
            p0 = xpos[b] + rot.quat_to_mat(xquat[b]) @ M["geom_pos"].reshape(-1, 3)[gs]
            # lowest point of the swept solid above the plane, valid when the plane normal is the rotation axis
            if abs(abs(n @ a) - 1) < 1e-9:
                lo = ((pm - p0) @ n).min() - rm
                if lo > 1e-6:
                    dropped.append((g1, g2)); continue
            keep.append((g1, g2)); continue
        ps, rs = _geom_points(M, gs, xpos, xquat)
        if ps is None:
            keep.append((g1, g2)); continue
        hs = (ps - c) @ a
        if hs.min() - rs > hm_hi + 1e-6 or hs.max() + rs < hm_lo - 1e-6:
            dropped.append((g1, g2)); continue
        # radial: the swept solid lies inside the cylinder of radius radm about the axis; if some half-space
        # {x : perp(x).u > radm} contains the whole static geom, the two are separated for every joint angle
        perp = (ps - c) - np.outer(hs, a)
        cen = perp.mean(0)
        d = np.linalg.norm(cen)
        if d > 1e-9:
            u = cen / d
            if (perp @ u).min() - rs > radm + 1e-6:   # a plane with normal u separates the static geom from the swept cylinder
                dropped.append((g1, g2)); continue
        keep.append((g1, g2))
    return np.array(keep, dtype=np.int32).reshape(-1, 2), dropped


def mix_pair(M, g1, g2, timestep):
    """condim / friction / solref / solimp of a geom pair (equal priority, equal solmix)."""
    condim = max(M["geom_condim"][g1], M["geom_condim"][g2])
    f = np.maximum(M["geom_friction"].reshape(-1, 3)[g1], M["geom_friction"].reshape(-1, 3)[g2])
    solref = 0.5 * (M["geom_solref"].reshape(-1, 2)[g1] + M["geom_solref"].reshape(-1, 2)[g2])
    solimp = 0.5 * (M["geom_solimp"].reshape(-1, 5)[g1] + M["geom_solimp"].reshape(-1, 5)[g2])
    solref = solref.copy()
    solref[0] = max(solref[0], 2 * timestep)  # refsafe
    mu = np.array([f[0], f[0], f[1], f[2], f[2]])
    margin = max(M["geom_margin"][g1], M["geom_margin"][g2])
    return condim, mu, solref, solimp, margin


def fuse(M, names, pairs):
    """Merge weld groups; return dict of f_* arrays."""
    nbody = int(M["nbody"][0])
    weld = M["body_weldid"]
    xpos, xquat, _, _ = kin.fk(M, M["qpos0"])
    roots = [b for b in range(1, nbody) if weld[b] == b]
    fid = {0: -1}
    for i, r in enumerate(roots):
        fid[r] = i
    nmov = len(roots)
    ipos = M["body_ipos"].reshape(-1, 3)
    inertia = M["body_inertia"].reshape(-1, 3, 3)

    def rel(b, w):
        """pose of body b in the frame of its weld root w (constant: no joints in between)."""
        qw = rot.quat_conj(xquat[w])
        return rot.rot_vec(qw, xpos[b] - xpos[w]), rot.quat_normalize(rot.quat_mul(qw, xquat[b]))

    F = {}
    fparent, fpos, fquat, faxis, fjpos, fq0, fmass, fcom, finert = [], [], [], [], [], [], [], [], []
    flim, frange, fdamp, ftype, fqadr, fdadr = [], [], [], [], [], []
    fstiff, fsref = [], []
    for r in roots:
        members = [b for b in range(1, nbody) if weld[b] == r]
        mass = sum(M["body_mass"][b] for b in members)
        com = np.zeros(3)
        for b in members:
            p, q = rel(b, r)
            com += M["body_mass"][b] * (p + rot.rot_vec(q, ipos[b]))
        com /= mass
        I = np.zeros((3, 3))
        for b in members:
            if M["body_mass"][b] <= 0:
                continue
            p, q = rel(b, r)
            R = rot.quat_to_mat(q)
            d = p + R @ ipos[b] - com
            I += R @ inertia[b] @ R.T + M["body_mass"][b] * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        pw = weld[M["body_parentid"][r]]
        fparent.append(fid[pw])
        # frame of r relative to its fused parent at qpos0 (hinge angle 0) == composition of static frames
        if pw == 0:
            p, q = xpos[r] - 0, xquat[r]
            # static ancestors of r may be offset from the world origin; xpos at qpos0 already includes them
        else:
            qw = rot.quat_conj(xquat[pw])
            p, q = rot.rot_vec(qw, xpos[r] - xpos[pw]), rot.quat_normalize(rot.quat_mul(qw, xquat[r]))
        fpos.append(p)
        fquat.append(q)
        j = M["body_jntadr"][r]
        assert M["body_jntnum"][r] == 1
        ftype.append(M["jnt_type"][j])
        faxis.append(M["jnt_axis"].reshape(-1, 3)[j])
        fjpos.append(M["jnt_pos"].reshape(-1, 3)[j])
        fq0.append(M["qpos0"][M["jnt_qposadr"][j]] if M["jnt_type"][j] == mjcf.JNT_HINGE else 0.0)
        fqadr.append(M["jnt_qposadr"][j])
        fdadr.append(M["jnt_dofadr"][j])
        flim.append(M["jnt_limited"][j])
        frange.append(M["jnt_range"].reshape(-1, 2)[j])
        fdamp.append(M["dof_damping"][M["jnt_dofadr"][j]])
        fstiff.append(M["jnt_stiffness"][j]); fsref.append(M["jnt_springref"][j])
        fmass.append(mass)
        fcom.append(com)
        finert.append([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])
    f64 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.float64).reshape(-1))
    i32 = lambda x: np.ascontiguousarray(np.array(x, dtype=np.int32).reshape(-1))
    F["f_nbody"] = i32([nmov])
    F["f_parent"], F["f_jtype"] = i32(fparent), i32(ftype)
    F["f_qposadr"], F["f_dofadr"] = i32(fqadr), i32(fdadr)
    F["f_pos"], F["f_quat"], F["f_axis"], F["f_jpos"] = f64(fpos), f64(fquat), f64(faxis), f64(fjpos)
    F["f_qpos0"], F["f_mass"], F["f_com"], F["f_inertia"] = f64(fq0), f64(fmass), f64(fcom), f64(finert)
    F["f_limited"], F["f_range"], F["f_damping"] = i32(flim), f64(frange), f64(fdamp)
    F["f_stiffness"], F["f_springref"] = f64(fstiff), f64(fsref)

    # ---- geoms that can collide with anything (appear in the pair list), re-indexed
    used = sorted(set(pairs.reshape(-1).tolist()))
    gmap = {g: i for i, g in enumerate(used)}
    biw = M["body_invweight0"].reshape(-1, 2)
    gbody, gpos, gquat, gmocap = [], [], [], []
    for g in used:
        b = M["geom_bodyid"][g]
        w = weld[b]
        gq = M["geom_quat"].reshape(-1, 4)[g]
        gp = M["geom_pos"].reshape(-1, 3)[g]
        mid = M["body_mocapid"][b]
        gmocap.append(mid)
        if mid >= 0:  # pose relative to the mocap body, composed at run time
            gbody.append(-1)
            gpos.append(gp)
            gquat.append(gq)
            continue
        if w == 0:
            p, q = xpos[b], xquat[b]
        else:
            p, q = rel(b, w)
        gbody.append(fid[w])
        gpos.append(p + rot.rot_vec(q, gp))
        gquat.append(rot.quat_normalize(rot.quat_mul(q, gq)))
    F["f_ngeom"] = i32([len(used)])
    F["f_geom_orig"] = i32(used)
    F["f_geom_body"], F["f_geom_mocap"] = i32(gbody), i32(gmocap)
    F["f_geom_origbody"] = i32([M["geom_bodyid"][g] for g in used])
    F["f_geom_type"] = i32([M["geom_type"][g] for g in used])
    F["f_geom_pos"], F["f_geom_quat"] = f64(gpos), f64(gquat)
    F["f_geom_size"] = f64([M["geom_size"].reshape(-1, 3)[g] for g in used])
    F["f_geom_rbound"] = f64([M["geom_rbound"][g] for g in used])
    F["f_geom_vertadr"] = i32([M["mesh_vertadr"][M["geom_dataid"][g]] if M["geom_dataid"][g] >= 0 else 0 for g in used])
    F["f_geom_vertnum"] = i32([M["mesh_vertnum"][M["geom_dataid"][g]] if M["geom_dataid"][g] >= 0 else 0 for g in used])
    # direction-indexed support tables of the hulls (supportmap.py), CSR over the cube-map cells of every mesh; the loader
    # expands them to 64 padded (x, y, z, index) slots per cell.  f_geom_cellR = 0: no table, the kernels scan the hull.
    used_mesh = sorted({int(M["geom_dataid"][g]) for g in used if M["geom_dataid"][g] >= 0})
    mesh_R, mesh_cell0, cnts, ids = {}, {}, [], []
    mv = M["mesh_vert"].reshape(-1, 3)
    for k in used_mesh:
        Vk = mv[M["mesh_vertadr"][k]:M["mesh_vertadr"][k] + M["mesh_vertnum"][k]]
        R, table = supportmap.build(Vk)
        mesh_R[k], mesh_cell0[k] = R, len(cnts)
        if R:
            nbad, _ = supportmap.verify(Vk, R, table)
            assert nbad == 0, "support table of mesh %d disagrees with the full scan" % k
            tid = table[:, 3].copy().view(np.int32).reshape(-1, supportmap.SLOTS)
            for row in tid:
                row = row[row >= 0]
                cnts.append(len(row)); ids.extend(row.tolist())
    F["f_hullmap_cnt"], F["f_hullmap_ids"] = i32(cnts if cnts else [0]), i32(ids if ids else [0])
    F["f_geom_cellR"] = i32([mesh_R[int(M["geom_dataid"][g])] if M["geom_dataid"][g] >= 0 else 0 for g in used])
    F["f_geom_cell0"] = i32([mesh_cell0[int(M["geom_dataid"][g])] if M["geom_dataid"][g] >= 0 else 0 for g in used])
    F["f_geom_invweight"] = f64([biw[M["geom_bodyid"][g]] for g in used])

    # ---- pair table with pre-mixed parameters
    dt = float(M["opt_timestep"][0])
    pp, pdim, pmu, pref, pimp, pmargin = [], [], [], [], [], []
    for g1, g2 in pairs:
        condim, mu, solref, solimp, margin = mix_pair(M, g1, g2, dt)
        pp.append((gmap[g1], gmap[g2]))
        pdim.append(condim)
        pmu.append(mu)
        pref.append(solref)
        pimp.append(solimp)
        pmargin.append(margin)
    F["f_npair"] = i32([len(pairs)])
    F["f_pair_geom"], F["f_pair_condim"] = i32(pp), i32(pdim)
    F["f_pair_mu"], F["f_pair_solref"], F["f_pair_solimp"] = f64(pmu), f64(pref), f64(pimp)
    F["f_pair_margin"] = f64(pmargin)

    # ---- touch sites
    sb, sp, sq = [], [], []
    for s in M["sensor_siteid"]:
        b = M["site_bodyid"][s]
        p, q = rel(b, weld[b])
        sb.append(fid[weld[b]])
        sp.append(p + rot.rot_vec(q, M["site_pos"].reshape(-1, 3)[s]))
        sq.append(rot.quat_normalize(rot.quat_mul(q, M["site_quat"].reshape(-1, 4)[s])))
    F["f_nsensor"] = i32([len(sb)])
    F["f_site_body"], F["f_site_pos"], F["f_site_quat"] = i32(sb), f64(sp), f64(sq)
    F["f_site_type"] = i32([M["site_type"][s] for s in M["sensor_siteid"]])
    F["f_site_size"] = f64([M["site_size"].reshape(-1, 3)[s] for s in M["sensor_siteid"]])
    F["f_site_origbody"] = i32([M["site_bodyid"][s] for s in M["sensor_siteid"]])

    # ---- named frames the task logic reads (env_mujoco_util.py:35,107,310)
    # mocap ids of the markers _take_action moves every env step (env_mujoco_util.py:613-615,644-646); -1 if absent
    F["f_marker_mocap"] = i32([M["body_mocapid"][names["body"].index(nm)] if nm in names["body"] else -1 for nm in ("hand", "subgoal_reach")])
    for nm in ("EE", "EE_obj", "link1", "object_body", "object_dest"):
        if nm in names["body"]:
            b = names["body"].index(nm)
            w = weld[b]
            p, q = (xpos[b], xquat[b]) if w == 0 else rel(b, w)
            F["f_frame_" + nm] = f64(np.concatenate([[fid[w]], p, q]))
    return F


def compile_model(xml_name, timestep=0.001):
    """timestep: the reference overrides model.opt.timestep with 0.001 (mujoco.py:39, env_mujoco_util.py:30)."""
    M, names = mjcf.parse(os.path.join(REF_ASSETS, xml_name + ".xml"), timestep=timestep)
    biw, diw, meaninertia = kin.invweights(M)
    M["body_invweight0"] = np.ascontiguousarray(biw.reshape(-1))
    M["dof_invweight0"] = diw
    M["meaninertia"] = np.array([meaninertia])
    pairs_all = collision_pairs(M)
    pairs, dropped = prune_static_pairs(M, pairs_all)
    M["pair_geom_unpruned"] = np.ascontiguousarray(pairs_all.reshape(-1))   # kept for the oracle-side equivalence test
    M["pair_geom"] = np.ascontiguousarray(pairs.reshape(-1))
    M["npair"] = np.array([len(pairs)], dtype=np.int32)
    M.update(fuse(M, names, pairs))
    return M, names


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    # jaco2_torque: 6 arm + 6 finger hinges (sprung distal joints), no free bodies -- stepped by the d12 build of the library;
    # jaco2_dual_torque: two arms + two objects (30 dofs, 106 geoms, 3 332 pairs) -- stepped by the d30 build (ctrl level);
    # jaco2_curtain_torque_sensor: the 12-hinge arm + one free object + five static cylinders, 61 touch sensors -- d12 build (ctrl level);
    # jaco2_curtain_torque_old: the same arm + object among 22 static boxes, 20 touch sensors -- d12 build (ctrl level)
    for xml_name in ("jaco2_curtain_torque", "jaco2_reaching_torque", "jaco2_torque", "jaco2_dual_torque", "jaco2_curtain_torque_sensor",
                     "jaco2_curtain_torque_old"):
        M, names = compile_model(xml_name)
        path = os.path.join(OUT_DIR, xml_name + ".jacomdl")
        blob.save(path, M)
        with open(os.path.join(OUT_DIR, xml_name + ".names.txt"), "w") as f:
            for k, v in names.items():
                f.write(k + ": " + " ".join(n if n else "-" for n in v) + "\n")
        print(xml_name, "nq", M["nq"][0], "nv", M["nv"][0], "nbody", M["nbody"][0], "ngeom", M["ngeom"][0],
              "npair", M["npair"][0], "f_nbody", M["f_nbody"][0], "f_ngeom", M["f_ngeom"][0],
              "hullverts", len(M["mesh_vert"]) // 3, "bytes", os.path.getsize(path))


if __name__ == "__main__":
    sys.exit(main())
