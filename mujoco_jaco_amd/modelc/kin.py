"""Plain numpy fp64 kinematics / mass matrix on the raw model (compiler-side only).

Used at compile time for the qpos0-dependent constants MuJoCo derives in its model
compiler [EXT]: body_invweight0, dof_invweight0, meaninertia.  Deliberately the
dumbest possible formulation (Jacobian sums, dense inverse) so that it can serve as
an independent cross-check of the C oracle's recursive algorithms in tests/.
"""
import numpy as np

from . import rot

JNT_FREE, JNT_HINGE = 0, 3


def fk(M, qpos, mocap_pos=None, mocap_quat=None):
    nbody = int(M["nbody"][0])
    xpos = np.zeros((nbody, 3))
    xquat = np.zeros((nbody, 4))
    xquat[0, 0] = 1
    bpos = M["body_pos"].reshape(-1, 3)
    bquat = M["body_quat"].reshape(-1, 4)
    jpos = M["jnt_pos"].reshape(-1, 3)
    jaxis = M["jnt_axis"].reshape(-1, 3)
    if mocap_pos is None:
        mocap_pos = M["mocap_pos0"].reshape(-1, 3)
        mocap_quat = M["mocap_quat0"].reshape(-1, 4)
    xanchor = np.zeros((int(M["njnt"][0]), 3))
    xaxis = np.zeros((int(M["njnt"][0]), 3))
    for b in range(1, nbody):
        p = M["body_parentid"][b]
        mid = M["body_mocapid"][b]
        if mid >= 0:
            xpos[b], xquat[b] = mocap_pos[mid], rot.quat_normalize(mocap_quat[mid])
            continue
        jn, ja = M["body_jntnum"][b], M["body_jntadr"][b]
        if jn == 1 and M["jnt_type"][ja] == JNT_FREE:
            qa = M["jnt_qposadr"][ja]
            xpos[b] = qpos[qa:qa + 3]
            xquat[b] = rot.quat_normalize(qpos[qa + 3:qa + 7])
            xanchor[ja], xaxis[ja] = xpos[b], np.array([0, 0, 1.0])
            continue
        pos = xpos[p] + rot.rot_vec(xquat[p], bpos[b])
        quat = rot.quat_mul(xquat[p], bquat[b])
        for j in range(ja, ja + jn):
            assert M["jnt_type"][j] == JNT_HINGE
            ang = qpos[M["jnt_qposadr"][j]] - M["qpos0"][M["jnt_qposadr"][j]]
            xanchor[j] = pos + rot.rot_vec(quat, jpos[j])
            xaxis[j] = rot.rot_vec(quat, jaxis[j])
            quat = rot.quat_mul(quat, rot.axis_angle_quat(jaxis[j], ang))
            pos = xanchor[j] - rot.rot_vec(quat, jpos[j])
        xpos[b], xquat[b] = pos, rot.quat_normalize(quat)
    return xpos, xquat, xanchor, xaxis


def jac_point(M, xpos, xquat, xanchor, xaxis, body, point):
    """3 x nv translational and rotational Jacobians of a world point moving with `body`."""
    nv = int(M["nv"][0])
    jp, jr = np.zeros((3, nv)), np.zeros((3, nv))
    b = body
    while b > 0:
        ja, jn = M["body_jntadr"][b], M["body_jntnum"][b]
        for j in range(ja, ja + jn):
            d = M["jnt_dofadr"][j]
            if M["jnt_type"][j] == JNT_FREE:
                R = rot.quat_to_mat(xquat[b])
                jp[:, d:d + 3] = np.eye(3)
                for k in range(3):
                    jr[:, d + 3 + k] = R[:, k]
                    jp[:, d + 3 + k] = np.cross(R[:, k], point - xpos[b])
            else:
                jr[:, d] = xaxis[j]
                jp[:, d] = np.cross(xaxis[j], point - xanchor[j])
        b = M["body_parentid"][b]
    return jp, jr


def mass_matrix(M, qpos):
    nv, nbody = int(M["nv"][0]), int(M["nbody"][0])
    xpos, xquat, xanchor, xaxis = fk(M, qpos)
    H = np.zeros((nv, nv))
    inertia = M["body_inertia"].reshape(-1, 3, 3)
    ipos = M["body_ipos"].reshape(-1, 3)
    jacs = {}
    for b in range(1, nbody):
        R = rot.quat_to_mat(xquat[b])
        com = xpos[b] + R @ ipos[b]
        jp, jr = jac_point(M, xpos, xquat, xanchor, xaxis, b, com)
        jacs[b] = (jp, jr)
        m = M["body_mass"][b]
        if m > 0:
            H += m * jp.T @ jp + jr.T @ (R @ inertia[b] @ R.T) @ jr
    return H, jacs


def invweights(M):
    """body_invweight0 [nbody,2], dof_invweight0 [nv], meaninertia at qpos0."""
    nv, nbody = int(M["nv"][0]), int(M["nbody"][0])
    H, jacs = mass_matrix(M, M["qpos0"])
    Hinv = np.linalg.inv(H)
    biw = np.zeros((nbody, 2))
    for b in range(1, nbody):
        if M["body_weldid"][b] == 0:
            continue
        jp, jr = jacs[b]
        biw[b, 0] = np.trace(jp @ Hinv @ jp.T) / 3
        biw[b, 1] = np.trace(jr @ Hinv @ jr.T) / 3
    diw = np.zeros(nv)
    for j in range(int(M["njnt"][0])):
        d = M["jnt_dofadr"][j]
        if M["jnt_type"][j] == JNT_FREE:
            diw[d:d + 3] = np.trace(Hinv[d:d + 3, d:d + 3]) / 3
            diw[d + 3:d + 6] = np.trace(Hinv[d + 3:d + 6, d + 3:d + 6]) / 3
        else:
            diw[d] = Hinv[d, d]
    return biw, diw, float(np.trace(H) / nv)
