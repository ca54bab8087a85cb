"""TEST INFRASTRUCTURE - NOT PRODUCT CODE.

numpy fp64 restatement of the reference's env-level glue, pinned by
tests/golden/glue_vectors.npz, which was produced by running the reference's own Python
(tests/golden/make_glue_vectors.py).  The operational-space controller (a4) lives in the absent abr_control
package and is restated from its published algorithm: unpinned.
Only tests/ (and __graft_entry__.smoke / bench.py's cpu_baseline leg) may import this module.
Quaternions are [w, x, y, z]; Euler angles are Gohlke 'rxyz' = intrinsic X-Y-Z, R = Rx(a) Ry(b) Rz(c).
"""
import numpy as np

OBJECT_Z = 0.1898  # env_mujoco_util.py:44


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def euler_from_quat(q):
    """transformations.euler_from_quaternion(q, 'rxyz') (env_mujoco_util.py:661-673)."""
    q = np.asarray(q, dtype=np.float64)
    M = quat_to_mat(q / np.linalg.norm(q))
    cy = np.sqrt(M[2, 2] ** 2 + M[1, 2] ** 2)
    if cy > 8.8e-16:
        return np.array([np.arctan2(-M[1, 2], M[2, 2]), np.arctan2(M[0, 2], cy), np.arctan2(-M[0, 1], M[0, 0])])
    return np.array([0.0, np.arctan2(M[0, 2], cy), np.arctan2(M[1, 0], M[1, 1])])


def quat_from_euler(a, b, c):
    """transformations.quaternion_from_euler(a, b, c, 'rxyz') up to sign."""
    def ax(i, t):
        q = np.zeros(4); q[0] = np.cos(t / 2); q[1 + i] = np.sin(t / 2); return q

    def mul(p, q):
        return np.array([p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3], p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2],
                         p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1], p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0]])
    return mul(mul(ax(0, a), ax(1, b)), ax(2, c))


def get_rotation(roll, pitch, yaw, vec, inv=False):
    """_get_rotation (env_mujoco_util.py:448-468)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]) @ np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return (R.T if inv else R) @ np.asarray(vec, dtype=np.float64)


def touch_class(sensordata):
    """_get_touch (env_mujoco_util.py:470-490). sensordata in XML order: EE_touch, 0_touch .. 18_touch."""
    t = np.concatenate([sensordata[1:20], sensordata[0:1]]) > 0.001
    thumb, index, pinky = t[1:5].any(), t[5:9].any(), t[9:13].any()
    if (thumb and index) or (thumb and pinky):
        return 3
    if t[:13].any():
        return 1
    if t[13:].any():
        return 2
    return 0


def reward_picking(ee_pos, ee_quat, obj_pos, touch):
    """_get_reward, task 'picking' (env_mujoco_util.py:392-431)."""
    roll, pitch, yaw = euler_from_quat(ee_quat)
    ee_vec = get_rotation(roll, pitch, yaw, [0, 0, -1], False)
    xyz = np.asarray(obj_pos, dtype=np.float64) - ee_pos
    d = np.linalg.norm(xyz)
    x, y, z = xyz / d
    tp, ty = -np.arccos(z), -np.arccos(-x / np.sqrt(1 - z * z))
    tv = get_rotation(0, tp, ty, [0, 0, 1], True)
    tv[2] *= -1
    ang = np.linalg.norm(ee_vec - tv)
    r = 5 * np.exp(-d / 0.2) / 2 + 2 * np.exp(-ang / (np.pi / 6)) / 2 / (d * 15 + 1)
    r += {1: 0.75, 2: -0.75, 3: 2.5}.get(int(touch), 0.0)
    r += 100 * (obj_pos[2] - OBJECT_Z)
    return r * 0.01


def reward_grasping(ee_pos, ee_quat, obj_pos, touch):
    """_get_reward, task 'grasping' (env_mujoco_util.py:352-391): the picking shape with its own coefficients
    (touch 1 / 2 / 3: +0.75 / -0.75 / +2.5, scale 0.05)."""
    roll, pitch, yaw = euler_from_quat(ee_quat)
    ee_vec = get_rotation(roll, pitch, yaw, [0, 0, -1], False)
    xyz = np.asarray(obj_pos, dtype=np.float64) - ee_pos
    d = np.linalg.norm(xyz)
    x, y, z = xyz / d
    tp, ty = -np.arccos(z), -np.arccos(-x / np.sqrt(1 - z * z))
    tv = get_rotation(0, tp, ty, [0, 0, 1], True)
    tv[2] *= -1
    ang = np.linalg.norm(ee_vec - tv)
    r = 5 * np.exp(-d / 0.2) / 2 + 2 * np.exp(-ang / (np.pi / 6)) / 2 / (d * 15 + 1)
    r += {1: 5 * 0.5 * 0.3, 2: -5 * 0.5 * 0.3, 3: 5 * 0.5}.get(int(touch), 0.0)
    r += 100 * (obj_pos[2] - OBJECT_Z)
    return r * 0.05


def grasp_reach_ori(ee_pos, obj_goal, gamma):
    """Orientation part of the pre-reach target of the grasping / carrying reset (env_mujoco_util.py:124-129): looks along EE -> object,
    yaw = the drawn gamma; stored as float16 like the reference's np.array(..., dtype=np.float16)."""
    x, y, z = np.asarray(obj_goal, np.float64) - np.asarray(ee_pos, np.float64)
    alpha = -np.arcsin(y / np.sqrt(y ** 2 + z ** 2)) * np.sign(x)
    beta = np.arccos(x / np.linalg.norm([x, y, z])) * np.sign(x)
    return np.array([alpha, beta, gamma], dtype=np.float16).astype(np.float64)


def grasp_prereach_conditions(ee_pos, ee_quat, obj_goal, reach_goal_ori):
    """What the two `while True` loops of the grasping reset test after each substep (env_mujoco_util.py:138-167): distance EE -> object
    goal, and |q_EE - q_goal| of the unit quaternions built from the Euler angles (goal = the sampled *reaching* goal's orientation)."""
    dist = np.linalg.norm(np.asarray(ee_pos) - np.asarray(obj_goal))
    grip = quat_from_euler(*euler_from_quat(ee_quat)); grip = grip / np.linalg.norm(grip)
    tar = quat_from_euler(*reach_goal_ori); tar = tar / np.linalg.norm(tar)
    return dist, np.linalg.norm(grip - tar)


def canon_euler(e):
    """euler -> unit quaternion -> euler, as the reaching reward / termination do before differencing (env_mujoco_util.py:323-330)."""
    return euler_from_quat(quat_from_euler(*e))


def reach_ang_diff(ee_quat, goal_euler):
    d = np.linalg.norm(canon_euler(euler_from_quat(ee_quat)) - canon_euler(goal_euler))
    return 2 * np.pi - d if d > np.pi else d


def reward_reaching(ee_pos, ee_quat, reach_goal, base_pos):
    """_get_reward, task 'reaching' (env_mujoco_util.py:314-351)."""
    dist = np.linalg.norm(np.asarray(ee_pos) - reach_goal[:3])
    ang = reach_ang_diff(ee_quat, reach_goal[3:6])
    r = 5 * np.exp(-dist) / 2 + 2 * np.exp(-ang / (np.pi / 6)) / (2 * (dist * 15 + 1))
    wb = np.linalg.norm(np.asarray(ee_pos) - base_pos)
    if wb < 0.15:
        r -= 0.15 - wb
    if ee_pos[2] < 0.1:
        r -= 0.1 - ee_pos[2]
    return 0.05 * r


def sample_reach_goal(u01, base_pos):
    """__sample_goal, reaching part (env_mujoco_util.py:199-207). u01: the draws in the reference's order --
    uniform(0.3, 0.42), choice([-1, 1]) for x, the same for y, uniform(0.3, 0.5) for z, uniform(-0.1, 0.1) for gamma -- given
    as the values the reference's calls returned."""
    mx, sx, my, sy, z, gamma = u01
    pos = np.array([mx * sx, my * sy, z])
    xyz = pos - base_pos
    x, y, z_ = xyz / np.linalg.norm(xyz)
    alpha = -np.arcsin(y / np.sqrt(y ** 2 + z_ ** 2)) * np.sign(x)
    beta = np.arccos(x / np.linalg.norm([x, y, z_])) * np.sign(x)
    return np.hstack([pos, np.array([alpha, beta, gamma], dtype=np.float16)])


def sample_reach_goal_from_buffer(buffer, u):
    """__sample_goal with a goal buffer (kwarg init_buffer, env_mujoco_util.py:46,208-212): random_idx = np.random.randint(0, len(buffer) - 1),
    i.e. one of rows 0 .. len - 2; the goal is buffer[idx][1:4] and [4:7] as they are (no float16 cast in this branch).  u: a uniform [0, 1)
    draw standing for the reference's randint (the build's counter-based RNG supplies it).  Returns (goal[6], idx)."""
    n = len(buffer)
    idx = min(int(u * (n - 1)), n - 2)
    row = np.asarray(buffer[idx], np.float64)
    return np.hstack([row[1:4], row[4:7]]), idx


def terminal(task, q2, ee_pos, obj_pos, dest_goal, touch, num_episodes, base_pos, ee_quat=None, reach_goal=None, picked=None, obj_vel=None):
    """_get_terminal_inspection (env_mujoco_util.py:492-600) for picking / placing (4-tuples in the reference) and reaching / grasping /
    pickAndplace / carrying / releasing / pushing (3-tuples there, which env_mujoco.py:125 cannot unpack: the success flag is the fix).
    `num_episodes` is the counter value *before* the call (the function increments it first).  pickAndplace carries the `picked`
    flag (self.picked): pass a one-element list, updated in place."""
    n = num_episodes + 1
    wb = np.linalg.norm(np.asarray(ee_pos) - base_pos)
    if np.pi - 0.1 < q2 < np.pi + 0.1:
        return True, -1.0, wb, 0
    if task in ("carrying", "pushing"):   # :549-550, 583-584: `return True, 0, wb` (3-tuples: success flag 0 added)
        return True, 0.0, wb, 0
    if task == "releasing":   # :551-566 (3-tuple in the reference); obj_vel = get_obj_vel() = qvel[9:12] (mujoco.py:212-215)
        dd = np.linalg.norm(np.asarray(dest_goal)[:2] - np.asarray(obj_pos)[:2])
        if obj_pos[2] < 0.1:
            return True, -20.0, wb, 0
        if dd < 0.04 and touch == 0 and obj_pos[2] < 0.35 and np.linalg.norm(obj_vel) < 0.01:
            return True, 200 - n * 0.1, wb, 1
        if dd > 0.04 and touch == 0 and obj_pos[2] < 0.20:
            return True, -20.0, wb, 0
        return False, 0.0, wb, 0
    if task == "grasping":   # :521-536
        if np.linalg.norm(np.asarray(ee_pos) - np.asarray(obj_pos)) > 0.2:
            return True, -20.0, wb, 0
        if obj_pos[2] > OBJECT_Z + 0.07 and touch in (1, 3):
            return True, 200 - n * 0.1, wb, 1
        if obj_pos[2] < 0.1:
            return True, -20.0, wb, 0
        return False, 0.0, wb, 0
    if task == "pickAndplace":   # :585-600
        dd = np.linalg.norm(np.asarray(dest_goal)[:2] - np.asarray(obj_pos)[:2])
        if obj_pos[2] > OBJECT_Z + 0.07 and touch in (1, 3) and not picked[0]:
            picked[0] = True
            return False, 20.0, wb, 0
        if dd < 0.04 and touch == 0 and obj_pos[2] < 0.35:
            return True, 180.0, wb, 1
        if obj_pos[2] < 0.1:
            return True, -20.0, wb, 0
        return False, 0.0, wb, 0
    if task == "reaching":   # :504-520 (a 3-tuple in the reference: success flag added)
        if np.linalg.norm(np.asarray(ee_pos) - reach_goal[:3]) < 0.025 and reach_ang_diff(ee_quat, reach_goal[3:6]) < np.pi / 6:
            return True, 200 - n * 0.1, wb, 1
        return False, 0.0, wb, 0
    if task == "picking":
        if obj_pos[2] > OBJECT_Z + 0.07 and touch in (1, 3):
            return True, 200 - n * 0.1, wb, 1
        if obj_pos[2] < 0.1:
            return True, -20.0, wb, 0
        return False, 0.0, wb, 0
    dest_diff = np.linalg.norm(np.asarray(dest_goal)[:2] - np.asarray(obj_pos)[:2])
    if obj_pos[2] < 0.1:
        return True, -20.0, wb, 0
    if dest_diff < 0.02 and touch == 0 and obj_pos[2] < 0.35:
        return True, 200 - n * 0.1, wb, 1
    if dest_diff > 0.02 and touch == 0 and obj_pos[2] < 0.20:
        return True, -20.0, wb, 0
    return False, 0.0, wb, 0


def env_terminal(task, current_steps, *args, **kw):
    """JacoMujocoEnv.terminal_inspection (env_mujoco.py:144-150): current_steps is the value *before* the call."""
    task_max = 700 if task in ("picking", "placing") else (1200 if task == "pickAndplace" else 500)   # env_mujoco.py:18-23
    if current_steps + 1 < task_max:
        return terminal(task, *args, **kw)
    return True, -10.0, 0.0, 0


def rulebased_subgoal(task, ee_pos, obj_goal, obj_y, dest_goal, noise6):
    """_get_rulebased_subgoal (env_mujoco_util.py:273-300); noise6 = the six np.random.uniform() draws."""
    D = np.asarray(ee_pos, dtype=np.float64) - obj_goal
    pos = D / np.linalg.norm(D) * 0.12 + obj_goal + (np.asarray(noise6[:3]) - 0.5) / 25
    if pos[2] < OBJECT_Z + 0.1:
        pos[2] = OBJECT_Z + 0.1
    if pos[1] > obj_y - 0.15:
        pos[1] = obj_y - 0.15
    x, y, z = -D / np.linalg.norm(D)
    s = np.sqrt(2) / 2
    rv = euler_from_quat(np.array([s, s * x, s * y, s * z]))
    rd = euler_from_quat(np.array([np.sqrt(3) / 2, 0.5 * x, 0.5 * y, 0.5 * z]))
    ori = np.cross(rv, rd)
    ori[0] -= np.pi / 2
    ori = ori + (np.asarray(noise6[3:6]) - 0.5) / 10
    if task == "placing":
        return np.array(dest_goal, dtype=np.float64), np.array([0, np.pi / 2, 0])
    return pos, ori


def observation(task, touch, ee_pos, ee_quat, grip, obj_pos, dest_goal, obj_goal, noise6, reach_goal=None):
    """_get_observation -> float32[26]: rule-based-subgoal branch (env_mujoco_util.py:240-254,271), or -- reach_goal given --
    the rulebased_subgoal = False branch with the reaching goal in [17:23] (:255-270)."""
    if reach_goal is not None:
        pos, ori = np.asarray(reach_goal[:3], np.float64), np.asarray(reach_goal[3:6], np.float64)
    else:
        pos, ori = rulebased_subgoal(task, ee_pos, obj_goal, obj_pos[1], dest_goal, noise6)
    o = np.hstack([[touch], ee_pos, euler_from_quat(ee_quat) / np.pi, [(grip - 0.8) / 0.2], obj_pos, [0, 0, 0], dest_goal, pos, ori / np.pi,
                   [0, np.pi / 2, 0]])
    return o.astype(np.float32)


def take_action(ee_pos, ee_quat, a, grip_prev, skip_frames=50):
    """_take_action (env_mujoco_util.py:602-646): EE target, gripper increment + ramp."""
    pose = np.concatenate([ee_pos, euler_from_quat(ee_quat)])
    a = np.asarray(a, dtype=np.float64)
    target = pose + np.hstack([a[:3] / 25, a[3:6] / 5])
    if abs(target[5]) > np.pi:
        target[5] += -np.sign(target[5]) * 2 * np.pi
    if len(a) == 6:
        return target, 0.6, np.full(skip_frames, 0.6)
    grip = float(np.clip(grip_prev + a[6] / 10, 0.6, 1.0))
    return target, grip, np.linspace(grip_prev, grip, skip_frames)


def osc_generate(q_dq, target, J, M, bias, ee_pos, ee_quat, kp=50.0, ko=180.0, kv=20.0, vmax=(0.4, 1.0472)):
    """abr_control OSC.generate as used at env_mujoco_util.py:59-63,85-90 [EXT, App. D.2; unpinned].
    J: 6x6 (pos rows then rot rows) for the arm dofs, M: 6x6 arm block, bias = qfrc_bias[arm]; all one substep stale."""
    dq = np.asarray(q_dq, dtype=np.float64)
    Minv = np.linalg.inv(M)
    Mx_inv = J @ Minv @ J.T
    if abs(np.linalg.det(Mx_inv)) >= 1e-3:
        Mx = np.linalg.inv(Mx_inv)
    else:
        u_, s_, vh = np.linalg.svd(Mx_inv)
        s_inv = np.array([0.0 if x < 0.005 else 1.0 / x for x in s_])
        Mx = vh.T @ np.diag(s_inv) @ u_.T
    u_task = np.zeros(6)
    u_task[:3] = np.asarray(ee_pos) - target[:3]
    qd = quat_from_euler(*target[3:6]); qd /= np.linalg.norm(qd)
    qe = np.asarray(ee_quat, dtype=np.float64)
    qc = np.array([qe[0], -qe[1], -qe[2], -qe[3]])
    w = qd[0] * qc[0] - qd[1:] @ qc[1:]
    v = qd[0] * qc[1:] + qc[0] * qd[1:] + np.cross(qd[1:], qc[1:])
    u_task[3:] = -v * np.sign(w)
    lamb = np.array([kp / kv] * 3 + [ko / kv] * 3)
    scale = np.ones(6)
    sat_xyz, sat_abg = vmax[0] / kp * kv, vmax[1] / ko * kv
    nx, na = np.linalg.norm(u_task[:3]), np.linalg.norm(u_task[3:])
    if nx > sat_xyz:
        scale[:3] *= sat_xyz / nx
    if na > sat_abg:
        scale[3:] *= sat_abg / na
    u_task = kv * scale * lamb * u_task
    u = -kv * (M @ dq) - J.T @ (Mx @ u_task)
    return u + bias      # "u -= g" with g = -qfrc_bias (mujoco_config.py:216)
