// Env-level task logic executed inside the physics kernels.
// Included by physics_kernel.h.  Everything here mirrors /root/reference/env_script/env_mujoco_util.py and
// env_mujoco.py; the operational-space controller mirrors abr_control's OSC.generate [EXT, SURVEY App. D.2].
//
// Staleness: the reference's controller reads qpos/qvel of the current substep but J, M, qfrc_bias and the
// EE pose from mjData, i.e. from the forward pass of the *previous* substep; likewise the observation reads body poses
// and sensordata computed at the start of the last substep.  In this kernel those quantities are simply still in LDS
// from the previous substep when the controller runs (and are carried across launches in the per-env cache row).
#pragma once

// per-env task row (floats)
#define JT_GRIP 0        // gripper command angle (env_mujoco_util.py:96,629-630)
#define JT_STEPS 1       // JacoMujocoEnv.current_steps
#define JT_EPISODES 2    // JacoMujocoEnvUtil.num_episodes
#define JT_DONE 3        // 1 after a terminal step until the env is reset (frozen)
#define JT_OBJGOAL 4     // [3]
#define JT_DESTGOAL 7    // [3]
#define JT_TARGET 10     // [6] EE target pose (xyz + euler rxyz)
#define JT_GRIP_PREV 16
#define JT_SUB 17        // substep index inside the current env step (tier hand-off)
#define JT_RNG 18        // draw counter: the 29-bit count with bit 30 set (rng_count / rng_slot below), i.e. always the bit pattern of a NORMAL
                         // float (exponent field 128..191) -- the library is built with -fgpu-flush-denormals-to-zero, and a bare small unsigned
                         // would be a denormal pattern that any canonicalising operation zeroes.  An all-zero row reads as count 0.
#define JT_PENDING 19    // 1: JT_CTRL holds the ctrl of the interrupted substep
#define JT_CTRL 20       // [9]
#define JT_SUCC 29       // success flag of the last terminal step
#define JT_WB 30         // last |EE - base| (get_wb)
#define JT_PICKED 31     // task pickAndplace: self.picked (env_mujoco_util.py:587-590)
#define JT_REACHGOAL 32  // [6] reaching goal: position + Euler rxyz (drawn at reset, env_mujoco_util.py:199-207)
#define JT_FWD 39        // auto-reset: 1 while the reset env's forward pass (sim.forward() + observation) is still to be done (survives a tier hand-off)
#define JT_PHASE 38      // task grasping, reset only: 0 pre-reach not started, 1 in its first loop (:138-159), 2 in its second (:160-170), 3 done
#define JTASK_N 40
static_assert(JTASK_N == JTASK_FLOATS, "task row size");
// per-env cache row: what the controller reads one substep late
#define JC_M 0           // [6][6] arm block of the mass matrix
#define JC_BIAS 36       // [6]
#define JC_CDOF 42       // [6][6] motion subspaces of the arm dofs
#define JC_EEPOS 78      // [3] frame of the body carrying the EE
#define JC_EEMAT 81      // [9]
#define JC_OBJPOS 90     // [3] object body position
#define JC_PIN 96        // [7] placing reset: pose (xyz, quat) the object is pinned to while the fingers close
#define JCACHE_N 104

#define JFLAG_OSC_SINGULAR 64u
#define JFLAG_PREREACH_CAP 0x20000u   // the grasping reset's pre-reach loops (unbounded `while True` in the reference) hit the substep cap

// task ids (include/jaco_env.h JACO_TASK_*)
#define JTASK_PICKING 0
#define JTASK_PLACING 1
#define JTASK_REACHING 2
#define JTASK_GRASPING 3
#define JTASK_PICKANDPLACE 4
#define JTASK_CARRYING 5     // object in hand after the hold AND the pre-reach loops (:106-117,123-170); every episode ends in its first step (:549-550)
#define JTASK_RELEASING 6    // own init pose with the fingers at 0.6 (:186-189), object in hand (:106-117), termination :551-566 (reads the object's velocity)
#define JTASK_PUSHING 7      // 6-wide action (env_mujoco.py:79-82); every episode ends in its first step (:583-584)

#define JOSC_KP 50.f
#define JOSC_KO 180.f
#define JOSC_KV 20.f
#define JOSC_VMAX_XYZ 0.4f
#define JOSC_VMAX_ABG 1.0472f

// ---------------------------------------------------------------- small rotation helpers
JDEV void mat_to_euler_rxyz(const m3& M, float* e) {   // Gohlke euler_from_matrix(..., 'rxyz'): R = Rx(a) Ry(b) Rz(c)
  float cy = sqrtf(M.m[8] * M.m[8] + M.m[5] * M.m[5]);
  if (cy > 8.8e-16f) { e[0] = atan2f(-M.m[5], M.m[8]); e[1] = atan2f(M.m[2], cy); e[2] = atan2f(-M.m[1], M.m[0]); }
  else { e[0] = 0.f; e[1] = atan2f(M.m[2], cy); e[2] = atan2f(M.m[3], M.m[4]); }
}
JDEV void euler_rxyz_to_quat(float a, float b, float c, float* q) {
  float ca = cosf(0.5f * a), sa = sinf(0.5f * a), cb = cosf(0.5f * b), sb = sinf(0.5f * b), cc = cosf(0.5f * c), sc = sinf(0.5f * c);
  // qx(a) * qy(b) * qz(c)
  float w1 = ca * cb, x1 = sa * cb, y1 = ca * sb, z1 = sa * sb;
  q[0] = w1 * cc - z1 * sc; q[1] = x1 * cc + y1 * sc; q[2] = y1 * cc - x1 * sc; q[3] = w1 * sc + z1 * cc;
}
JDEV m3 euler_rxyz_to_mat(float a, float b, float c) {   // Rx(a) Ry(b) Rz(c)
  float ca = cosf(a), sa = sinf(a), cb = cosf(b), sb = sinf(b), cc = cosf(c), sc = sinf(c);
  m3 M;
  M.m[0] = cb * cc; M.m[1] = -cb * sc; M.m[2] = sb;
  M.m[3] = sa * sb * cc + ca * sc; M.m[4] = -sa * sb * sc + ca * cc; M.m[5] = -sa * cb;
  M.m[6] = -ca * sb * cc + sa * sc; M.m[7] = ca * sb * sc + sa * cc; M.m[8] = ca * cb;
  return M;
}
JDEV void mat_to_quat(const m3& R, float* q) {
  float t = R.m[0] + R.m[4] + R.m[8];
  if (t > 0.f) {
    float s = sqrtf(t + 1.f) * 2.f;
    q[0] = 0.25f * s; q[1] = (R.m[7] - R.m[5]) / s; q[2] = (R.m[2] - R.m[6]) / s; q[3] = (R.m[3] - R.m[1]) / s;
  } else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) {
    float s = sqrtf(1.f + R.m[0] - R.m[4] - R.m[8]) * 2.f;
    q[0] = (R.m[7] - R.m[5]) / s; q[1] = 0.25f * s; q[2] = (R.m[1] + R.m[3]) / s; q[3] = (R.m[2] + R.m[6]) / s;
  } else if (R.m[4] > R.m[8]) {
    float s = sqrtf(1.f + R.m[4] - R.m[0] - R.m[8]) * 2.f;
    q[0] = (R.m[2] - R.m[6]) / s; q[1] = (R.m[1] + R.m[3]) / s; q[2] = 0.25f * s; q[3] = (R.m[5] + R.m[7]) / s;
  } else {
    float s = sqrtf(1.f + R.m[8] - R.m[0] - R.m[4]) * 2.f;
    q[0] = (R.m[3] - R.m[1]) / s; q[1] = (R.m[2] + R.m[6]) / s; q[2] = (R.m[5] + R.m[7]) / s; q[3] = 0.25f * s;
  }
}
JDEV void quat_to_euler_rxyz(const float* q, float* e) {
  float n = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  m3 M = quat2mat(q[0] * n, q[1] * n, q[2] * n, q[3] * n);
  mat_to_euler_rxyz(M, e);
}

// counter-based uniform [0,1) (the reference draws from the global, unseeded numpy RNG: not reproducible by design)
JDEV float rng_uniform(unsigned long long seed, unsigned env, unsigned counter) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)env * 0x100000001ull + counter + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// EE frame from the (stale) frame of the body that carries it
template <class L>
JDEV void ee_frame(const JacoModelDev* m, const L& s, v3* pos, m3* R) {
  int eb = m->ee_body;
  m3 Rb = ldm(s.xmat[eb]);
  *pos = ld3(s.xpos[eb]) + mul(Rb, ld3(m->ee_pos));
  *R = mul(Rb, ldm(m->ee_mat));
}

// EE_obj frame (the grasp frame the placing reset pins the object to, env_mujoco_util.py:107)
template <class L>
JDEV void eeobj_frame(const JacoModelDev* m, const L& s, v3* pos, m3* R) {
  int eb = m->eeobj_body;
  m3 Rb = ldm(s.xmat[eb]);
  *pos = ld3(s.xpos[eb]) + mul(Rb, ld3(m->eeobj_pos));
  *R = mul(Rb, ldm(m->eeobj_mat));
}

// ---------------------------------------------------------------- a8: rule-based subgoal (env_mujoco_util.py:273-300)
JDEV void rulebased_subgoal(int task, v3 ee, v3 obj_goal, float obj_y, v3 dest_goal, const float* nz, float* pos, float* ori) {
  v3 D = ee - obj_goal;
  float dn = norm(D);
  v3 p = D * (0.12f / dn) + obj_goal + mk3((nz[0] - 0.5f) / 25.f, (nz[1] - 0.5f) / 25.f, (nz[2] - 0.5f) / 25.f);
  if (p.z < 0.1898f + 0.1f) p.z = 0.1898f + 0.1f;
  if (p.y > obj_y - 0.15f) p.y = obj_y - 0.15f;
  v3 x = D * (-1.f / dn);
  const float s2 = 0.70710678118654752f, s3 = 0.86602540378443865f;
  float qv[4] = {s2, s2 * x.x, s2 * x.y, s2 * x.z}, qd[4] = {s3, 0.5f * x.x, 0.5f * x.y, 0.5f * x.z}, rv[3], rd[3];
  quat_to_euler_rxyz(qv, rv);
  quat_to_euler_rxyz(qd, rd);
  v3 c = cross(mk3(rv[0], rv[1], rv[2]), mk3(rd[0], rd[1], rd[2]));
  ori[0] = c.x - 1.57079632679489662f + (nz[3] - 0.5f) / 10.f;
  ori[1] = c.y + (nz[4] - 0.5f) / 10.f;
  ori[2] = c.z + (nz[5] - 0.5f) / 10.f;
  pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
  if (task == 1) {  // placing (:297-298)
    pos[0] = dest_goal.x; pos[1] = dest_goal.y; pos[2] = dest_goal.z;
    ori[0] = 0.f; ori[1] = 1.57079632679489662f; ori[2] = 0.f;
  }
}

// ---------------------------------------------------------------- a9: touch class (env_mujoco_util.py:470-490); sens in XML order
JDEV int touch_class(float sens, int lane) {
  // touch_array[i] = "i_touch" = sensordata[1 + i] (i = 0..18), touch_array[19] = EE_touch = sensordata[0]
  bool hit = sens > 0.001f;
  unsigned long long m = wave_ballot(hit && lane < JNSENS);
  unsigned long long arr = ((m >> 1) & 0x7FFFFull) | ((m & 1ull) << 19);
  bool thumb = (arr & 0x1Eull) != 0, index = (arr & 0x1E0ull) != 0, pinky = (arr & 0x1E00ull) != 0;
  if ((thumb && index) || (thumb && pinky)) return 3;
  if (arr & 0x1FFFull) return 1;
  if (arr & 0xFE000ull) return 2;
  return 0;
}

// ---------------------------------------------------------------- a10: picking reward (env_mujoco_util.py:392-431)
// task 'grasping' (env_mujoco_util.py:352-391) has the same shape with touch terms 0.75 / -0.75 / 2.5 and scale 0.05 (picking: 0.01)
JDEV float reward_picking(v3 ee, const float* eul, v3 obj, int touch, float scale = 0.01f) {
  float cr = cosf(eul[0]), sr = sinf(eul[0]), cp = cosf(eul[1]), sp = sinf(eul[1]);
  // Rx(r) Ry(p) Rz(y) applied to (0,0,-1): third column negated
  v3 ee_vec = mk3(-sp, sr * cp, -cr * cp);
  v3 d = obj - ee;
  float dn = norm(d);
  v3 u = d * (1.f / dn);
  // (arguments clamped to [-1, 1]: in fp32 the EE straight above / beside the object puts them an ulp outside in ~3e-4 of
  // all steps, which numpy's arccos would turn into a NaN reward; the reference's fp64 hits that window ~1e-8 of the time)
  float pitch = -acosf(fminf(1.f, fmaxf(-1.f, u.z)));
  float yaw = -acosf(fminf(1.f, fmaxf(-1.f, -u.x / sqrtf(fmaxf(1e-30f, 1.f - u.z * u.z)))));
  // rot_ee^T (0,0,1) with roll = 0: third row of Ry(pitch) Rz(yaw)
  float cpi = cosf(pitch), spi = sinf(pitch), cya = cosf(yaw), sya = sinf(yaw);
  v3 tv = mk3(-spi * cya, spi * sya, -cpi);   // (z already negated: target_vec[2] *= -1)
  float ang = norm(ee_vec - tv);
  float r = 2.5f * expf(-dn / 0.2f) + expf(-ang / 0.52359877559829887f) / (dn * 15.f + 1.f);
  r += touch == 1 ? 0.75f : (touch == 2 ? -0.75f : (touch == 3 ? 2.5f : 0.f));
  r += 100.f * (obj.z - 0.1898f);
  return r * scale;
}

// ---------------------------------------------------------------- a10 / a11 for task 'reaching' (env_mujoco_util.py:314-351, 504-520)
// The reference differences Euler angles after an euler -> unit quaternion -> euler round trip (canonical angles, pitch in
// [-pi/2, pi/2]): the goal's (alpha, beta, gamma) has beta = acos(.) in [0, pi] and is NOT canonical as drawn.
JDEV float reach_ang_diff(const float* eul_ee, const float* eul_goal) {
  float q[4], a[3], b[3];
  euler_rxyz_to_quat(eul_ee[0], eul_ee[1], eul_ee[2], q); quat_to_euler_rxyz(q, a);
  euler_rxyz_to_quat(eul_goal[0], eul_goal[1], eul_goal[2], q); quat_to_euler_rxyz(q, b);
  float d = sqrtf((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
  const float PI = 3.14159265358979323846f;
  return d > PI ? 2.f * PI - d : d;
}
JDEV float reward_reaching(v3 ee, const float* eul, const float* goal, v3 base) {
  const float dist = norm(ee - mk3(goal[0], goal[1], goal[2])), ang = reach_ang_diff(eul, goal + 3);
  float r = 5.f * expf(-dist) * 0.5f + 2.f * expf(-ang / 0.52359877559829887f) / (2.f * (dist * 15.f + 1.f));
  const float wb = norm(ee - base);
  if (wb < 0.15f) r -= 0.15f - wb;      // gripper too close to the robot base
  if (ee.z < 0.1f) r -= 0.1f - ee.z;    // gripper too low
  return 0.05f * r;
}

// ---------------------------------------------------------------- a11: termination (env_mujoco.py:144-150, env_mujoco_util.py:492-582)
// returns done; *bonus, *succ; updates steps / episodes counters in the task row
JDEV bool terminal_inspection(int task, float* trow, float q2, v3 ee, v3 base, v3 obj, v3 dest_goal, int touch, float* bonus, int* succ, float* wb,
                              const float* eul = nullptr, const float* reachgoal = nullptr, float* picked = nullptr, float objvel = 0.f) {
  float steps = trow[JT_STEPS] + 1.f;
  trow[JT_STEPS] = steps;
  // picking / placing 700, pickAndplace 1200, reaching / grasping / carrying / releasing / pushing 500 (env_mujoco.py:18-23)
  const float task_max = (task == JTASK_PICKING || task == JTASK_PLACING) ? 700.f : (task == JTASK_PICKANDPLACE ? 1200.f : 500.f);
  *succ = 0; *wb = 0.f;
  if (!(steps < task_max)) { *bonus = -10.f; return true; }
  float n = trow[JT_EPISODES] + 1.f;
  trow[JT_EPISODES] = n;
  *wb = norm(ee - base);
  const float PI = 3.14159265358979323846f;
  if (PI - 0.1f < q2 && q2 < PI + 0.1f) { *bonus = -1.f; return true; }
  if (task == 2) {   // reaching (:504-520; the reference returns a 3-tuple here, which env_mujoco.py:125 cannot unpack: the success flag is the fix)
    const float dist = norm(ee - mk3(reachgoal[0], reachgoal[1], reachgoal[2]));
    if (dist < 0.025f && reach_ang_diff(eul, reachgoal + 3) < PI / 6.f) { *bonus = 200.f - n * 0.1f; *succ = 1; return true; }
    *bonus = 0.f;
    return false;
  }
  if (task == JTASK_GRASPING) {   // :521-536 (3-tuple in the reference: the success flag is the fix)
    if (norm(ee - obj) > 0.2f) { *bonus = -20.f; return true; }             // gripper too far away from the object
    if (obj.z > 0.1898f + 0.07f && (touch == 1 || touch == 3)) { *bonus = 200.f - n * 0.1f; *succ = 1; return true; }
    if (obj.z < 0.1f) { *bonus = -20.f; return true; }
    *bonus = 0.f;
    return false;
  }
  if (task == JTASK_PICKANDPLACE) {   // :585-600, with the `picked` flag carried in the task row
    const float dx = dest_goal.x - obj.x, dy = dest_goal.y - obj.y, dd = sqrtf(dx * dx + dy * dy);
    if (obj.z > 0.1898f + 0.07f && (touch == 1 || touch == 3) && *picked == 0.f) { *picked = 1.f; *bonus = 20.f; return false; }
    if (dd < 0.04f && touch == 0 && obj.z < 0.35f) { *bonus = 180.f; *succ = 1; return true; }
    if (obj.z < 0.1f) { *bonus = -20.f; return true; }
    *bonus = 0.f;
    return false;
  }
  if (task == JTASK_CARRYING || task == JTASK_PUSHING) { *bonus = 0.f; return true; }   // `return True, 0, wb` (:549-550,583-584; 3-tuples: success flag 0 added)
  if (task == JTASK_RELEASING) {   // :551-566 (3-tuple in the reference: the success flag is the fix); objvel = |qvel[9:12]| (mujoco.py:212-215)
    const float dx = dest_goal.x - obj.x, dy = dest_goal.y - obj.y, dd = sqrtf(dx * dx + dy * dy);
    if (obj.z < 0.1f) { *bonus = -20.f; return true; }
    if (dd < 0.04f && touch == 0 && obj.z < 0.35f && objvel < 0.01f) { *bonus = 200.f - n * 0.1f; *succ = 1; return true; }
    if (dd > 0.04f && touch == 0 && obj.z < 0.20f) { *bonus = -20.f; return true; }
    *bonus = 0.f;
    return false;
  }
  if (task == 0) {
    if (obj.z > 0.1898f + 0.07f && (touch == 1 || touch == 3)) { *bonus = 200.f - n * 0.1f; *succ = 1; return true; }
    if (obj.z < 0.1f) { *bonus = -20.f; return true; }
    *bonus = 0.f;
    return false;
  }
  float dx = dest_goal.x - obj.x, dy = dest_goal.y - obj.y, dd = sqrtf(dx * dx + dy * dy);
  if (obj.z < 0.1f) { *bonus = -20.f; return true; }
  if (dd < 0.02f && touch == 0 && obj.z < 0.35f) { *bonus = 200.f - n * 0.1f; *succ = 1; return true; }
  if (dd > 0.02f && touch == 0 && obj.z < 0.20f) { *bonus = -20.f; return true; }
  *bonus = 0.f;
  return false;
}

// ---------------------------------------------------------------- a12 (grasping): pre-reach target / loop conditions (env_mujoco_util.py:123-170)
// value of np.float16(x) as a float: round to nearest even on 11 significant bits (normal range; |x| < 65504, subnormals flushed
// like values that small never matter here: the angles are O(1))
JDEV float f16_round(float x) {
  unsigned u = __float_as_uint(x);
  const unsigned sign = u & 0x80000000u;
  u &= 0x7FFFFFFFu;
  if (u < 0x38800000u) {   // below the smallest normal half (6.1e-5): subnormal halves are multiples of 2^-24
    const float q = rintf(__uint_as_float(u) * 16777216.f) * (1.f / 16777216.f);
    return __uint_as_float(__float_as_uint(q) | sign);
  }
  u += 0x00000FFFu + ((u >> 13) & 1u);   // round to nearest even at bit 13
  u &= 0xFFFFE000u;
  return __uint_as_float(u | sign);
}
// orientation part of the pre-reach target: looks along EE -> object, yaw = the drawn gamma; float16 like the reference's array
JDEV void grasp_reach_ori(v3 ee, v3 obj_goal, float gamma, float* ori) {
  const v3 d = obj_goal - ee;
  const float sx = d.x > 0.f ? 1.f : (d.x < 0.f ? -1.f : 0.f);
  ori[0] = f16_round(-asinf(d.y / sqrtf(d.y * d.y + d.z * d.z)) * sx);
  ori[1] = f16_round(acosf(d.x / norm(d)) * sx);
  ori[2] = f16_round(gamma);
}
// |unit(quat_from_euler(EE euler)) - unit(quat_from_euler(goal euler))| (:144-150)
JDEV float grasp_ang_diff(const float* eul_ee, const float* eul_goal) {
  float a[4], b[4];
  euler_rxyz_to_quat(eul_ee[0], eul_ee[1], eul_ee[2], a);
  euler_rxyz_to_quat(eul_goal[0], eul_goal[1], eul_goal[2], b);
  const float na = rsqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]), nb = rsqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3]);
  float s = 0.f;
  for (int k = 0; k < 4; k++) { const float d = a[k] * na - b[k] * nb; s += d * d; }
  return sqrtf(s);
}

JDEV unsigned rng_count(float slot) { return __float_as_uint(slot) & 0x1FFFFFFFu; }
JDEV float rng_slot(unsigned count) { return __uint_as_float((count & 0x1FFFFFFFu) | 0x40000000u); }   // (wraps after 5e8 draws = 4e7 env steps)

// ---------------------------------------------------------------- a12: the draws of _reset (env_mujoco_util.py:92-121,176-219)
// One env's randomised initial state: arm angles (_create_init_angle), the reaching / object / destination goals (__sample_goal), object on
// the holder and pedestal at its goal (set_obj_xyz with the zero quaternion, set_dest_xyz), task row cleared (draw counter kept).
// `q` (nq, already holding qpos0) and `t` (the task row) may live in global memory (jaco_reset_kernel: one thread per env) or in LDS (the
// in-kernel auto-reset of a finished env: one lane).  Same code, same counter-based RNG stream: both paths produce the same bits.
// Optional buffer of recorded goals (kwarg init_buffer, env_mujoco_util.py:46,208-212): rows of `stride` floats, row[1:4] = goal position, row[4:7] = goal orientation.
struct GoalBuffer { const float* rows; int n, stride; };
JDEV void reset_draws(int task_id, unsigned long long seed, unsigned env, int has_free, const float* base, float* q, float* t, GoalBuffer gb = GoalBuffer{nullptr, 0, 0}) {
  unsigned c = rng_count(t[JT_RNG]);
#define JRU(lo, hi) ((lo) + ((hi) - (lo)) * rng_uniform(seed, env, c++))
  const float PI = 3.14159265358979323846f;
  if (task_id == JTASK_PLACING || task_id == JTASK_GRASPING || task_id == JTASK_CARRYING) {   // 'carrying', 'grasping', 'placing' (:181-185)
    const float pick = JRU(0.f, 1.f);
    const float a0 = pick < 0.5f ? JRU(3.f * PI / 8.f, PI / 2.f) : JRU(PI / 2.f, 5.f * PI / 8.f);
    q[0] = a0; q[1] = 3.85f; q[2] = JRU(1.f, 1.1f); q[3] = JRU(2.f, 2.1f); q[4] = JRU(0.8f, 2.3f); q[5] = JRU(-1.2f, -1.1f);
  } else if (task_id == JTASK_RELEASING) {                       // 'releasing' (:186-189): nine values, the finger joints start at 0.6
    q[0] = JRU(1.9f, 2.f); q[1] = JRU(3.3f, 3.6f); q[2] = JRU(0.5f, 0.8f); q[3] = JRU(1.8f, 2.5f); q[4] = JRU(1.3f, 2.f);
    q[5] = JRU(-0.4f, -0.9f);   // (uniform(-0.4, -0.9): numpy draws low + (high - low) u whatever the order of the bounds)
    q[6] = 0.6f; q[7] = 0.6f; q[8] = 0.6f;
  } else {                                                       // 'reaching', 'picking', 'pickAndplace' (:177-180); 'pushing': the reference has
                                                                 // no branch for it (UnboundLocalError in _create_init_angle) -- this one is used
    q[0] = JRU(0.7f, 2.5f); q[1] = JRU(3.8f, 4.f); q[2] = JRU(1.f, 1.7f); q[3] = JRU(1.8f, 2.5f); q[4] = JRU(1.f, 2.5f); q[5] = JRU(0.8f, 2.3f);
  }
  for (int k = 0; k < JTASK_N; k++) if (k != JT_RNG) t[k] = 0.f;
  t[JT_GRIP] = 0.6f; t[JT_GRIP_PREV] = 0.6f;
  if (gb.rows && gb.n >= 2) {   // goal_buffer branch (:208-212): random_idx = np.random.randint(0, len(buffer) - 1), i.e. one of rows 0 .. len - 2 (the
                               // last row is never drawn); position = row[1:4], orientation = row[4:7], taken as they are (no float16 cast here)
    int idx = (int)(rng_uniform(seed, env, c++) * (float)(gb.n - 1));
    idx = idx > gb.n - 2 ? gb.n - 2 : idx;
    const float* row = gb.rows + (size_t)idx * gb.stride;
    for (int k = 0; k < 6; k++) t[JT_REACHGOAL + k] = row[1 + k];
  } else
  {   // reaching goal (__sample_goal, :199-207; drawn for every task, read by task 'reaching' and by the reaching-goal observation)
    float g[3];
    for (int k = 0; k < 2; k++) { const float mag = JRU(0.3f, 0.42f); const float sgn = JRU(0.f, 1.f); g[k] = sgn < 0.5f ? -mag : mag; }
    g[2] = JRU(0.3f, 0.5f);
    float x = g[0] - base[0], y = g[1] - base[1], z = g[2] - base[2];
    const float n = sqrtf(x * x + y * y + z * z);
    x /= n; y /= n; z /= n;
    const float sx = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    const float alpha = -asinf(y / sqrtf(y * y + z * z)) * sx, beta = acosf(x) * sx, gamma = JRU(-0.1f, 0.1f);   // (|xyz| = 1)
    t[JT_REACHGOAL] = g[0]; t[JT_REACHGOAL + 1] = g[1]; t[JT_REACHGOAL + 2] = g[2];
    // np.array([alpha, beta, gamma], dtype=np.float16) (:206): the orientation is stored with 11 bits of mantissa
    t[JT_REACHGOAL + 3] = f16_round(alpha); t[JT_REACHGOAL + 4] = f16_round(beta); t[JT_REACHGOAL + 5] = f16_round(gamma);
  }
  if (has_free) {   // __sample_goal (:215-219), set_dest_xyz (mujoco.py:229-237), set_obj_xyz with the zero quaternion (:119-121)
    const float ox = JRU(-0.1f, 0.1f), oy = 0.65f + JRU(-0.08f, 0.02f), dx = 0.4f + JRU(-0.05f, 0.05f), dy = 0.3f + JRU(-0.05f, 0.05f);
    q[9] = ox; q[10] = oy; q[11] = 0.1898f; q[12] = 1.f; q[13] = 0.f; q[14] = 0.f; q[15] = 0.f;
    q[16] = dx; q[17] = dy;
    t[JT_OBJGOAL] = ox; t[JT_OBJGOAL + 1] = oy; t[JT_OBJGOAL + 2] = 0.1898f;
    t[JT_DESTGOAL] = dx; t[JT_DESTGOAL + 1] = dy; t[JT_DESTGOAL + 2] = 0.3468f;
  }
#undef JRU
  t[JT_RNG] = rng_slot(c);
}

// ---------------------------------------------------------------- a4/a5: operational-space controller on the wave
// 6x6 Gauss-Jordan, one matrix row per lane (lanes 0..5), no pivoting (SPD input). B in: identity row; out: inverse row.
JDEV float gj_inverse6(float (&A)[6], float (&B)[6], int lane) {
  float det = 1.f;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    float piv = wave_bcast(A[k], k);
    det *= piv;
    float inv = 1.f / piv, sc = lane == k ? inv : 1.f;
#pragma unroll
    for (int j = 0; j < 6; j++) { A[j] *= sc; B[j] *= sc; }
    float f = lane == k ? 0.f : A[k];
    float pa[6], pb[6];   // (broadcasts first, then the updates: physics_kernel.h ldl_solve)
#pragma unroll
    for (int j = 0; j < 6; j++) { pa[j] = wave_bcast(A[j], k); pb[j] = wave_bcast(B[j], k); }
#pragma unroll
    for (int j = 0; j < 6; j++) { A[j] -= f * pa[j]; B[j] -= f * pb[j]; }
  }
  return det;
}

// Same elimination with a single right-hand side: returns det, b becomes (A^-1 b)[lane].
JDEV float gj_solve6(float (&A)[6], float& b, int lane) {
  float det = 1.f;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    float piv = wave_bcast(A[k], k);
    det *= piv;
    float inv = 1.f / piv, sc = lane == k ? inv : 1.f;
#pragma unroll
    for (int j = 0; j < 6; j++) A[j] *= sc;
    b *= sc;
    float f = lane == k ? 0.f : A[k];
    float pa[6];
#pragma unroll
    for (int j = 0; j < 6; j++) pa[j] = wave_bcast(A[j], k);
    const float pbk = wave_bcast(b, k);
#pragma unroll
    for (int j = 0; j < 6; j++) A[j] -= f * pa[j];
    b -= f * pbk;
  }
  return det;
}

// Pseudo-inverse of a symmetric PSD 6x6 matrix X (LDS, destroyed) with abr_control's rule "singular values < 0.005 are
// dropped" (np.linalg.svd branch of OSC.generate): Jacobi eigen-decomposition, V accumulated in LDS, result in `out`.
// Parallel (round-robin) ordering: each of the 5 rounds of a sweep applies 3 rotations on disjoint index pairs at once,
// lane = (matrix row or column, pair).  Rare path (|det| < 1e-3, a few % of env steps under random actions); all control
// flow is wave-uniform.
JDEV void pinv6_jacobi(float* X, float* V, float* out, int lane) {
  // round r, pair j: p = nibble 2j, q = nibble 2j+1 (circle method for 6 players)
  const unsigned rounds[5] = {0x324150u, 0x213540u, 0x152430u, 0x541320u, 0x435210u};
  if (lane < 36) V[lane] = (lane / 6 == lane % 6) ? 1.f : 0.f;
  wave_sync();
  const int k = lane % 6, jp = lane / 6;   // lanes 0..17: row/column k, pair jp
  for (int sweep = 0; sweep < 6; sweep++) {
#pragma unroll
    for (int r = 0; r < 5; r++) {
      const unsigned code = rounds[r] >> (8 * (jp < 3 ? jp : 0));
      const int p = code & 15, q = (code >> 4) & 15;
      float app = X[p * 6 + p], aqq = X[q * 6 + q], apq = X[p * 6 + q];
      float c = 1.f, sn = 0.f;
      if (fabsf(apq) > 1e-12f * (fabsf(app) + fabsf(aqq)) && apq != 0.f) {
        float theta = (aqq - app) / (2.f * apq);
        float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
        c = 1.f / sqrtf(t * t + 1.f);
        sn = t * c;
      }
      wave_sync();   // every lane has read its pivot entries
      if (lane < 18) {   // X <- X G, V <- V G   (G: the three Givens rotations of this round)
        float akp = X[k * 6 + p], akq = X[k * 6 + q], vkp = V[k * 6 + p], vkq = V[k * 6 + q];
        X[k * 6 + p] = c * akp - sn * akq; X[k * 6 + q] = sn * akp + c * akq;
        V[k * 6 + p] = c * vkp - sn * vkq; V[k * 6 + q] = sn * vkp + c * vkq;
      }
      wave_sync();
      if (lane < 18) {   // X <- G^T X
        float apk = X[p * 6 + k], aqk = X[q * 6 + k];
        X[p * 6 + k] = c * apk - sn * aqk; X[q * 6 + k] = sn * apk + c * aqk;
      }
      wave_sync();
    }
  }
  if (lane < 36) {
    int r = lane / 6, c = lane - 6 * r;
    float a = 0.f;
    for (int kk = 0; kk < 6; kk++) {
      float ev = fabsf(X[kk * 6 + kk]);
      a += V[r * 6 + kk] * (ev < 0.005f ? 0.f : 1.f / ev) * (X[kk * 6 + kk] < 0.f ? -1.f : 1.f) * V[c * 6 + kk];
    }
    out[lane] = a;
  }
  wave_sync();
}

// Target orientation of the env step as a unit quaternion (abr_control: transformations.quaternion_from_euler(..., 'rxyz')).
// The target (task row) only changes in take_action, so this runs once per env step, not once per substep.
template <class L>
JDEV void osc_target_quat(L& s, int lane) {
  const float* tg = s.task + JT_TARGET;
  float qd[4];
  euler_rxyz_to_quat(tg[3], tg[4], tg[5], qd);
  float qn = rsqrtf(qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2] + qd[3] * qd[3]);
  if (lane < 4) s.osc_qd[lane] = (lane == 0 ? qd[0] : lane == 1 ? qd[1] : lane == 2 ? qd[2] : qd[3]) * qn;
}

// Writes the six arm torques into s.ctrl[0..5].  Scratch: the (not yet built) constraint-row area s.J.
// General form, as abr_control writes it: Mx = (J M^-1 J^T)^-1 (or its thresholded pseudo-inverse), u = -kv M dq - J^T Mx u_task + bias.
template <class L>
JDEV void stage_osc_general(const JacoModelDev* m, L& s, int lane, unsigned& flags) {
  float* Jm = s.J;            // [6][6] J[r][c], rows: 3 translational, 3 rotational; columns: arm dofs
  float* T = s.J + 72;        // M^-1 J^T
  float* X = s.J + 108;       // J M^-1 J^T, then its (pseudo-)inverse Mx
  float* w = s.J + 144;       // Mx u_task
  v3 pe; m3 Re;
  ee_frame(m, s, &pe, &Re);
  const int r = lane / 6, c = lane - 6 * r;   // lanes 0..35 = matrix entry (r, c)
  if (lane < 6) {
    sv S = ldsv(s.cdof[lane]);
    v3 jp = S.b + cross(S.a, pe);
    Jm[0 * 6 + lane] = jp.x; Jm[1 * 6 + lane] = jp.y; Jm[2 * 6 + lane] = jp.z;
    Jm[3 * 6 + lane] = S.a.x; Jm[4 * 6 + lane] = S.a.y; Jm[5 * 6 + lane] = S.a.z;
  }
  // T = M^-1 J^T by elimination on [M | J^T] (lanes 0..5 own rows; row i of J^T is this lane's own Jacobian column)
  float A[6], B[6];
  int i = lane < 6 ? lane : 0;
  {
    sv S = ldsv(s.cdof[i]);
    v3 jp = S.b + cross(S.a, pe);
    B[0] = jp.x; B[1] = jp.y; B[2] = jp.z; B[3] = S.a.x; B[4] = S.a.y; B[5] = S.a.z;
  }
#pragma unroll
  for (int j = 0; j < 6; j++) A[j] = s.M[m_index(i, j)];
  gj_inverse6(A, B, lane);
  if (lane < 6) for (int j = 0; j < 6; j++) T[lane * 6 + j] = B[j];
  wave_sync();
  if (lane < 36) { float a = 0.f; for (int k = 0; k < 6; k++) a += Jm[r * 6 + k] * T[k * 6 + c]; X[lane] = a; }
  wave_sync();
  // task-space error (uniform across lanes)
  const float* tg = s.task + JT_TARGET;
  float ut[6];
  ut[0] = pe.x - tg[0]; ut[1] = pe.y - tg[1]; ut[2] = pe.z - tg[2];
  float qd[4] = {s.osc_qd[0], s.osc_qd[1], s.osc_qd[2], s.osc_qd[3]}, qe[4];   // osc_target_quat(), once per env step
  mat_to_quat(Re, qe);
  // q_e = q_d * conj(q_EE)
  float cw = qe[0], cx = -qe[1], cy = -qe[2], cz = -qe[3];
  float ew = qd[0] * cw - qd[1] * cx - qd[2] * cy - qd[3] * cz;
  float ex = qd[0] * cx + qd[1] * cw + qd[2] * cz - qd[3] * cy;
  float ey = qd[0] * cy - qd[1] * cz + qd[2] * cw + qd[3] * cx;
  float ez = qd[0] * cz + qd[1] * cy - qd[2] * cx + qd[3] * cw;
  float sg = ew > 0.f ? 1.f : (ew < 0.f ? -1.f : 0.f);
  ut[3] = -ex * sg; ut[4] = -ey * sg; ut[5] = -ez * sg;
  // velocity limiting (vmax = [0.4, 1.0472]) then gains: u_task <- kv * scale * lambda * u_task
  float nx = sqrtf(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]), na = sqrtf(ut[3] * ut[3] + ut[4] * ut[4] + ut[5] * ut[5]);
  const float sat_xyz = JOSC_VMAX_XYZ / JOSC_KP * JOSC_KV, sat_abg = JOSC_VMAX_ABG / JOSC_KO * JOSC_KV;
  float sx = nx > sat_xyz ? sat_xyz / nx : 1.f, sa = na > sat_abg ? sat_abg / na : 1.f;
  for (int k = 0; k < 3; k++) { ut[k] *= JOSC_KP * sx; ut[3 + k] *= JOSC_KO * sa; }
  // w = Mx u_task:  Mx = X^-1 as a solve when |det X| >= 1e-3 (abr_control's plain inverse), else the SVD pseudo-inverse
  // that drops singular values < 0.005
#pragma unroll
  for (int j = 0; j < 6; j++) A[j] = X[i * 6 + j];
  float wi = ut[0];
#pragma unroll
  for (int j = 1; j < 6; j++) wi = i == j ? ut[j] : wi;
  float det = gj_solve6(A, wi, lane);
  if (fabsf(det) < 1e-3f) {
    flags |= JFLAG_OSC_SINGULAR;   // informational
    pinv6_jacobi(X, s.J + 150, s.J + 186, lane);
    wi = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) wi += s.J[186 + i * 6 + k] * ut[k];
  }
  if (lane < 6) w[lane] = wi;
  wave_sync();
  if (lane < 6) {
    float u = s.bias[lane];
    for (int k = 0; k < 6; k++) u -= JOSC_KV * s.M[m_index(lane, k)] * s.qvel[k] + Jm[k * 6 + lane] * w[k];
    s.ctrl[lane] = u;
  }
  wave_sync();
}

// Direct form for the regular case (build option -DJACO_OSC_DIRECT; NOT the default: measured on MI355X it is 0.1-0.7 % slower end to end --
// stage profile 6.5 k instead of 5.8 k cycles per substep: the pivot search adds dependent cross-lane round trips, and the kernel is
// bound by those, not by the VALU work it saves -- profiles/r03_ab_osc.txt).  J is square (6 task dimensions, 6 arm dofs), so wherever J M^-1 J^T is invertible
//     J^T (J M^-1 J^T)^-1 = J^T J^-T M J^-1 = M J^-1        and        det(J M^-1 J^T) = det(J)^2 / det(M):
// one 6x6 solve J x = u_task (partial pivoting: J is not definite) and one pivot product of M replace the elimination on [M | J^T], the
// 6x6x6 product and the second elimination of the general form -- and the solve is better conditioned (cond(J) instead of
// cond(J)^2 cond(M)).  The branch rule stays abr_control's: |det(J M^-1 J^T)| < 1e-3 -> the general form with its pseudo-inverse.
#ifdef JACO_OSC_DIRECT
template <class L>
JDEV void stage_osc(const JacoModelDev* m, L& s, int lane, unsigned& flags) {
  float* Jm = s.J;            // [6][6] J[r][c], rows: 3 translational, 3 rotational; columns: arm dofs (same layout as the general form)
  v3 pe; m3 Re;
  ee_frame(m, s, &pe, &Re);
  if (lane < 6) {
    sv S = ldsv(s.cdof[lane]);
    v3 jp = S.b + cross(S.a, pe);
    Jm[0 * 6 + lane] = jp.x; Jm[1 * 6 + lane] = jp.y; Jm[2 * 6 + lane] = jp.z;
    Jm[3 * 6 + lane] = S.a.x; Jm[4 * 6 + lane] = S.a.y; Jm[5 * 6 + lane] = S.a.z;
  }
  const int i = lane < 6 ? lane : 0;
  // det(M): pivot product of the (SPD) arm block, no pivoting
  float Am[6], detM = 1.f;
#pragma unroll
  for (int j = 0; j < 6; j++) Am[j] = s.M[m_index(i, j)];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float piv = wave_bcast(Am[k], k);
    detM *= piv;
    const float f = lane > k ? Am[k] / piv : 0.f;
#pragma unroll
    for (int j = k + 1; j < 6; j++) Am[j] -= f * wave_bcast(Am[j], k);
  }
  // task-space error and its gains (uniform across lanes), as in the general form
  const float* tg = s.task + JT_TARGET;
  float ut[6];
  ut[0] = pe.x - tg[0]; ut[1] = pe.y - tg[1]; ut[2] = pe.z - tg[2];
  float qd[4] = {s.osc_qd[0], s.osc_qd[1], s.osc_qd[2], s.osc_qd[3]}, qe[4];
  mat_to_quat(Re, qe);
  float cw = qe[0], cx = -qe[1], cy = -qe[2], cz = -qe[3];
  float ew = qd[0] * cw - qd[1] * cx - qd[2] * cy - qd[3] * cz;
  float ex = qd[0] * cx + qd[1] * cw + qd[2] * cz - qd[3] * cy;
  float ey = qd[0] * cy - qd[1] * cz + qd[2] * cw + qd[3] * cx;
  float ez = qd[0] * cz + qd[1] * cy - qd[2] * cx + qd[3] * cw;
  float sg = ew > 0.f ? 1.f : (ew < 0.f ? -1.f : 0.f);
  ut[3] = -ex * sg; ut[4] = -ey * sg; ut[5] = -ez * sg;
  float nx = sqrtf(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]), na = sqrtf(ut[3] * ut[3] + ut[4] * ut[4] + ut[5] * ut[5]);
  const float sat_xyz = JOSC_VMAX_XYZ / JOSC_KP * JOSC_KV, sat_abg = JOSC_VMAX_ABG / JOSC_KO * JOSC_KV;
  float sx = nx > sat_xyz ? sat_xyz / nx : 1.f, sa = na > sat_abg ? sat_abg / na : 1.f;
  for (int k = 0; k < 3; k++) { ut[k] *= JOSC_KP * sx; ut[3 + k] *= JOSC_KO * sa; }
  wave_sync();   // Jm visible
  // J x = u_task: Gauss-Jordan over lanes 0..5 (lane = row), implicit partial pivoting (the pivot row of column k is the unused lane with
  // the largest |entry|; rows are never moved)
  float A[6], b = ut[0];
#pragma unroll
  for (int j = 0; j < 6; j++) A[j] = Jm[i * 6 + j];
#pragma unroll
  for (int j = 1; j < 6; j++) b = i == j ? ut[j] : b;
  bool used = lane >= 6;
  float detJ = 1.f, xs = 0.f;
  int prow[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float cand = used ? -1.f : fabsf(A[k]);
    float best = -1.f;
    int p = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { const float v = wave_bcast(cand, j); if (v > best) { best = v; p = j; } }
    prow[k] = p;
    const float piv = wave_bcast(A[k], p);
    detJ *= piv;
    const float inv = 1.f / piv;
    const float f = lane == p ? 0.f : A[k] * inv;
#pragma unroll
    for (int j = k + 1; j < 6; j++) A[j] -= f * wave_bcast(A[j], p);
    b -= f * wave_bcast(b, p);
    if (lane == p) { used = true; xs = inv; }
  }
  const float detX = detJ * detJ / detM;
  if (fabsf(detX) < 1e-3f || !(detX == detX)) {   // (wave-uniform) abr_control's SVD branch: rare
    wave_sync();
    stage_osc_general(m, s, lane, flags);
    return;
  }
  xs *= b;   // the lane that was pivot row of column k holds x_k
  float x[6];
#pragma unroll
  for (int k = 0; k < 6; k++) x[k] = wave_bcast(xs, prow[k]);
  wave_sync();
  if (lane < 6) {   // u = bias - M (kv dq + x)
    float u = s.bias[lane];
#pragma unroll
    for (int k = 0; k < 6; k++) u -= s.M[m_index(lane, k)] * (JOSC_KV * s.qvel[k] + x[k]);
    s.ctrl[lane] = u;
  }
  wave_sync();
}
#else
template <class L>
JDEV void stage_osc(const JacoModelDev* m, L& s, int lane, unsigned& flags) { stage_osc_general(m, s, lane, flags); }
#endif

// ---------------------------------------------------------------- a2: action -> target / gripper ramp (env_mujoco_util.py:602-646)
template <class L>
JDEV void take_action(const JacoModelDev* m, L& s, const float* act_in, int nact, int lane) {
  v3 pe; m3 Re;
  ee_frame(m, s, &pe, &Re);
  float eul[3];
  mat_to_euler_rxyz(Re, eul);
  if (lane == 0) {
    float* t = s.task;
    // np.clip(action, act_min, act_max) with act_max = -act_min = 1 (env_mujoco.py:117; a NaN stays a NaN, as under np.clip)
    float a_[7];
    for (int k = 0; k < 7; k++) { const float x = k < nact ? act_in[k] : 0.f; a_[k] = x > 1.f ? 1.f : (x < -1.f ? -1.f : x); }
    const float* act = a_;
    t[JT_TARGET + 0] = pe.x + act[0] / 25.f; t[JT_TARGET + 1] = pe.y + act[1] / 25.f; t[JT_TARGET + 2] = pe.z + act[2] / 25.f;
    t[JT_TARGET + 3] = eul[0] + act[3] / 5.f; t[JT_TARGET + 4] = eul[1] + act[4] / 5.f;
    float yaw = eul[2] + act[5] / 5.f;
    const float PI = 3.14159265358979323846f;
    if (fabsf(yaw) > PI) yaw += (yaw > 0.f ? -1.f : 1.f) * 2.f * PI;
    t[JT_TARGET + 5] = yaw;
    float prev = t[JT_GRIP], g = 0.6f;
    if (nact == 7) g = fminf(1.f, fmaxf(0.6f, prev + act[6] / 10.f));
    t[JT_GRIP_PREV] = nact == 7 ? prev : 0.6f;
    t[JT_GRIP] = g;
  }
}
