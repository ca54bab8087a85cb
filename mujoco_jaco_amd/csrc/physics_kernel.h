// Batched Jaco physics step for gfx950: one 64-lane wavefront (= one workgroup) per environment.
//
// Replaces, per environment, what the reference does in  Mujoco.send_forces -> sim.step()
// (/root/reference/env_script/mujoco.py:258-278): MuJoCo's mj_step under its default options
// (SURVEY.md section 8 row a6 / App. D.1).  Stage order inside one substep:
//   K  tree walk: frames, motion subspaces, velocities, bias accelerations   (lane = root->leaf path)
//   G  geom poses (lane = geom), body spatial inertias and RNE forces (lane = body)
//   M  composite inertias -> mass matrix (lane = dof), bias forces, actuation, qacc_smooth (LDL^T, lane = row)
//   C  collision: pair whitelist + bounding spheres + OBB cull (lane = pair), narrowphase (wave per pair)
//   R  constraint rows: joint limits + pyramidal contacts (lane = dof / lane = row)
//   S  primal Newton solve with exact line search (MuJoCo's default solver; see DESIGN.md for why not PGS)
//   T  touch sensors (lane = sensor)
//   E  semi-implicit Euler with implicit joint damping, quaternion integration for the free bodies
// Per-body / per-row state lives in LDS; vectors of length nv live one element per lane and are
// broadcast with v_readlane; small reductions use DPP butterflies (jaco/wave_ops.h).
#pragma once
#include <jaco/model_dev.h>
#include <jaco/wave_ops.h>

// Three capacity tiers of the same kernel (DESIGN.md "Tiers"):
//   light: <= 64 rows / 32 contacts / 64 candidates -> 1 constraint row per lane, ~17 KB LDS, the common case;
//   medium: <= 128 rows / 48 contacts -> 2 rows per lane, ~28 KB LDS, for envs that overflow light (marker-stick contacts);
//   heavy: <= 256 rows / 64 contacts -> 4 rows per lane, ~43 KB LDS, for envs that overflow medium;
//   huge: <= 512 rows / 128 contacts -> 8 rows per lane, ~80 KB LDS, for envs that overflow heavy (beyond that rows are dropped and flagged).
// A light wave that meets an overflow hands its env (state untouched for that substep) to the heavy launch.
// Constraint rows in body space (round 5; an A/B build option: -DJACO_WRENCH=1, default OFF -- measured slower, see below).  A contact row is
// J_r = w_r^T (J_B - J_A): a 6-vector wrench w_r = (pos x dir + rotational part, dir) applied to the motion subspaces S_k of the dofs that move
// body B but not body A (and with the opposite sign vice versa).  The row is stored as that wrench (8 floats with the solver's two per-row
// staging slots) plus the two body ids -- 48 bytes with its parameters instead of 100 -- and every product with J is formed from it:
//   J v      = w . D_c,  D_c = signed sum of S_k v_k over the dofs that move exactly one of the contact's two bodies (lane = (contact, component))
//   J^T D J  = the matrix-core pass generates its J entries on the fly: lane (column k) holds S_k and forms sgn_k(r) w_r . S_k per row
//   J^T f    = S_k . (sum of the wrenches of the contacts dof k moves)
// so the light tier holds 128 rows in the LDS that holds 64 dense ones (13 584 B, 168 VGPRs, 3 waves per SIMD either way; medium / heavy /
// huge: 14.7 / 22.1 / 39.5 KB instead of 20.2 / 43 / 80 KB).  Joint-limit rows (J = +-e_d) are "unit rows": flag + dof + sign.
// Parity: every emulator suite green (tests/, built with -DJACO_WRENCH=1).  MEASURED on MI355X (profiles/r05_ab_wrench_rows.txt): headline
// 1.84 -> 1.70 M env-steps/s, action scale 0.05 0.50 -> 0.21 M, policy-driven 1.33 -> 1.22 M -- the envs do stay in the light tier (bigger-tier
// share 0.60 -> 0.09) but every product with J now costs VALU work and dependent LDS round trips that a dense row in registers did not
// (rows_dot: 21 v_readlane + 21 fma -> two LDS hand-offs and a bit-scan loop; the matrix-core pass: 1 LDS read per entry -> 14 VALU): the
// dense rows stay the product path.
#ifndef JACO_WRENCH
#define JACO_WRENCH 0
#endif
template <int MAXEFC_, int MAXCON_, int MAXCAND_, bool CONTACT_ = true>
struct JacoCaps {
  static_assert(MAXEFC_ % 64 == 0, "rows are dealt out 64 at a time (one per lane): the row capacity must be a multiple of 64");
  static constexpr bool WRENCH = CONTACT_ && JACO_WRENCH != 0;
  static_assert(WRENCH || JMAXGEOM != 64 || JNV != 21 || MAXCAND_ <= MAXEFC_ * 25 - 520 - 256 - 576, "default layout: the candidate list must fit behind the geom poses in the constraint-row area");
  static constexpr int MAXEFC = MAXEFC_, MAXCON = MAXCON_, MAXCAND = MAXCAND_, NR = MAXEFC_ / 64;
  // CONTACT = false: the contact-free instantiation (arm-only models, option disable_contact): no geom poses, no contact list, and
  // Jacobian storage for the joint-limit rows only (at most one per body)
  static constexpr bool CONTACT = CONTACT_;
  static constexpr int JROWS = CONTACT_ ? MAXEFC_ : (JNB <= 16 ? 16 : 32), NGEOM = CONTACT_ ? JMAXGEOM : 1;
  static_assert(CONTACT_ || JNB <= 32, "limit rows of the contact-free instantiation");
};
// Candidates = bounding-sphere survivors (closed fingers alone contribute > 64; with the EE sticks on the marker's sticks 130-200): their
// list shares LDS with the constraint rows, which are bigger, so its capacity costs nothing -- and it must not be what sends
// an env to a bigger tier (round 2 found 74 % of the envs of a small-action rollout in the heavy tier because of a 128-entry list).
#if JACO_WRENCH
typedef JacoCaps<128, 32, 240> JacoLight;    // body-space rows: 128 rows x 48 B fit where 64 dense rows x 100 B did
#else
typedef JacoCaps<64, 32, 240> JacoLight;
#endif
typedef JacoCaps<128, 32, 512> JacoMedium;   // (32 contacts x 4 pyramid rows = 128 rows; 20.2 KB -> 8 envs per CU = the 2 waves per SIMD its 256 VGPRs allow)  2 rows per lane: the EE axis sticks resting on the "hand" marker's sticks add ~36 rows to the usual 32
#ifndef JACO_HEAVY_ROWS
#define JACO_HEAVY_ROWS 256
#define JACO_HEAVY_CON 64
#define JACO_HEAVY_WAVES 1
#endif
typedef JacoCaps<JACO_HEAVY_ROWS, JACO_HEAVY_CON, 512> JacoHeavy;
typedef JacoCaps<64, 1, 1, false> JacoArm;    // contact-free: 8.4 KB of LDS, 128 VGPRs -> 16 envs per CU = 4 waves per SIMD (BASELINE config 2: 4 096 envs resident in one round)
typedef JacoCaps<512, 128, 512> JacoHuge;    // 8 rows per lane: a reset that puts the hand inside the pedestal (1 % of picking resets: up to ~90 contacts / ~410 rows)
#define JDBG_MAXCON 64
#define JDBG_MAXEFC 256
#define JW_WT 6       // body-space row record: slot of the solver's row weight (D on active rows, else 0)
#define JW_X 7        // ... and of the row residual J a - aref
#define JW_UNIT (1 << 30)   // e_con flag of a joint-limit row: J = w[0] e_dof
#define JLD (JNV)     // row stride of per-row dof vectors in LDS (21: odd, conflict-free for lane-per-row access)

#define JFLAG_CON_OVERFLOW 1u
#define JFLAG_EFC_OVERFLOW 2u
#define JFLAG_CAND_OVERFLOW 4u
#define JFLAG_NAN 8u
#define JFLAG_SOLVER_MAXITER 16u
#define JFLAG_HEAVY_TIER 32u   // informational: env was stepped by the heavy tier at least once
#define JFLAG_BAIL_CAUSE_SHIFT 8   // informational, bits 8..16: (contacts | rows | candidates) overflow that made tier 0 / 1 / 2 hand the env on
#define JFLAG_TIER_RETURN 128u  // informational: the heavy tier handed the env back to the light code in mid-step

#define JMINVAL 1e-15f
#define JTASK_FLOATS 40   // per-env task row (layout: JT_* in env_logic.h)

// Diagnostic build only (-DJACO_PROFILE_STAGES): lane 0 accumulates shader-clock cycles per stage into
// JacoStepArgs::prof[env][JPROF_N]. The shipped library is built without it (no stamp executes).
#define JPROF_N 16
struct JProfCtx {
  unsigned long long tprev;
  unsigned long long* row;   // [JPROF_N] of this env, or nullptr
};
#ifdef JACO_PROFILE_STAGES
JDEV void jprof_stamp(JProfCtx& pc, int i, int lane) {
  unsigned long long t = __builtin_amdgcn_s_memtime();
  if (lane == 0 && pc.row) pc.row[i] += t - pc.tprev;
  pc.tprev = __builtin_amdgcn_s_memtime();
}
#define JSTAMP(i) jprof_stamp(pc, (i), lane)
#else
#define JSTAMP(i)
#endif
// -DJACO_NARROW_PROFILE (with JACO_PROFILE_STAGES; tools/gpu_stage_profile.py --narrow): the narrowphase split by what it ran -- slot 11 candidate
// records (L2 round trip), 12 box-box, 13 plane-*, 14 MPR hits, 1 MPR misses incl. the cached-direction test; the Newton solver's four intervals,
// the usual owners of 11 .. 14, all go to slot 6 in such a build
#ifdef JACO_NARROW_PROFILE
#define JSTAMP_NEWTON(i) JSTAMP(6)
#define JSTAMP_NARROW(i) JSTAMP(i)
#else
#define JSTAMP_NEWTON(i) JSTAMP(i)
#define JSTAMP_NARROW(i)
#endif

struct JacoStepArgs {
  const JacoModelDev* model;
  const float* hull;   // float4 per hull vertex
  float* qpos;         // [nenv][nq]
  float* qvel;         // [nenv][nv]
  float* qpos_lo;      // [nenv][nq] low-order part of the compensated state (qpos + qpos_lo is the state; |lo| <= ulp(qpos) / 2), or nullptr:
  float* qvel_lo;      // [nenv][nv]   starts at 0 and is dropped at the end of the launch
  float* qacc_ws;      // [nenv][nv]  warm start (= last qacc)
  const float* ctrl;   // [nenv][nu]
  float* sensordata;   // [nenv][nsensor]
  unsigned* flags;     // [nenv] sticky error bits
  int* stats;          // [nenv][4]: ncon, nefc, newton iterations, candidates (last substep) or nullptr
  int* remaining;      // [nenv] substeps left for the heavy tier (written by the light tier)
  // work queues of the three bigger tiers (q[0] medium, q[1] heavy, q[2] huge): env ids appended by whoever meets an overflow
  // (or, at the start of a step, by the light grid for envs whose last step needed that tier: `hint`), served by resident worker
  // workgroups while the light grid runs and by the drain launches after it
  struct Queue {
    int* list;          // [nenv] env ids; -1 = not yet published, -2 = taken by a worker
    int* count;         // [1] entries appended
    int* taken;         // [1] slots claimed by workers
    const int* limit;   // [1] workers beyond this index leave at once (sized from the previous launch's demand), or nullptr
    const int* reserve; // [1] workers beyond this index leave as soon as the queue has run dry (the others wait for late arrivals)
  } q[3];
  const int* routed_mark;  // [nenv] == launch_id: the env was queued for a bigger tier before the launch (hint > 0): not the light grid's
  int launch_id;
  int* light_left;     // [1] light-tier workgroups still running (0: the resident workers leave, the drains take the rest)
  int* hint;           // [nenv] highest tier (0..3) the env's last step really needed, or nullptr: where its next step starts
  int nenv, nsub, disable_contact;
  int no_pairlist;     // 1: the bounding-sphere phase tests every pair in every substep (option "pair_list" = 0: comparison runs)
  int mpr_pairs;       // 1: two hull candidates in a row go through MPR side by side, one per half wave (option "mpr_pairs"; collision.h mpr_pair2)
  float* sepdir;       // [nenv][JMAXPAIR][4] cached separating direction of every hull pair (collision.h), or nullptr (option "sep_cache" = 0)
  int handdown;        // 1: a heavy-tier workgroup (4 per CU) passes an env that has calmed down on to the medium queue instead of running the
                       //    medium / light code itself for the rest of the step (first heavy drain only: a second medium drain follows it)
  int hint_mode;       // 1: an env's next step starts in the biggest tier this step really needed; 2: in the tier its last substep needed
  int no_tier_return;  // 1: an env handed to the heavy tier stays there for the rest of the launch (option "tier_return" = 0)
  // env-level mode (jaco_step / jaco_reset): nsub = frame_skip
  int env_mode;              // 0 ctrl-level, 1 env step, 2 forward only (reset: fill cache + observation),
                             // 3 placing reset: nsub controlled substeps with the object pinned in the hand (env_mujoco_util.py:106-117)
                             // 4 take_action only (env_mujoco.py:158-159), 5 terminal_inspection only (env_mujoco.py:144-150): no physics
                             // 6 grasping reset: the two pre-reach loops (env_mujoco_util.py:123-170), at most nsub substeps, then the observation
  const unsigned char* mask; // modes 2 and 3: envs to run (nullptr = all)
  float* marker;             // [nenv][2][12] poses (position, rotation) of the "hand" / "subgoal_reach" markers, or nullptr = XML rest pose
  int task_id, nact;
  unsigned long long seed;
  float* task;               // [nenv][JTASK_N]
  float* cache;              // [nenv][JCACHE_N]
  const float* action;       // [nenv][nact]
  const float* noise;        // optional [nenv][12] sub-goal noise draws (6 for the marker, 6 for the observation), else RNG
  int obs_mode;              // 0: rule-based sub-goal in obs[17:23] (what main.py selects), 1: the reaching goal (rulebased_subgoal = False, env_mujoco_util.py:255-270)
  int auto_reset;            // mode 1, option "auto_reset": an env whose step ends its episode is reset (draws of _reset + sim.forward() + first
                             // observation) by the wave that finished it, instead of by a separate masked jaco_reset launch chain
  const float* qpos0;        // [nq] reset pose (auto_reset)
  const float* goal_buf;     // auto_reset, optional: recorded reaching goals (jaco_set_init_buffer), [goal_n][goal_stride]
  int goal_n, goal_stride;
  const float* subgoal;      // obs_mode 1, optional [nenv][6]: the policy's sub-goal offset; "subgoal_reach" marker = it + the previous target (:609)
  float* obs;                // [nenv][26]
  float* reward;             // [nenv]
  unsigned char* done;       // [nenv]
  float* terminal;           // [nenv][2] (success flag, wb) of the env's most recent terminal step (jaco_get_last_terminal), or nullptr
  float* terminal_obs;       // [nenv][26] observation of that terminal step (auto_reset replaces the env's obs row with the new episode's first), or nullptr
  unsigned* cost;            // [nenv] shader-clock ticks (>> 4) the env's last step took (launch-order heuristic), or nullptr
  const int* order;          // light tier, optional: workgroup -> env permutation (expensive envs first), else identity
  const int* nslots;         // light tier, optional: [1] number of valid entries of `order` (masked resets launch a small grid over the list of reset envs), else nenv
  unsigned long long* prof;  // diagnostic build only: [nenv][JPROF_N] cycle sums, else nullptr
  float* dbg;          // optional stage dump of env dbg_env (see JDBG_* offsets), else nullptr
  int dbg_env;
};

// The kernel's argument block, read afresh from the kernarg segment.  The block holds ~50 pointers (100 SGPRs' worth); read through
// the by-value parameter they are all loaded at kernel entry and the ones the epilogue needs stay live -- in SGPRs, in VGPR lanes
// (v_writelane / v_readlane spill code around every SGPR-hungry stretch of the substep loop) and, as per-lane row addresses, in
// scratch -- across the whole substep loop.  run_env re-reads the block at the start of the env, of every substep and of the
// epilogue instead: an s_load per use, nothing carried.  Every kernel that reaches run_env takes the block as its only parameter.
#ifdef JACO_EMULATED
JDEV const JacoStepArgs* args_view(const JacoStepArgs& A) { return &A; }
#else
JDEV const JacoStepArgs* args_view(const JacoStepArgs&) {
  typedef const JacoStepArgs __attribute__((address_space(4))) * KP;
  KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return (const JacoStepArgs*)p;
}
#endif

// debug dump layout (floats)
#define JDBG_XPOS 0                       // [JNB][3]
#define JDBG_XMAT (JDBG_XPOS + 3 * JNB)   // [JNB][9]
#define JDBG_M (JDBG_XMAT + 9 * JNB)      // [JNV][JNV]
#define JDBG_BIAS (JDBG_M + JNV * JNV)    // [JNV]
#define JDBG_SMOOTH (JDBG_BIAS + 24)      // qfrc_smooth
#define JDBG_QACC_SMOOTH (JDBG_SMOOTH + 24)
#define JDBG_QACC (JDBG_QACC_SMOOTH + 24)
#define JDBG_QFRC_CON (JDBG_QACC + 24)
#define JDBG_NCON (JDBG_QFRC_CON + 24)    // ncon, nefc, iters, ncand
#define JDBG_CONTACT (JDBG_NCON + 4)      // [JDBG_MAXCON][8]: dist, pos3, normal3, pair
#define JDBG_EFC (JDBG_CONTACT + 8 * JDBG_MAXCON)  // [JDBG_MAXEFC][4]: aref, R, x(final jar), force
#define JDBG_GPOS (JDBG_EFC + 4 * JDBG_MAXEFC)     // [JMAXGEOM][3]
#define JDBG_CFN (JDBG_GPOS + 3 * JMAXGEOM)      // [JMAXCON] per-contact normal force seen by the touch stage
#define JDBG_SENS (JDBG_CFN + JDBG_MAXCON)         // [JNSENS] sensordata
#define JDBG_SIZE (JDBG_SENS + JNSENS)

// The mass matrix is block diagonal over the dof blocks [0,JB0) arm + fingers, [JB0,JB1) object, [JB1,JNV) pedestal (one block per
// kinematic tree): only the blocks are stored, row-major one after the other (153 instead of 441 floats).
#define JMBLK (JB0 * JB0 + (JB1 - JB0) * (JB1 - JB0) + (JNV - JB1) * (JNV - JB1))
JDEV int m_index(int d, int j) {   // (d, j in the same block)
  return d < JB0 ? d * JB0 + j : (d < JB1 ? JB0 * JB0 + (d - JB0) * (JB1 - JB0) + (j - JB0) : JB0 * JB0 + (JB1 - JB0) * (JB1 - JB0) + (d - JB1) * (JNV - JB1) + (j - JB1));
}
// Floats of the constraint-row area that the early stages of a substep use as scratch (stage_walk, stage_mass_bias, stage_osc): S_d qvel_d and
// S_d-dot qvel_d ([JNV][6] each) from its start, the frame records of the tree walk ([JNB + 2][16]) from JTB_OFF.
#define JTB_OFF (12 * JNV > 256 ? 12 * JNV : 256)
#define JSCRATCH (JTB_OFF + 16 * (JNB + 2) > 520 ? JTB_OFF + 16 * (JNB + 2) : 520)
// state rows in LDS (sizes of the default layout are the floor: 24 / 24 / 12)
#define JQ_LDS (JNQ + 1 > 24 ? JNQ + 1 : 24)
#define JV_LDS (JNV + 3 > 24 ? JNV + 3 : 24)
#define JU_LDS (JNU + 3 > 12 ? JNU + 3 : 12)
// 64-lane passes of the frame-column lanes (4 per body + the two markers) and of the geom lanes
#define JFRAME_PASSES ((4 * (JNB + 2) + 63) / 64)
#define JGEOM_PASSES (JMAXGEOM / 64)

// Per-env working state.  LDS is what caps the number of resident envs per CU (160 KB / sizeof), so arrays whose lifetimes inside
// a substep do not overlap share storage:
//   * body inertias / forces (tree walk .. mass matrix)            with   the contact list (collision .. Euler);
//   * geom poses + broadphase survivors (tree walk .. collision)   with   the constraint rows (row builders .. Euler), behind the
//     first JSCRATCH floats of that area, which the early stages use as scratch.
// Light tier: 13.3 KB -> 12 envs per CU (3 waves per SIMD); it was 20.5 KB -> 8.
// (body-space rows only: per-body dof chain masks; an empty base otherwise -- four more bytes take the medium tier from 20 480 B = 8 workgroups
//  per CU to 20 496 B = 7)
template <bool W> struct JacoChainTab { unsigned b_chain[JNB]; };   // bit d set: dof d moves body b
template <> struct JacoChainTab<false> {};
template <class C>
struct JacoLDS {
  typedef C Caps;
  float qpos[JQ_LDS], qvel[JV_LDS], qacc_ws[JV_LDS], ctrl[JU_LDS];
  float qpos_lo[JQ_LDS], qvel_lo[JV_LDS];       // compensated state: what every stage reads is the fp32 rounding (qpos, qvel) of hi + lo
  float xpos[JNB][3], xmat[JNB][9];
  float cdof[JNV][6];
  float cvel[JNB][6];
  float M[JMBLK];
  float bias[JV_LDS], smooth[72];               // smooth[0..nv); the light tier also stages its <= 64 row residuals here (MFMA pass)
  float mk[24];                                 // poses of the two task-layer markers ("hand", "subgoal_reach") during this launch
  union {
    struct {                                    // tree walk .. mass matrix
      alignas(16) float cinert[JNB][10];        // (16-byte aligned: cinert+crb double as the frame scratch of stage K)
      float crb[JNB][10];
      float cacc[JNB][6], cfrc[JNB][6];
    };
    struct {                                    // collision .. Euler: contacts
      float c_dist[C::MAXCON], c_pos[C::MAXCON][3], c_frame[C::MAXCON][9], c_fn[C::MAXCON];
      int c_pair[C::MAXCON], c_efc[C::MAXCON];
      unsigned c_m1[C::MAXCON], c_m2[C::MAXCON];   // dof chain masks of the two bodies
      int c_ob[C::MAXCON];                          // per geom: original (unfused) body id | (fused body + 1) << 8; geom 2 in the upper half
      int c_dim[C::MAXCON];
    };
  };
  union {
    struct {                                    // row builders .. Euler: constraint rows
      // dense rows: J[r][JLD]; body-space rows (C::WRENCH): 8 floats per row = wrench (angular 3, linear 3), then the solver's two staging
      // slots (JW_WT: D * active, JW_X: residual); never smaller than the JSCRATCH floats the early stages borrow
      alignas(16) float J[C::WRENCH ? (8 * C::MAXEFC > JSCRATCH ? 8 * C::MAXEFC : JSCRATCH) : C::JROWS * JLD];
      float e_aref[C::MAXEFC], e_D[C::MAXEFC], e_f[C::MAXEFC];
      int e_con[C::MAXEFC];                     // contact | edge << 8 | block bits << 16; body-space rows: | (body A + 1) << 19 | (body B + 1) << 24 (0: static),
                                                // joint-limit rows: JW_UNIT | dof << 19
    };
    struct {                                    // tree walk .. collision: geom poses, broadphase survivors
      float early_scratch[JSCRATCH];
      alignas(16) float gpos[C::NGEOM][4];      // world position + bounding radius: one 16-byte LDS read per geom in the broadphase
      float gmat[C::NGEOM][9];
      int cand[C::MAXCAND];
    };
  };
  float e_x[(C::MAXEFC > 64 && !C::WRENCH) ? C::MAXEFC : 1];    // dense rows, bigger tiers: residuals staged for the MFMA pass (light reuses `smooth`)
  int ncon, nefc, ncand, nlimit, nsphere;      // (ncand: narrowphase candidates after the OBB cull; nsphere: bounding-sphere survivors before it)
  int nside, nside_cand;                       // rows in the side buffer (light tier, split mode; collision.h) / rows that could go there (every tier)
  float task[JTASK_FLOATS];                    // (= JTASK_N of env_logic.h)
  float osc_qd[4];                              // target orientation quaternion of the current env step (constant over its substeps)
  // per-launch copy of the small, hot model tables (per-lane gathers from LDS instead of dependent global loads)
  struct McTab : JacoChainTab<C::WRENCH> {
    float b_pos[JNB][3], b_mat[JNB][9], b_axis[JNB][3], b_com[JNB][3], b_qpos0[JNB];
    int b_jtype[JNB], b_qadr[JNB], b_dadr[JNB], b_parent[JNB];
    int inner_body[JMAXINNER];
    unsigned b_descmask[JNB];
    int b_anc[JNB][3];                          // ancestors 1, 2 and 4 levels up (-1: none), for the pointer-jumping tree stages
    int d_body[JNV], d_parent[JNV];
    int q_dof[JNQ + 1];                         // dof that advances position coordinate q linearly (hinge angle, free-body translation), -1: quaternion component
  } mc;
};
static_assert((JB1 - JB0 == 6 || JB1 == JB0) && (JNV - JB1 == 6 || JNV == JB1) && JB0 >= 6 && JNV <= 31 && JNV - JB0 <= 15 && JNV <= 64,
              "dof blocks: the arm tree(s) of at least the six arm joints, then zero, one or two free bodies; dofs + the residual column fit the 32-column matrix-core tile");

template <class L>
JDEV void stage_model(const JacoModelDev* m, L& s, int lane) {
  if (lane < JNB) {
    int b = lane;
    for (int k = 0; k < 3; k++) { s.mc.b_pos[b][k] = m->b_pos[b][k]; s.mc.b_axis[b][k] = m->b_axis[b][k]; s.mc.b_com[b][k] = m->b_com[b][k]; }
    for (int k = 0; k < 9; k++) s.mc.b_mat[b][k] = m->b_mat[b][k];
    s.mc.b_qpos0[b] = m->b_qpos0[b]; s.mc.b_jtype[b] = m->b_jtype[b]; s.mc.b_qadr[b] = m->b_qadr[b]; s.mc.b_dadr[b] = m->b_dadr[b];
    int p1 = m->b_parent[b], p2 = p1 >= 0 ? m->b_parent[p1] : -1, p3 = p2 >= 0 ? m->b_parent[p2] : -1, p4 = p3 >= 0 ? m->b_parent[p3] : -1;
    s.mc.b_parent[b] = p1;
    s.mc.b_descmask[b] = m->b_descmask[b];
    if constexpr (L::Caps::WRENCH) s.mc.b_chain[b] = m->b_chainmask[b];
    if (b < JMAXINNER) s.mc.inner_body[b] = m->inner_body[b];
    s.mc.b_anc[b][0] = p1; s.mc.b_anc[b][1] = p2; s.mc.b_anc[b][2] = p4;
    const int qa = m->b_qadr[b], da = m->b_dadr[b];
    if (b < m->nbody) {
      if (m->b_jtype[b] == JJ_HINGE) s.mc.q_dof[qa] = da;
      else { for (int k = 0; k < 3; k++) s.mc.q_dof[qa + k] = da + k; for (int k = 3; k < 7; k++) s.mc.q_dof[qa + k] = -1; }
    }
  }
  if (lane < JNV) { s.mc.d_body[lane] = m->d_body[lane]; s.mc.d_parent[lane] = m->d_parent[lane]; }
}

// ---------------------------------------------------------------- small vector helpers
struct v3 { float x, y, z; };
JDEV v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
JDEV v3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
JDEV void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
struct alignas(16) v4 { float x, y, z, w; };
JDEV v4 ld4(const float* p) { return *reinterpret_cast<const v4*>(p); }   // p must be 16-byte aligned
JDEV v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
JDEV v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
JDEV v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
JDEV v3 operator*(float s, v3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
JDEV v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
JDEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
JDEV v3 cross(v3 a, v3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
JDEV float norm(v3 a) { return sqrtf(dot(a, a)); }
JDEV v3 normalized(v3 a) {
  float n = norm(a);
  return n < JMINVAL ? mk3(1.f, 0.f, 0.f) : a * (1.f / n);
}
// row-major 3x3
struct m3 { float m[9]; };
JDEV m3 ldm(const float* p) { m3 r; for (int i = 0; i < 9; i++) r.m[i] = p[i]; return r; }
JDEV void stm(float* p, const m3& a) { for (int i = 0; i < 9; i++) p[i] = a.m[i]; }
JDEV v3 mul(const m3& a, v3 v) {
  return mk3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
JDEV v3 mulT(const m3& a, v3 v) {
  return mk3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z, a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
JDEV m3 mul(const m3& a, const m3& b) {
  m3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
JDEV v3 col(const m3& a, int k) { return mk3(a.m[k], a.m[3 + k], a.m[6 + k]); }
JDEV m3 quat2mat(float w, float x, float y, float z) {
  m3 r;
  r.m[0] = w * w + x * x - y * y - z * z; r.m[1] = 2 * (x * y - w * z); r.m[2] = 2 * (x * z + w * y);
  r.m[3] = 2 * (x * y + w * z); r.m[4] = w * w - x * x + y * y - z * z; r.m[5] = 2 * (y * z - w * x);
  r.m[6] = 2 * (x * z - w * y); r.m[7] = 2 * (y * z + w * x); r.m[8] = w * w - x * x - y * y + z * z;
  return r;
}
// sin and cos of a joint angle: two-constant Cody-Waite reduction to [-pi/4, pi/4] + the cephes sinf/cosf minimax
// polynomials (< 1.5 ulp for |x| < 1e4; beyond that the library routine with its large-argument reduction takes over).
JDEV void joint_sincos(float x, float* sn, float* cs) {
  if (fabsf(x) > 1.0e4f) { sincosf(x, sn, cs); return; }
  float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(k, -1.5707963705062866f, x);
  r = fmaf(k, 4.3711390001862412e-8f, r);
  float z = r * r;
  float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.f));
  int q = (int)k;
  float a = (q & 1) ? pc : ps, b = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -a : a;
  *cs = ((q + 1) & 2) ? -b : b;
}
// rotation by angle about unit axis (Rodrigues)
JDEV m3 axis_rot(v3 a, float ang, float ang_lo = 0.f) {
  float s, c;
  joint_sincos(ang, &s, &c);
  { const float s0 = s; s = fmaf(ang_lo, c, s); c = fmaf(-ang_lo, s0, c); }   // first-order in the low part of the angle (|ang_lo| < 3e-7)
  float t = 1.f - c;
  m3 r;
  r.m[0] = c + t * a.x * a.x; r.m[1] = t * a.x * a.y - s * a.z; r.m[2] = t * a.x * a.z + s * a.y;
  r.m[3] = t * a.x * a.y + s * a.z; r.m[4] = c + t * a.y * a.y; r.m[5] = t * a.y * a.z - s * a.x;
  r.m[6] = t * a.x * a.z - s * a.y; r.m[7] = t * a.y * a.z + s * a.x; r.m[8] = c + t * a.z * a.z;
  return r;
}
// spatial vectors about the world origin: motion (w, v), force (n, f)
struct sv { v3 a, b; };
JDEV sv ldsv(const float* p) { sv r; r.a = ld3(p); r.b = ld3(p + 3); return r; }
JDEV void stsv(float* p, sv s) { st3(p, s.a); st3(p + 3, s.b); }
JDEV sv operator+(sv x, sv y) { sv r; r.a = x.a + y.a; r.b = x.b + y.b; return r; }
JDEV sv operator*(sv x, float s) { sv r; r.a = x.a * s; r.b = x.b * s; return r; }
JDEV float dot(sv x, sv y) { return dot(x.a, y.a) + dot(x.b, y.b); }
JDEV sv cross_motion(sv v, sv s) { sv r; r.a = cross(v.a, s.a); r.b = cross(v.a, s.b) + cross(v.b, s.a); return r; }
JDEV sv cross_force(sv v, sv f) { sv r; r.a = cross(v.a, f.a) + cross(v.b, f.b); r.b = cross(v.a, f.b); return r; }
// spatial inertia [m, h = m c, I_O (xx yy zz xy xz yz)] applied to a motion vector
JDEV sv inert_mul(const float* I, sv mv) {
  v3 h = mk3(I[1], I[2], I[3]);
  sv r;
  r.b = mv.b * I[0] + cross(mv.a, h);
  v3 t = cross(h, mv.b);
  r.a = mk3(I[4] * mv.a.x + I[7] * mv.a.y + I[8] * mv.a.z + t.x, I[7] * mv.a.x + I[5] * mv.a.y + I[9] * mv.a.z + t.y,
            I[8] * mv.a.x + I[9] * mv.a.y + I[6] * mv.a.z + t.z);
  return r;
}

// ---------------------------------------------------------------- compensated state (hi + lo floats)
// The state the integrators advance is the unevaluated sum hi + lo (|lo| <= ulp(hi) / 2, ~48 significant bits): per-step fp32
// rounding of qpos (2.4e-7 at q ~ 4 rad) would otherwise be ~4 000 x the rounding of the increment h * qvel itself and
// random-walks the state away from the fp64 trajectory (tools/drift_control.py variants B / E / F).
// Every stage still computes in fp32 on the rounded view `hi`; rounding perturbs each evaluation but no longer accumulates.
struct f2 { float hi, lo; };
// (hi, lo) + (inc + inc_lo): Knuth two-sum for the leading parts, then renormalisation.  No products inside (nothing to contract).
JDEV f2 comp_add(float hi, float lo, float inc, float inc_lo) {
  const float s = hi + inc, bb = s - hi;
  const float e = (hi - (s - bb)) + (inc - bb);
  const float l = lo + (e + inc_lo);
  f2 r;
  r.hi = s + l;
  r.lo = l - (r.hi - s);
  return r;
}
// (hi, lo) + h * (v + v_lo) with the step h = hh + hl (the fp64 value of opt.timestep): exact product error by fma
JDEV f2 comp_advance(float hi, float lo, float hh, float hl, float v, float v_lo) {
  const float p = fmul_rn(hh, v), pe = fmaf(hh, v, -p);
  return comp_add(hi, lo, p, pe + fmaf(hh, v_lo, hl * v));
}

// ---------------------------------------------------------------- LDL^T solve, one matrix row per lane
// h[j] = A[lane][j] (lanes >= n must hold identity rows), b = rhs[lane]; returns x[lane]. Registers only.
// FULL = false: A is block diagonal over the dof blocks [0,JB0) [JB0,JB1) [JB1,JNV) (always true for the mass
// matrix; true for the Newton Hessian unless an active row couples two blocks): 66 instead of 210 eliminations.
template <bool FULL>
JDEV float ldl_solve(float (&h)[JNV], float b, int lane) {
  float dinv = 1.f;
#pragma unroll
  for (int k = 0; k < JNV; k++) {
    const int jend = FULL ? JNV : (k < JB0 ? JB0 : (k < JB1 ? JB1 : JNV));
    float dk = wave_bcast(h[k], k);
    float inv = 1.f / dk;
    dinv = lane == k ? inv : dinv;
    float lik = lane > k ? h[k] * inv : 0.f;
    // (all of the pivot row's broadcasts first, then the updates: a v_readlane straight in front of the VALU that reads its SGPR
    // costs two wait states on gfx950, and the compiler pairs them up when the source does)
    float pk[JNV];
#pragma unroll
    for (int j = k + 1; j < jend; j++) pk[j] = wave_bcast(h[j], k);
    const float bk = wave_bcast(b, k);
#pragma unroll
    for (int j = k + 1; j < jend; j++) h[j] -= lik * pk[j];
    b -= lik * bk;
  }
  // now: b = y (L y = rhs); h[k>lane] = d_lane * L[k][lane]
  float z = b * dinv, acc = 0.f, x = 0.f;
#pragma unroll
  for (int k = JNV - 1; k >= 0; k--) {
    x = lane == k ? z - dinv * acc : x;
    float xk = wave_bcast(x, k);
    acc += lane < k ? h[k] * xk : 0.f;
  }
  return x;
}
// Block-diagonal variant that factors only the dof blocks named in `mask` (bit 0: [0,JB0), bit 1: [JB0,JB1), bit 2: [JB1,JNV));
// lanes of skipped blocks get x = 0.
template <int LO, int HI>
JDEV float ldl_block(float (&h)[JNV], float b, int lane) {
  float dinv = 1.f;
#pragma unroll
  for (int k = LO; k < HI; k++) {
    float dk = wave_bcast(h[k], k);
    float inv = 1.f / dk;
    dinv = lane == k ? inv : dinv;
    float lik = (lane > k && lane < HI) ? h[k] * inv : 0.f;
    float pk[JNV];
#pragma unroll
    for (int j = k + 1; j < HI; j++) pk[j] = wave_bcast(h[j], k);
    const float bk = wave_bcast(b, k);
#pragma unroll
    for (int j = k + 1; j < HI; j++) h[j] -= lik * pk[j];
    b -= lik * bk;
  }
  float z = b * dinv, acc = 0.f, x = 0.f;
#pragma unroll
  for (int k = HI - 1; k >= LO; k--) {
    x = lane == k ? z - dinv * acc : x;
    float xk = wave_bcast(x, k);
    acc += (lane < k && lane >= LO) ? h[k] * xk : 0.f;
  }
  return (lane >= LO && lane < HI) ? x : 0.f;
}
JDEV float ldl_solve_blocks(float (&h)[JNV], float b, int lane, int mask) {
  float x = 0.f;
  if (mask & 1) x += ldl_block<0, JB0>(h, b, lane);
  if (mask & 2) x += ldl_block<JB0, JB1>(h, b, lane);
  if (mask & 4) x += ldl_block<JB1, JNV>(h, b, lane);
  return x;
}
// Arm/finger block [0, JB0) with implicit joint damping: x = M^-1 b and xd = (M + diag(hd))^-1 b from ONE elimination where
// the two matrices agree.  hd is non-zero only on the last JB0 - NSH dofs (the finger joints, leaves of the arm's tree), so
// the first NSH pivots, their multipliers and the forward-substituted right-hand side are identical; only the trailing
// (JB0 - NSH) x (JB0 - NSH) Schur complement differs (by hd on its diagonal).  Returns x, *xd_out = xd (0 outside the block).
#define JLDL_NSH 6
JDEV float ldl_block0_dual(float (&h)[JNV], float b, float hd, int lane, float* xd_out) {
  float dinv = 1.f;
#pragma unroll
  for (int k = 0; k < JLDL_NSH; k++) {
    float dk = wave_bcast(h[k], k);
    float inv = 1.f / dk;
    dinv = lane == k ? inv : dinv;
    float lik = (lane > k && lane < JB0) ? h[k] * inv : 0.f;
    float pk[JNV];
#pragma unroll
    for (int j = k + 1; j < JB0; j++) pk[j] = wave_bcast(h[j], k);
    const float bk = wave_bcast(b, k);
#pragma unroll
    for (int j = k + 1; j < JB0; j++) h[j] -= lik * pk[j];
    b -= lik * bk;
  }
  // trailing block, twice: plain (h, b, dinv) and damped (g, bd, dinvd)
  float g[JB0 - JLDL_NSH], bd = b, dinvd = dinv;
#pragma unroll
  for (int j = JLDL_NSH; j < JB0; j++) g[j - JLDL_NSH] = h[j] + (lane == j ? hd : 0.f);
#pragma unroll
  for (int k = JLDL_NSH; k < JB0; k++) {
    float dk = wave_bcast(h[k], k), dkd = wave_bcast(g[k - JLDL_NSH], k);
    float inv = 1.f / dk, invd = 1.f / dkd;
    dinv = lane == k ? inv : dinv;
    dinvd = lane == k ? invd : dinvd;
    const bool below = lane > k && lane < JB0;
    float lik = below ? h[k] * inv : 0.f, likd = below ? g[k - JLDL_NSH] * invd : 0.f;
    float pk[JNV], pg[JNV];
#pragma unroll
    for (int j = k + 1; j < JB0; j++) { pk[j] = wave_bcast(h[j], k); pg[j] = wave_bcast(g[j - JLDL_NSH], k); }
    const float bk = wave_bcast(b, k), bdk = wave_bcast(bd, k);
#pragma unroll
    for (int j = k + 1; j < JB0; j++) { h[j] -= lik * pk[j]; g[j - JLDL_NSH] -= likd * pg[j]; }
    b -= lik * bk;
    bd -= likd * bdk;
  }
  // back-substitution: h[k > lane] = d_lane * L[k][lane] (shared for k < NSH columns... rows < NSH use h for both systems)
  float z = b * dinv, zd = bd * dinvd, acc = 0.f, accd = 0.f, x = 0.f, xd = 0.f;
#pragma unroll
  for (int k = JB0 - 1; k >= 0; k--) {
    x = lane == k ? z - dinv * acc : x;
    xd = lane == k ? zd - dinvd * accd : xd;
    float xk = wave_bcast(x, k), xdk = wave_bcast(xd, k);
    // multiplier of row k in column `lane`: the damped system's differs only inside the trailing block
    float mult = lane < k ? h[k] : 0.f;
    float multd = (lane < k) ? ((k >= JLDL_NSH && lane >= JLDL_NSH) ? g[k - JLDL_NSH] : h[k]) : 0.f;
    acc += mult * xk;
    accd += multd * xdk;
  }
  *xd_out = lane < JB0 ? xd : 0.f;
  return lane < JB0 ? x : 0.f;
}
// inclusive prefix sum over lanes (6 ds_bpermute steps)
JDEV int wave_scan_incl(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = wave_shfl_i(v, lane >= o ? lane - o : lane);
    v += lane >= o ? t : 0;
  }
  return v;
}
JDEV void load_rows(float (&h)[JNV], const float* M, int nv, int lane) {
#pragma unroll
  for (int j = 0; j < JNV; j++) h[j] = lane < nv ? M[lane * JNV + j] : (lane == j ? 1.f : 0.f);
}

// ---------------------------------------------------------------- stage K: kinematic tree
// Frames are handled column-wise: lane 4 b + c owns column c of body b's frame (c < 3: rotation column, c = 3: position), a 16-byte
// record slot, so that composing with a parent is one 3 x 3 times vector product per lane instead of a matrix product per body.
// K1: joint-local transform T_b = [R_b0 Rot(axis, q - q0) | pos_b] (free bodies: world pose from qpos).
// K2: world frame = product of the ancestors' T, by pointer jumping: round r composes each body's partial product with the one 2^r
//     levels up (3 rounds cover the 7-body chain link1..link6, finger).  The two task-layer markers ride along as records nb, nb + 1.
// K3 (one interval, everything that only needs the frames): spatial inertias (lane 4 b + i: row i of R I R^T; lane 4 b + 3: mass,
//     first moment), geom poses (lane = geom), motion subspaces S_d and S_d * qvel_d (lane = dof).
// K4: body velocity = sum of S_d qvel_d over the dofs that move it (bit mask of the body's dof chain), lanes 4 b (angular), 4 b + 1 (linear);
// K5 (lane = dof): S_d-dot * qvel_d with the velocity "before" that dof;  K6 (lane = body): bias acceleration = -gravity + sum over
//     the chain, and the body's RNE force.
// Scratch: s.cinert/s.crb (rebuilt in K3) and the not-yet-built constraint-row area s.J.
JDEV void ld_cols(const float* T, v3& c0, v3& c1, v3& c2, v3& p) {   // 16 floats, 16-byte aligned: three rotation columns, position
  v4 a = ld4(T), b = ld4(T + 4), c = ld4(T + 8), d = ld4(T + 12);
  c0 = mk3(a.x, a.y, a.z); c1 = mk3(b.x, b.y, b.z); c2 = mk3(c.x, c.y, c.z); p = mk3(d.x, d.y, d.z);
}
JDEV void st_col(float* T, v3 v) { v4 a; a.x = v.x; a.y = v.y; a.z = v.z; a.w = 0.f; *reinterpret_cast<v4*>(T) = a; }
// R v with R given by its columns: the same expression, element by element, as mul(m3, v3)
JDEV v3 mul_cols(v3 c0, v3 c1, v3 c2, v3 v) {
  return mk3(c0.x * v.x + c1.x * v.y + c2.x * v.z, c0.y * v.x + c1.y * v.y + c2.y * v.z, c0.z * v.x + c1.z * v.y + c2.z * v.z);
}
// sum of the 6-vectors T[d] over the dofs d of a chain mask.  A body is moved by at most JMAXCHAIN dofs (6 arm joints + its own
// finger joint; a free body's 6): the loads of all terms are issued together (one LDS round trip) instead of one dependent
// iteration per set bit.
JDEV sv chain_sum(const float* T, unsigned mask) {
  sv t[JMAXCHAIN];
  bool on[JMAXCHAIN];
#pragma unroll
  for (int k = 0; k < JMAXCHAIN; k++) {
    on[k] = mask != 0u;
    t[k] = ldsv(T + 6 * (on[k] ? __builtin_ctz(mask) : 0));
    mask &= mask - 1u;
  }
  sv v; v.a = v.b = mk3(0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < JMAXCHAIN; k++) if (on[k]) v = v + t[k];
  return v;
}
JDEV v3 chain_sum3(const float* T, unsigned mask) {   // one half (3 components) of the same sum, same order
  v3 t[JMAXCHAIN];
  bool on[JMAXCHAIN];
#pragma unroll
  for (int k = 0; k < JMAXCHAIN; k++) {
    on[k] = mask != 0u;
    t[k] = ld3(T + 6 * (on[k] ? __builtin_ctz(mask) : 0));
    mask &= mask - 1u;
  }
  v3 v = mk3(0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < JMAXCHAIN; k++) if (on[k]) v = v + t[k];
  return v;
}
static_assert(JTB_OFF % 4 == 0 && JTB_OFF + 16 * (JNB + 2) <= JSCRATCH && 16 * JNB <= 20 * JNB && JNB <= 64, "frame records: scratch sizes, one lane per body");
// (every per-slot loop below runs JFRAME_PASSES / JGEOM_PASSES times: once in the default layout -- slot = lane -- where it unrolls to straight-line code)
template <class L>
JDEV void stage_walk(const JacoModelDev* m, L& s, int lane, bool markers, JProfCtx* wpc = nullptr) {
#ifdef JACO_WALK_PROFILE   // diagnostic: split this stage over profile slots 9..14 (their usual owners are wrong in such a build)
#define JWSTAMP(i) if (wpc) jprof_stamp(*wpc, (i), lane)
#else
#define JWSTAMP(i)
#endif
  (void)wpc;
  const int nb = m->nbody, nv = m->nv;
  float* TA = &s.cinert[0][0];   // [JNB][16] frame records, ping  (cinert+crb are contiguous: 2 * JNB * 10 floats)
  float* TB = s.J + JTB_OFF;     // [JNB + 2][16] frame records, pong  (the first JSCRATCH floats of the constraint-row area are free at this point)
  float* Sq = s.J;               // [JNV][6] S_d * qvel_d
  float* Sq2 = s.J + JNV * 6;    // [JNV][6] S_d-dot * qvel_d
  const bool isb = lane < nb;
  const unsigned chainb = isb ? m->b_chainmask[isb ? lane : 0] : 0u;   // dofs that move body `lane`
  constexpr bool GEOMS = L::Caps::CONTACT;   // (the contact-free instantiation has no use for geom poses)
  // per frame slot (pass p: slot 64 p + lane = column fc of body fb)
  int fbA[JFRAME_PASSES], fcA[JFRAME_PASSES], bA[JFRAME_PASSES], jtA[JFRAME_PASSES], qaA[JFRAME_PASSES], a1A[JFRAME_PASSES], a2A[JFRAME_PASSES], a4A[JFRAME_PASSES];
  bool isfA[JFRAME_PASSES], ismkA[JFRAME_PASSES];
  unsigned chainA[JFRAME_PASSES];
  float q0loA[JFRAME_PASSES], massA[JFRAME_PASSES], ImodA[JFRAME_PASSES][6];
  v3 vA[JFRAME_PASSES];   // my column
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    const int slot = 64 * p + lane;
    const int fb = slot >> 2, fc = slot & 3;
    const bool isf = fb < nb;                            // frame lane: column fc of body fb
    const bool ismk = markers && fb >= nb && fb < nb + 2;   // ... of marker fb - nb
    const int b = isf ? fb : 0;
    fbA[p] = fb; fcA[p] = fc; isfA[p] = isf; ismkA[p] = ismk; bA[p] = b;
    jtA[p] = s.mc.b_jtype[b]; qaA[p] = s.mc.b_qadr[b];
    a1A[p] = s.mc.b_anc[b][0]; a2A[p] = s.mc.b_anc[b][1]; a4A[p] = s.mc.b_anc[b][2];
    chainA[p] = isf ? m->b_chainmask[b] : 0u;             // dofs that move body fb
    q0loA[p] = m->b_qpos0_lo[b];
    // Model constants of the later phases, issued now: their L2 latency overlaps the frame composition below instead of being paid
    // right before their use (nothing may cross the wave_sync fences on its own).
#pragma unroll
    for (int k = 0; k < 6; k++) ImodA[p][k] = m->b_inertia[b][k];
    massA[p] = m->b_mass[b];
    vA[p] = mk3(0.f, 0.f, 0.f);
  }
  // per geom slot
  bool isgA[JGEOM_PASSES];
  int gbA[JGEOM_PASSES], kmA[JGEOM_PASSES];
  float grbA[JGEOM_PASSES];
  v3 gp0A[JGEOM_PASSES];
  m3 gR0A[JGEOM_PASSES];
#pragma unroll
  for (int p = 0; p < JGEOM_PASSES; p++) {
    const int g = 64 * p + lane;
    isgA[p] = GEOMS && g < m->ngeom;
    const int gl = isgA[p] ? g : 0;
    gbA[p] = -1; kmA[p] = -1; grbA[p] = 0.f; gp0A[p] = mk3(0.f, 0.f, 0.f);
    if (GEOMS) {
      gbA[p] = m->g_body[gl]; kmA[p] = markers ? m->g_marker[gl] : -1;
      grbA[p] = m->g_rbound[gl];
      const bool onmk = kmA[p] >= 0 && gbA[p] < 0;
      gp0A[p] = ld3(onmk ? m->g_lpos[gl] : m->g_pos[gl]);
      gR0A[p] = ldm(onmk ? m->g_lmat[gl] : m->g_mat[gl]);
    }
  }
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    if (isfA[p]) {
      const int b = bA[p], fc = fcA[p], qa = qaA[p];
      v3 v;
      if (jtA[p] == JJ_HINGE) {
        // joint angle relative to the reference, from the compensated state: hi part q - q0 with its exact rounding error (two-sum),
        // low part = state's low part - reference's low part (ref 3.14: 1.05e-7 off its float) + that error
        const float q = s.qpos[qa], q0 = s.mc.b_qpos0[b];
        const float ah = q - q0, bb = ah - q;
        const float al = ((q - (ah - bb)) + (-q0 - bb)) + (s.qpos_lo[qa] - q0loA[p]);
        const m3 Rj = axis_rot(ld3(s.mc.b_axis[b]), ah, al);
        const v3 rc = fc == 0 ? col(Rj, 0) : (fc == 1 ? col(Rj, 1) : col(Rj, 2));
        v = fc == 3 ? ld3(s.mc.b_pos[b]) : mul(ldm(s.mc.b_mat[b]), rc);
      } else {
        float w = s.qpos[qa + 3], x = s.qpos[qa + 4], y = s.qpos[qa + 5], z = s.qpos[qa + 6];
        float n = sqrtf(w * w + x * x + y * y + z * z);
        if (n < JMINVAL) { w = 1.f; x = y = z = 0.f; } else { float in = 1.f / n; w *= in; x *= in; y *= in; z *= in; }
        const m3 R = quat2mat(w, x, y, z);
        v = fc == 0 ? col(R, 0) : (fc == 1 ? col(R, 1) : (fc == 2 ? col(R, 2) : ld3(&s.qpos[qa])));
      }
      vA[p] = v;
      st_col(TA + 4 * (64 * p + lane), v);
    }
  }
  wave_sync();
  JWSTAMP(9);
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    if (isfA[p]) {
      v3 v = vA[p];
      if (a1A[p] >= 0) { v3 c0, c1, c2, pp; ld_cols(TA + 16 * a1A[p], c0, c1, c2, pp); v = mul_cols(c0, c1, c2, v); if (fcA[p] == 3) v = pp + v; }
      vA[p] = v;
      st_col(TB + 4 * (64 * p + lane), v);
    }
  }
  wave_sync();
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    if (isfA[p]) {
      v3 v = vA[p];
      if (a2A[p] >= 0) { v3 c0, c1, c2, pp; ld_cols(TB + 16 * a2A[p], c0, c1, c2, pp); v = mul_cols(c0, c1, c2, v); if (fcA[p] == 3) v = pp + v; }
      vA[p] = v;
      st_col(TA + 4 * (64 * p + lane), v);
    }
  }
  wave_sync();
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    if (isfA[p]) {
      const int b = bA[p], fc = fcA[p];
      v3 v = vA[p];
      if (a4A[p] >= 0) { v3 c0, c1, c2, pp; ld_cols(TA + 16 * a4A[p], c0, c1, c2, pp); v = mul_cols(c0, c1, c2, v); if (fc == 3) v = pp + v; }
      st_col(TB + 4 * (64 * p + lane), v);
      if (fc == 3) st3(s.xpos[b], v);
      else { s.xmat[b][fc] = v.x; s.xmat[b][3 + fc] = v.y; s.xmat[b][6 + fc] = v.z; }
    } else if (ismkA[p]) {   // marker frames as records nb, nb + 1 (set_mocap_xyz / set_mocap_orientation between env steps): one path for the geoms below
      const float* P = s.mk + 12 * (fbA[p] - nb);
      const int fc = fcA[p];
      st_col(TB + 4 * (64 * p + lane), fc == 3 ? ld3(P) : mk3(P[3 + fc], P[6 + fc], P[9 + fc]));
    }
  }
  wave_sync();   // the frame scratch TA (= cinert / crb) is dead from here on; TB holds the world frames
  JWSTAMP(10);
  // one interval for everything that only needs the body frames: inertias, moving geom poses, S_d
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    if (isfA[p]) {
      const int b = bA[p], fc = fcA[p];
      const float mass = massA[p];
      v3 c0, c1, c2, pos;
      ld_cols(TB + 16 * b, c0, c1, c2, pos);
      const v3 c = pos + mul_cols(c0, c1, c2, ld3(s.mc.b_com[b]));
      if (fc == 3) {   // [m, m c]
        s.cinert[b][0] = mass; s.crb[b][0] = mass;
        const v3 mc_ = mk3(mass * c.x, mass * c.y, mass * c.z);
        st3(&s.cinert[b][1], mc_); st3(&s.crb[b][1], mc_);
      } else {
        // row i = fc of I_w = R I R^T (T = R I first, as a matrix product would), shifted to the world origin
        const float* I = ImodA[p];
        const v3 Ri = fc == 0 ? mk3(c0.x, c1.x, c2.x) : (fc == 1 ? mk3(c0.y, c1.y, c2.y) : mk3(c0.z, c1.z, c2.z));
        const v3 Ti = mk3(Ri.x * I[0] + Ri.y * I[3] + Ri.z * I[4], Ri.x * I[3] + Ri.y * I[1] + Ri.z * I[5], Ri.x * I[4] + Ri.y * I[5] + Ri.z * I[2]);
        const float w0 = Ti.x * c0.x + Ti.y * c1.x + Ti.z * c2.x;   // I_w[i][0]
        const float w1 = Ti.x * c0.y + Ti.y * c1.y + Ti.z * c2.y;   // I_w[i][1]
        const float w2 = Ti.x * c0.z + Ti.y * c1.z + Ti.z * c2.z;   // I_w[i][2]
        const float cc = dot(c, c);
        if (fc == 0) {
          const float d = w0 + mass * (cc - c.x * c.x), o1 = w1 - mass * c.x * c.y, o2 = w2 - mass * c.x * c.z;
          s.cinert[b][4] = d; s.cinert[b][7] = o1; s.cinert[b][8] = o2;
          s.crb[b][4] = d; s.crb[b][7] = o1; s.crb[b][8] = o2;
        } else if (fc == 1) {
          const float d = w1 + mass * (cc - c.y * c.y), o = w2 - mass * c.y * c.z;
          s.cinert[b][5] = d; s.cinert[b][9] = o;
          s.crb[b][5] = d; s.crb[b][9] = o;
        } else {
          const float d = w2 + mass * (cc - c.z * c.z);
          s.cinert[b][6] = d; s.crb[b][6] = d;
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < JGEOM_PASSES; p++) {
    if (GEOMS && isgA[p]) {   // geom poses (they share LDS with the constraint rows: every substep writes all of them)
      const int g = 64 * p + lane;
      v3 gp = gp0A[p]; m3 gR = gR0A[p];   // static: world pose from the model
      const int src = gbA[p] >= 0 ? gbA[p] : (kmA[p] >= 0 ? nb + kmA[p] : -1);   // rides on a moving body / on one of the two task-layer markers
      if (src >= 0) {
        v3 c0, c1, c2, pos;
        ld_cols(TB + 16 * src, c0, c1, c2, pos);
        gp = pos + mul_cols(c0, c1, c2, gp0A[p]);
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const v3 gc = mul_cols(c0, c1, c2, col(gR0A[p], j));
          gR.m[j] = gc.x; gR.m[3 + j] = gc.y; gR.m[6 + j] = gc.z;
        }
      }
      st3(s.gpos[g], gp);
      s.gpos[g][3] = grbA[p];
      stm(s.gmat[g], gR);
    }
  }
  if (lane < nv) {   // S_d: hinge -> world axis through the body origin; free joint -> 3 world translations, 3 body-frame rotations
    const int d = lane, bd = s.mc.d_body[d], k = d - s.mc.b_dadr[bd];
    const bool hinge = s.mc.b_jtype[bd] == JJ_HINGE, rotational = hinge || k >= 3;
    v3 al = hinge ? ld3(s.mc.b_axis[bd]) : mk3((k % 3) == 0 ? 1.f : 0.f, (k % 3) == 1 ? 1.f : 0.f, (k % 3) == 2 ? 1.f : 0.f);
    v3 c0, c1, c2, pos;
    ld_cols(TB + 16 * bd, c0, c1, c2, pos);
    sv S;
    S.a = mul_cols(c0, c1, c2, al);   // (the joint rotation leaves its own axis invariant)
    S.b = cross(pos, S.a);
    if (!rotational) { S.b = al; S.a = mk3(0.f, 0.f, 0.f); }
    stsv(s.cdof[d], S);
    stsv(Sq + 6 * d, S * s.qvel[d]);
  }
  wave_sync();
  JWSTAMP(11);
#pragma unroll
  for (int p = 0; p < JFRAME_PASSES; p++) {
    if (isfA[p] && fcA[p] < 2) {   // body velocity: sum of S_d qvel_d over the dofs that move the body (own + all ancestors'), straight from the mask
      st3(s.cvel[bA[p]] + 3 * fcA[p], chain_sum3(Sq + 3 * fcA[p], chainA[p]));
    }
  }
  wave_sync();
  JWSTAMP(12);
  if (lane < nv) {   // S_d-dot * qvel_d, S_d-dot = (velocity before dof d) x_m S_d
    int d = lane, bd = s.mc.d_body[d], pb = s.mc.b_parent[bd];
    sv vb; vb.a = vb.b = mk3(0, 0, 0);
    if (pb >= 0) vb = ldsv(s.cvel[pb]);
    sv r; r.a = r.b = mk3(0, 0, 0);
    int k = d - s.mc.b_dadr[bd];
    if (s.mc.b_jtype[bd] == JJ_HINGE) r = cross_motion(vb, ldsv(s.cdof[d])) * s.qvel[d];
    else if (k >= 3) {   // free joint: translations have no S-dot; all three rotations see parent + translational velocity
      int d0 = s.mc.b_dadr[bd];
      sv vt = vb + ldsv(Sq + 6 * d0) + ldsv(Sq + 6 * (d0 + 1)) + ldsv(Sq + 6 * (d0 + 2));
      r = cross_motion(vt, ldsv(s.cdof[d])) * s.qvel[d];
    }
    stsv(Sq2 + 6 * d, r);
  }
  wave_sync();
  JWSTAMP(13);
  if (isb) {   // bias acceleration: -gravity + sum of S_d-dot qvel_d over the same dofs; then the body's RNE force right away
    sv a; a.a = mk3(0, 0, 0);
    a.b = mk3(-m->gravity[0], -m->gravity[1], -m->gravity[2]);
    a = a + chain_sum(Sq2, chainb);
    const sv vv = ldsv(s.cvel[lane]);
    const sv f = inert_mul(s.cinert[lane], a) + cross_force(vv, inert_mul(s.cinert[lane], vv));
    stsv(s.cfrc[lane], f);
    stsv(s.cacc[lane], f);   // subtree force sum, completed for bodies with children by stage_accumulate
  }
}

// ---------------------------------------------------------------- stage M helpers
// Subtree sums for the bodies that have children (leaves keep their own values from stage G): composite inertias
// crb[a] = sum of cinert over a's subtree (lane = (a, component), 10 components) and RNE forces cacc[a] = sum of cfrc over
// the subtree (6 components; cacc is dead as an acceleration by now and doubles as the summed force).  Sources and
// destinations are different arrays, so there is no ordering between lanes.
// sum of T[stride * x] over the bodies x of a subtree mask (at most JMAXDESC: the arm's links and fingers), loads issued together
JDEV float subtree_sum(const float* T, int stride, unsigned mask) {
  float t[JMAXDESC];
  bool on[JMAXDESC];
#pragma unroll
  for (int k = 0; k < JMAXDESC; k++) {
    on[k] = mask != 0u;
    t[k] = T[stride * (on[k] ? __builtin_ctz(mask) : 0)];
    mask &= mask - 1u;
  }
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < JMAXDESC; k++) acc += on[k] ? t[k] : 0.f;
  return acc;
}
template <class L>
JDEV void stage_accumulate(const JacoModelDev* m, L& s, int lane) {
  const int ni = m->ninner;
#pragma unroll
  for (int base = 0; base < 10 * JMAXINNER; base += 64) {   // (one pass for the default layout: 6 inner bodies x 10 components)
    const int i = base + lane;
    if (i < 10 * ni) {
      const int a = s.mc.inner_body[i / 10], c = i % 10;
      s.crb[a][c] = subtree_sum(&s.cinert[0][0] + c, 10, s.mc.b_descmask[a]);
    }
  }
#pragma unroll
  for (int base = 0; base < 6 * JMAXINNER; base += 64) {   // (one pass in the default layout: 6 x 6 <= 64)
    const int i = base + lane;
    if (i < 6 * ni) {
      const int a = s.mc.inner_body[i / 6], c = i % 6;
      s.cacc[a][c] = subtree_sum(&s.cfrc[0][0] + c, 6, s.mc.b_descmask[a]);
    }
  }
}

// Per-lane model constants of the stages after the tree walk, fetched in one go right after it: their L2 round trip is hidden
// behind the subtree sums and the mass matrix instead of being paid, one dependent load at a time, in front of every stage.
struct StagePrefetch {
  int mp[JMAXMPAIR / 64];     // stage M: this lane's mass-matrix entries
  float damping;              // joint damping of dof `lane`
  float stiffness, springref; int sqadr;   // joint spring of dof `lane` (models that have any)
  int limited; float lo, hi;  // joint limit of body `lane`
};
JDEV StagePrefetch stage_prefetch(const JacoModelDev* m, int lane) {
  StagePrefetch P;
#pragma unroll
  for (int h = 0; h < JMAXMPAIR / 64; h++) P.mp[h] = m->mpair[64 * h + lane];             // (zero-padded to JMAXMPAIR)
  P.damping = m->d_damping[lane < JNV ? lane : 0];
  P.stiffness = 0.f; P.springref = 0.f; P.sqadr = 0;
  if (m->has_springs) { P.stiffness = m->d_stiffness[lane < JNV ? lane : 0]; P.springref = m->d_springref[lane < JNV ? lane : 0]; P.sqadr = m->d_qadr[lane < JNV ? lane : 0]; }
  const int b = lane < JNB ? lane : 0;
  P.limited = m->b_limited[b]; P.lo = m->b_range[b][0]; P.hi = m->b_range[b][1];
  return P;
}
// ---------------------------------------------------------------- stage M: mass matrix, bias, actuation
// lane = dof: F_d = Ic(body d) S_d and the bias force; then lane = structurally non-zero entry (d, j), j an ancestor-or-self
// dof of d: M[d][j] = S_j . F_d.  Scratch: the not-yet-built constraint-row area s.J.
template <class L>
JDEV void stage_mass_bias(const JacoModelDev* m, L& s, int lane, const StagePrefetch& pf) {
  float* Fd = s.J;   // [JNV][6]
  const int nmp = m->nmpair;
  if (lane < m->nv) {
    int d = lane, b = s.mc.d_body[d];
    sv S = ldsv(s.cdof[d]);
    stsv(Fd + 6 * d, inert_mul(s.crb[b], S));
    float bias = dot(S, ldsv(s.cacc[b]));
    s.bias[d] = bias;
    float passive = -pf.damping * s.qvel[d];
    if (m->has_springs && pf.sqadr >= 0) passive -= pf.stiffness * (s.qpos[pf.sqadr] - pf.springref);   // (wave-uniform test: one model per launch)
    s.smooth[d] = passive - bias;
  }
  wave_sync();
#pragma unroll
  for (int h = 0; h < JMAXMPAIR / 64; h++) {
    int code = pf.mp[h];
    if (h * 64 + lane < nmp) {
      int d = code & 255, j = code >> 8;
      float v = dot(ldsv(s.cdof[j]), ldsv(Fd + 6 * d));
      s.M[m_index(d, j)] = v;
      s.M[m_index(j, d)] = v;
    }
  }
}
// The actuator's model constants are fetched ahead of their use (act_fetch before the subtree sums, stage_actuation after the mass
// matrix): one L2 round trip hidden behind two stages instead of a dependent load chain right before the torque is applied.
struct ActParams { int ctrllimited, position, forcelimited, qadr, dof; float c0, c1, kp, f0, f1; };
JDEV ActParams act_fetch(const JacoModelDev* m, int lane) {
  const int a = lane < m->nu ? lane : 0;
  ActParams P;
  P.ctrllimited = m->a_ctrllimited[a]; P.position = m->a_position[a]; P.forcelimited = m->a_forcelimited[a]; P.qadr = m->a_qadr[a]; P.dof = m->a_dof[a];
  P.c0 = m->a_ctrlrange[a][0]; P.c1 = m->a_ctrlrange[a][1]; P.kp = m->a_kp[a]; P.f0 = m->a_forcerange[a][0]; P.f1 = m->a_forcerange[a][1];
  return P;
}
template <class L>
JDEV void stage_actuation(const JacoModelDev* m, L& s, int lane, const ActParams& P) {
  if (lane < m->nu) {
    float c = s.ctrl[lane];
    if (P.ctrllimited) c = fmaxf(P.c0, fminf(P.c1, c));
    float f = P.position ? P.kp * (c - s.qpos[P.qadr]) : c;
    if (P.forcelimited) f = fmaxf(P.f0, fminf(P.f1, f));
    s.smooth[P.dof] += f;
  }
}

// ---------------------------------------------------------------- stage R: constraint rows
// impedance d(r) and reference acceleration of MuJoCo's soft constraints
JDEV float impedance(const float* solimp, float pos) {
  float dmin = fminf(0.9999f, fmaxf(0.0001f, solimp[0])), dmax = fminf(0.9999f, fmaxf(0.0001f, solimp[1]));
  if (dmin == dmax) return dmax;   // flat impedance curve (every contact pair of this model that involves the object or a finger)
  float width = fmaxf(JMINVAL, solimp[2]), mid = fminf(0.9999f, fmaxf(0.0001f, solimp[3])), power = fmaxf(1.f, solimp[4]);
  float x = fabsf(pos) / width, y;
  if (x >= 1.f) return dmax;
  if (power == 1.f) y = x;
  else if (power == 2.f) y = x <= mid ? x * x / mid : 1.f - (1.f - x) * (1.f - x) / (1.f - mid);
  else if (x <= mid) y = powf(x, power) / powf(mid, power - 1.f);
  else y = 1.f - powf(1.f - x, power) / powf(1.f - mid, power - 1.f);
  return dmin + y * (dmax - dmin);
}
// returns aref; *Rout = regulariser R for the given diagApprox; *imp_out = impedance (reused by the caller)
JDEV float row_params(const float* solref, const float* solimp, float pos, float vel, float diagApprox, float* Rout, float* imp_out = nullptr) {
  float imp = impedance(solimp, pos);
  float dmax = fminf(0.9999f, fmaxf(0.0001f, solimp[1]));
  float K = 1.f / fmaxf(JMINVAL, dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1]);
  float B = 2.f / fmaxf(JMINVAL, dmax * solref[0]);
  *Rout = fmaxf(JMINVAL, (1.f - imp) * diagApprox / imp);
  if (imp_out) *imp_out = imp;
  return -B * vel - K * imp * pos;
}

// joint limits: lane = body; one row per violated limit, compacted with a ballot
template <class L>
JDEV void stage_limit_rows(const JacoModelDev* m, L& s, int lane, const StagePrefetch& pf) {
  bool act = false;
  float dist = 0.f, sgn = 1.f;
  int d = 0;
  if (lane < m->nbody && s.mc.b_jtype[lane] == JJ_HINGE && pf.limited) {
    float q = s.qpos[s.mc.b_qadr[lane]];
    float lo = q - pf.lo, hi = pf.hi - q;
    d = s.mc.b_dadr[lane];
    if (lo < 0.f) { act = true; dist = lo; sgn = 1.f; }
    else if (hi < 0.f) { act = true; dist = hi; sgn = -1.f; }
  }
  unsigned long long mask = wave_ballot(act);
  int r = wave_prefix_count(mask);
  if (act) {
    if (L::Caps::WRENCH) {
      for (int k = 0; k < 8; k++) s.J[r * 8 + k] = 0.f;
      s.J[r * 8] = sgn;
    } else {
      for (int k = 0; k < JNV; k++) s.J[r * JLD + k] = 0.f;
      s.J[r * JLD + d] = sgn;
    }
    float R;
    s.e_aref[r] = row_params(m->b_solref[lane], m->b_solimp[lane], dist, sgn * s.qvel[d], m->d_invweight[d], &R);
    s.e_D[r] = 1.f / R;
    s.e_con[r] = ((d < JB0 ? 1 : (d < JB1 ? 2 : 4)) << 16) | (L::Caps::WRENCH ? (JW_UNIT | (d << 19)) : 0);
  }
  if (lane == 0) { s.nefc = popc64(mask); s.nlimit = s.nefc; }
}

// ---------------------------------------------------------------- stage S: primal Newton solver
// Lane k < nv owns element k of every dof vector and row k of M / H; lane r owns constraint rows r + 64 q.
struct NewtonOut { float qacc, qfrc_con, qdamped; int iters; bool have_qdamped; };
#define JDAMPED_BLOCKS 1   // implicit joint damping only exists on the finger dofs (block 0); checked by the host loader

// out[q] = sum_k J[row(q)][k] * v[k] for the lane's NR rows; v distributed one element per lane.
// NR == 1 keeps the lane's J row in registers (jrow) for the whole solve: no LDS traffic here.
template <int NR>
JDEV void rows_dot(const float* J, const float (&jrow)[JNV], float vk, int lane, int ne, int nv, float (&out)[NR]) {
  // (the broadcasts first, then the products: a v_readlane straight in front of the VALU that reads its SGPR costs wait states)
  float vb[JNV];
#pragma unroll
  for (int k = 0; k < JNV; k++) vb[k] = wave_bcast(vk, k);
  if (NR == 1) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < JNV; k++) acc += jrow[k] * vb[k];
    out[0] = acc;
    return;
  }
  const float* Jr[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) { int r = lane + 64 * q; Jr[q] = J + (r < ne ? r : 0) * JLD; out[q] = 0.f; }
#pragma unroll
  for (int k = 0; k < JNV; k++) {
#pragma unroll
    for (int q = 0; q < NR; q++) out[q] += Jr[q][k] * vb[k];
  }
  (void)nv;
}
JDEV float mat_vec(const float (&mrow)[JNV], float vk) {  // (M v)[lane], mrow = M[lane][:] (zero for lanes >= nv)
  float vb[JNV];
#pragma unroll
  for (int k = 0; k < JNV; k++) vb[k] = wave_bcast(vk, k);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < JNV; k++) acc += mrow[k] * vb[k];
  return acc;
}
// sum_r J[r][lane] * f_r, f distributed over the row slots; 4 rows in flight per step to cover LDS latency
template <int NR>
JDEV float jt_vec(const float* J, const float (&f)[NR], int ne, int lane, int nv) {
  float acc = 0.f;
  int kk = lane < nv ? lane : 0;
#pragma unroll
  for (int q = 0; q < NR; q++) {
    int n = ne - 64 * q;
    n = n > 64 ? 64 : n;
    const float* Jq = J + (64 * q) * JLD + kk;
    int rl = 0;
    for (; rl + 4 <= n; rl += 4) {
      float j0 = Jq[(rl + 0) * JLD], j1 = Jq[(rl + 1) * JLD], j2 = Jq[(rl + 2) * JLD], j3 = Jq[(rl + 3) * JLD];
      const float f0 = wave_bcast(f[q], rl), f1 = wave_bcast(f[q], rl + 1), f2 = wave_bcast(f[q], rl + 2), f3 = wave_bcast(f[q], rl + 3);
      acc += j0 * f0 + j1 * f1 + j2 * f2 + j3 * f3;
    }
    for (; rl < n; rl++) acc += Jq[rl * JLD] * wave_bcast(f[q], rl);
  }
  return lane < nv ? acc : 0.f;
}

// The stopping test one iteration ahead.  When the step just taken switched no row on or off, the cost along it was ONE quadratic, of which
// the search direction was the exact Newton step: the gradient left over is (1 - al) times the old one and the iteration that would follow would
// improve the cost by exactly (1 - al)^2 times what this one did.  MuJoCo's own criterion (improvement * scale < tolerance, [EXT] engine_solver.c)
// applied to that prediction ends the solve here -- without the matrix-core pass, Hessian fetch and gradient that the next round would spend on
// finding out the same thing (half of all Hessian builds: most solves are one real Newton step).
#ifndef JACO_NEWTON_LOOKAHEAD
#define JACO_NEWTON_LOOKAHEAD 1   // (0: A/B builds, tools/build_variant.sh)
#endif
JDEV bool newton_next_round_is_idle(bool any_row_switched, float al, float improvement, float scale, float tol) {
  const float r = 1.f - al;
  return JACO_NEWTON_LOOKAHEAD && !any_row_switched && r * r * improvement * scale < tol;
}
template <class L>
JDEV NewtonOut stage_newton(const JacoModelDev* m, L& s, const float (&mrow)[JNV], float smooth, float hd, int lane, JProfCtx& pc) {
  (void)pc;
  constexpr int NR = L::Caps::NR, MAXEFC = L::Caps::MAXEFC;
  NewtonOut out;
  int nv = m->nv, ne = wave_uniform_i(s.nefc);
  out.qfrc_con = 0.f; out.iters = 0; out.qdamped = 0.f; out.have_qdamped = false;
  float h0[JNV];
#pragma unroll
  for (int j = 0; j < JNV; j++) h0[j] = lane < nv ? mrow[j] : (lane == j ? 1.f : 0.f);
  // dof blocks the model really has (an arm-only model in the default layout: block 0 alone; the others would be identity rows)
  const int blockmask = 1 | (nv > JB0 ? 2 : 0) | (nv > JB1 ? 4 : 0);
  if (ne == 0) {   // unconstrained: qacc = M^-1 qfrc_smooth
    if (m->has_damping == 1) {   // ... and the Euler step's (M + h D)^-1 qfrc_smooth out of the same elimination
      out.qacc = ldl_block0_dual(h0, smooth, hd, lane, &out.qdamped) + ldl_solve_blocks(h0, smooth, lane, blockmask & 6);
      out.have_qdamped = true;
    } else out.qacc = ldl_solve_blocks(h0, smooth, lane, blockmask);
    return out;
  }
  if (ne > MAXEFC) ne = MAXEFC;
  bool valid[NR];
  float D[NR], ar[NR], x[NR], f[NR], jp[NR];
  int blk[NR];
  float jrow[JNV];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    int r = lane + 64 * q;
    valid[q] = r < ne;
    D[q] = valid[q] ? s.e_D[r] : 0.f;
    ar[q] = valid[q] ? s.e_aref[r] : 0.f;
    blk[q] = valid[q] ? (s.e_con[r] >> 16) & 7 : 0;
  }
#pragma unroll
  for (int k = 0; k < JNV; k++) jrow[k] = (NR == 1 && lane < ne) ? s.J[lane * JLD + k] : 0.f;
  float scale = 1.f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  float tol = m->tolerance;
  // dof blocks that carry any constraint row; for the others the optimum is exactly M^-1 qfrc_smooth and stays there
  int rowblk = 0;
#pragma unroll
  for (int q = 0; q < NR; q++) rowblk |= blk[q];
  const int rowblocks = (wave_ballot(rowblk & 1) ? 1 : 0) | (wave_ballot(rowblk & 2) ? 2 : 0) | (wave_ballot(rowblk & 4) ? 4 : 0);
  const bool mine = lane < nv && ((lane < JB0 ? 1 : (lane < JB1 ? 2 : 4)) & rowblocks) != 0;
  float afree;
  if ((rowblocks & 1) == 0 && m->has_damping == 1) {   // (1: damping on the finger joints only, dofs >= JLDL_NSH)
    // the arm/finger block carries no row: its acceleration is M^-1 qfrc_smooth, and the Euler step's implicitly damped
    // (M + h D)^-1 qfrc_smooth comes out of the same elimination (qfrc_constraint is zero on these dofs)
    afree = ldl_block0_dual(h0, smooth, hd, lane, &out.qdamped);
    out.have_qdamped = true;
    afree += ldl_solve_blocks(h0, smooth, lane, ~rowblocks & 6 & blockmask);
  } else afree = ldl_solve_blocks(h0, smooth, lane, ~rowblocks & 7 & blockmask);
  // Primal problem in terms of qfrc_smooth (same optimum as MuJoCo's (a - a_s)' M (a - a_s) form, no M^-1 needed for
  // the blocks that carry rows):  minimise 1/2 a'Ma - a'qfrc_smooth + sum_i 1/2 D_i min(0, J_i a - aref_i)^2.
  // Start: warm start on the row-carrying blocks (MuJoCo additionally compares it with the unconstrained point; with
  // an exact line search either start reaches the same optimum), exact solution elsewhere.
  float a = mine ? s.qacc_ws[lane] : afree;
  float Ma = mat_vec(mrow, a) - smooth;   // from here on "Ma" = gradient of the smooth part, M a - qfrc_smooth
  Ma = mine ? Ma : 0.f;
  rows_dot<NR>(s.J, jrow, a, lane, ne, nv, x);
#pragma unroll
  for (int q = 0; q < NR; q++) x[q] = valid[q] ? x[q] - ar[q] : 0.f;
  JSTAMP_NEWTON(11);
#ifdef JACO_EMULATED
  if (lane == 0) emu_counter[8]++;   // (CPU tests: constrained solves, and -- below -- the Hessian builds they took)
#endif
  int it = 0, nls = 0;
  for (; it < m->iterations; it++) {
    // (lane-id predicates are recomputed inside the iteration: hoisted out of it they live as ~60 SGPR masks, spilled to VGPR lanes
    //  on entry and fetched back with two v_readlane per use -- a v_cmp at the use is one instruction)
    lane = wave_opaque_i(lane);
#ifdef JACO_EMULATED
    if (lane == 0) emu_counter[9]++;
#endif
    bool coupled = false;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      f[q] = x[q] < 0.f ? -D[q] * x[q] : 0.f;
      coupled = coupled || (x[q] < 0.f && (blk[q] & (blk[q] - 1)) != 0);   // active row touching two dof blocks
    }
    bool full = wave_ballot(coupled) != 0ull;
    // One matrix-core pass builds both the Hessian term and J^T f:  C = Jh^T diag(D*active) Jh  with Jh = [J | -x]
    // (32 columns: 21 dofs, column 21 = -x, rest zero), two constraint rows per v_mfma_f32_32x32x2_f32.
    // C[0..20][0..20] = sum_active D_r J_r^T J_r,  C[21][0..20] = J^T f  (f_r = -D_r x_r on active rows).
    // Per-row weights and residuals are staged in LDS so that every operand of a step is an independent LDS read
    // (4 steps in flight): e_f is only written at the very end; `smooth` (already in registers) is dead by now.
    float* xs = NR == 1 ? s.smooth : s.e_x;
#pragma unroll
    for (int q = 0; q < NR; q++) if (valid[q]) { s.e_f[lane + 64 * q] = x[q] < 0.f ? D[q] : 0.f; xs[lane + 64 * q] = x[q]; }
    wave_sync();
    float h[JNV], jtf;
    if ((rowblocks & 1) == 0) {
      // rows only on the free bodies (dofs [JB0, JNV), 12 of them): the tile is 16 x 16 -- 12 dof columns + the -x column --
      // and v_mfma_f32_16x16x4_f32 takes four constraint rows per issue (a quarter of the matrix-core time of the 32 x 32 form)
      acc16x16 C;
      acc_zero(C);
      const int uq = lane >> 4, col = lane & 15;
      // (rows beyond the last one read the last row with weight 0; the column masks are exact multipliers: no selects in the loop)
      const float cj = col < JNV - JB0 ? 1.f : 0.f, cx = col == JNV - JB0 ? -1.f : 0.f;
      const float* jcol = s.J + JB0 + (col < JNV - JB0 ? col : 0);
      const int last = ne - 1;
      // (fetching the next round's operands ahead of this round's matrix-core issues was tried in round 5: neutral, profiles/r05_ab_variants.txt)
      for (int r0 = 0; r0 < ne; r0 += 16) {
        float jv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int r = r0 + 4 * u + uq, rc = r < last ? r : last;
          jv[u] = cj * jcol[rc * JLD] + cx * xs[rc];
          wv[u] = r < ne ? s.e_f[rc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) wave_mfma_16x16x4(jv[u], wv[u] * jv[u], C);
      }
      JSTAMP_NEWTON(12);
      // symmetric tile: H[jd][c] = C[c][jd] sits in lane jd + 16 * (c >> 2), register c & 3 (jd = this lane's dof - JB0)
      const int jd = (lane >= JB0 && lane < JNV) ? lane - JB0 : 0;
#pragma unroll
      for (int j2 = 0; j2 < JNV; j2++) {
        float cv = 0.f;
        if (j2 >= JB0) { const int c = j2 - JB0; cv = wave_shfl(C.v[c & 3], jd + 16 * (c >> 2)); }
        h[j2] = lane < nv ? mrow[j2] + (lane >= JB0 ? cv : 0.f) : (lane == j2 ? 1.f : 0.f);
      }
      jtf = wave_shfl(C.v[(JNV - JB0) & 3], jd + 16 * ((JNV - JB0) >> 2));   // row 12: J^T f
      jtf = lane >= JB0 ? jtf : 0.f;
    } else {
    acc32x32 C;
    acc_zero(C);
    const int uh = lane >> 5, col = lane & 31;
    const float cj = col < JNV ? 1.f : 0.f, cx = col == JNV ? -1.f : 0.f;
    const float* jcol = s.J + (col < JNV ? col : 0);
    const int last = ne - 1;
    for (int r0 = 0; r0 < ne; r0 += 8) {
      float jv[4], wv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int r = r0 + 2 * u + uh, rc = r < last ? r : last;
        jv[u] = cj * jcol[rc * JLD] + cx * xs[rc];
        wv[u] = r < ne ? s.e_f[rc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) wave_mfma_32x32x2(jv[u], wv[u] * jv[u], C);
    }
    JSTAMP_NEWTON(12);
    // C is symmetric: row j of the tile, column `lane`, is where lane `lane` finds H[lane][j]
#pragma unroll
    for (int j2 = 0; j2 < JNV; j2++) {
      const int reg = (j2 & 3) + 4 * (j2 >> 3);
      float cv = ((j2 >> 2) & 1) ? wave_shfl(C.v[reg], (lane + 32) & 63) : C.v[reg];
      h[j2] = lane < nv ? mrow[j2] + cv : (lane == j2 ? 1.f : 0.f);
    }
    // row JNV = J^T f (default layout: row 21, which sits in the upper lane half)
    jtf = ((JNV >> 2) & 1) ? wave_shfl(C.v[(JNV & 3) + 4 * (JNV >> 3)], (lane + 32) & 63) : C.v[(JNV & 3) + 4 * (JNV >> 3)];
    }
    float grad = Ma - (lane < nv ? jtf : 0.f);
    float gn = sqrtf(wave_sum(grad * grad));
    if (gn * scale < tol) break;
    // coupled rows: one factorisation over the coupled dofs -- the two free bodies alone ([JB0, JNV), e.g. object resting on
    // the pedestal) when the arm block carries no row, else all 21
    float p = !full ? ldl_solve_blocks(h, -grad, lane, rowblocks) : ((rowblocks & 1) == 0 ? ldl_block<JB0, JNV>(h, -grad, lane) : ldl_solve<true>(h, -grad, lane));
    p = lane < nv ? p : 0.f;
    JSTAMP_NEWTON(13);
    // exact line search on phi(al) = cost(a + al p)
    float Mp = mat_vec(mrow, p);
    float pMp, pMa;
    wave_sum2(p * Mp, p * Ma, &pMp, &pMa);
    rows_dot<NR>(s.J, jrow, p, lane, ne, nv, jp);
#pragma unroll
    for (int q = 0; q < NR; q++) jp[q] = valid[q] ? jp[q] : 0.f;
    float al = 0.f, lo = 0.f, hi = 3.0e38f, d10 = 0.f, dlo = 0.f, dhi = 0.f;
    for (int ls = 0; ls < m->ls_iterations; ls++) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < NR; q++) {
        float xa = x[q] + al * jp[q];
        s1 += xa < 0.f ? D[q] * xa * jp[q] : 0.f;
        s2 += xa < 0.f ? D[q] * jp[q] * jp[q] : 0.f;
      }
      float w1, w2;
      wave_sum2(s1, s2, &w1, &w2);
      float d1 = pMa + al * pMp + w1, d2 = pMp + w2;
      if (ls == 0) d10 = fabsf(d1);
#ifdef JACO_TRACE_LS
      if (lane == 0) printf("it %d ls %d al %.9g d1 %.6g d2 %.6g lo %.9g hi %.9g pMa %.6g pMp %.6g\n", it, ls, al, d1, d2, lo, hi, pMa, pMp);
#endif
      if (ls > 0 && fabsf(d1) <= 1e-6f * d10) break;
      if (d1 < 0.f) { lo = al; dlo = -d1; } else { hi = al; dhi = d1; }
      // The minimiser of the piecewise-quadratic often sits on a kink (a row switching on/off), where no point satisfies the
      // derivative test and bisection would run down to the float spacing (20-40 rounds in a quarter of all substeps).
      // Inside a bracket the cost varies by at most max|phi'| x width: once that is far below the solver's own
      // tolerance on cost improvements, any point of the bracket is as good as the exact minimiser.
      if (hi < 1.0e38f && fmaxf(dlo, dhi) * (hi - lo) * scale < 1e-3f * tol) break;
      const float stepn = d1 / d2;
      float nx = al - stepn;
      if (!(nx > lo && nx < hi)) nx = hi < 1.0e38f ? 0.5f * (lo + hi) : 2.f * al + 1.f;
      if (nx == al) break;
      al = nx;
      nls++;
    }
    // move; the cost decrease is evaluated along the line in cancellation-free form so that MuJoCo's absolute
    // tolerance stays meaningful in fp32 (the 1280 kg pedestal's cost terms are ~1e3, the object's ~1e-3)
    float dc = 0.f;
    bool switched = false;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      float dx = al * jp[q], xn = x[q] + dx;
      bool was = x[q] < 0.f, is = xn < 0.f;
      switched = switched || was != is;
      dc += (was && is) ? 0.5f * D[q] * dx * (2.f * x[q] + dx) : (is ? 0.5f * D[q] * xn * xn : (was ? -0.5f * D[q] * x[q] * x[q] : 0.f));
      x[q] = xn;
    }
    const bool any_switched = wave_ballot(switched) != 0ull;
    float improvement = -(al * pMa + 0.5f * al * al * pMp + wave_sum(dc));
    JSTAMP_NEWTON(14);
    a += al * p; Ma += al * Mp;
    if (improvement * scale < tol || newton_next_round_is_idle(any_switched, al, improvement, scale, tol)) { it++; break; }
  }
#pragma unroll
  for (int q = 0; q < NR; q++) f[q] = x[q] < 0.f ? -D[q] * x[q] : 0.f;
  out.qacc = a;
  out.qfrc_con = jt_vec<NR>(s.J, f, ne, lane, nv);
  out.iters = it | (nls << 8);
#pragma unroll
  for (int q = 0; q < NR; q++) if (valid[q]) s.e_f[lane + 64 * q] = f[q];
  return out;
}

// ---------------------------------------------------------------- stage S on body-space rows (JacoCaps::WRENCH)
// A lane's dof column: the motion subspace S_col (angular, linear at the world origin) and the bodies the dof moves (bit b + 1; bit 0 = static, never set).
struct WCol { sv S; unsigned mv; };
template <class L>
JDEV WCol wcol_load(const L& s, int col, int nv) {
  WCol c;
  const bool ok = col < nv;
  c.S = ldsv(s.cdof[ok ? col : 0]);
  if (!ok) { c.S.a = mk3(0.f, 0.f, 0.f); c.S.b = mk3(0.f, 0.f, 0.f); }
  c.mv = ok ? s.mc.b_descmask[s.mc.d_body[col]] << 1 : 0u;
  return c;
}
// J[r][col] of a row record (a = w[0..3], b = w[4], w[5], weight, residual) with e_con word `con`
JDEV float wrow_entry(const WCol& c, int col, const v4& a, const v4& b, int con) {
  const float sgn = (float)((int)((c.mv >> ((con >> 24) & 31)) & 1u) - (int)((c.mv >> ((con >> 19) & 31)) & 1u));
  const float body = sgn * (a.x * c.S.a.x + a.y * c.S.a.y + a.z * c.S.a.z + a.w * c.S.b.x + b.x * c.S.b.y + b.y * c.S.b.z);
  return (con & JW_UNIT) ? (((con >> 19) & 31) == col ? a.x : 0.f) : body;
}
// out[q] = (J v)[row(q)]: v (one element per dof lane) -> one 6-vector per CONTACT, D_c = sum of S_k v_k over the dofs that move body B but not
// body A minus the same over those that move A but not B (lane = (contact, component); dofs common to both chains are left out exactly, as in a
// dense row -- formed per body and subtracted, two arm bodies' common prefix would cost the difference its last bits) -> w . D_c per row.
// D_c is staged in the contact frame's tangent slots (dead since the row builder); a joint-limit row reads v[dof].  `vs`: v in LDS (s.smooth + 32).
template <int NR, class L>
JDEV void rows_dot_w(L& s, const float (&w)[NR][6], const int (&con)[NR], float vk, int lane, int nv, int ncon, float (&out)[NR]) {
  float* vs = s.smooth + 32;
  wave_sync();   // the previous call's readers of vs / the staged vectors are done
  if (lane < nv) vs[lane] = vk;
  wave_sync();
  for (int idx = lane; idx < 6 * ncon; idx += 64) {
    const int c = idx / 6, comp = idx - 6 * c, obs = s.c_ob[c];
    const int B1 = (obs >> 8) & 31, B2 = (obs >> 24) & 31;
    const unsigned m1 = B1 ? s.mc.b_chain[B1 - 1] : 0u, m2 = B2 ? s.mc.b_chain[B2 - 1] : 0u;
    const unsigned plus = m2 & ~m1;
    unsigned mask = m1 ^ m2;
    float acc = 0.f;
    while (mask) {
      const int k = __builtin_ctz(mask);
      mask &= mask - 1u;
      const float t = s.cdof[k][comp] * vs[k];
      acc += ((plus >> k) & 1u) ? t : -t;
    }
    s.c_frame[c][3 + comp] = acc;
  }
  wave_sync();
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const bool unit = (con[q] & JW_UNIT) != 0;
    const float* Dc = s.c_frame[unit ? 0 : (con[q] & 255)] + 3;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 6; c++) acc += w[q][c] * Dc[c];
    out[q] = unit ? w[q][0] * vs[(con[q] >> 19) & 31] : acc;
  }
}

// The same primal Newton iteration as stage_newton (start point, tolerances, block logic, exact line search), on body-space rows.
template <class L>
JDEV NewtonOut stage_newton_w(const JacoModelDev* m, L& s, const float (&mrow)[JNV], float smooth, float hd, int lane, JProfCtx& pc) {
  (void)pc;
  constexpr int NR = L::Caps::NR, MAXEFC = L::Caps::MAXEFC;
  NewtonOut out;
  const int nv = m->nv, ncon = wave_uniform_i(s.ncon), nlim = wave_uniform_i(s.nlimit);
  int ne = wave_uniform_i(s.nefc);
  out.qfrc_con = 0.f; out.iters = 0; out.qdamped = 0.f; out.have_qdamped = false;
  float h0[JNV];
#pragma unroll
  for (int j = 0; j < JNV; j++) h0[j] = lane < nv ? mrow[j] : (lane == j ? 1.f : 0.f);
  const int blockmask = 1 | (nv > JB0 ? 2 : 0) | (nv > JB1 ? 4 : 0);
  if (ne == 0) {   // unconstrained: qacc = M^-1 qfrc_smooth
    if (m->has_damping == 1) {
      out.qacc = ldl_block0_dual(h0, smooth, hd, lane, &out.qdamped) + ldl_solve_blocks(h0, smooth, lane, blockmask & 6);
      out.have_qdamped = true;
    } else out.qacc = ldl_solve_blocks(h0, smooth, lane, blockmask);
    return out;
  }
  if (ne > MAXEFC) ne = MAXEFC;
  bool valid[NR];
  float D[NR], ar[NR], x[NR], f[NR], jp[NR], w[NR][6];
  int con[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
    const int r = lane + 64 * q;
    valid[q] = r < ne;
    const int rc = valid[q] ? r : 0;
    D[q] = valid[q] ? s.e_D[rc] : 0.f;
    ar[q] = valid[q] ? s.e_aref[rc] : 0.f;
    con[q] = valid[q] ? s.e_con[rc] : 0;
    const v4 a = ld4(s.J + 8 * rc), b = ld4(s.J + 8 * rc + 4);
    w[q][0] = valid[q] ? a.x : 0.f; w[q][1] = valid[q] ? a.y : 0.f; w[q][2] = valid[q] ? a.z : 0.f;
    w[q][3] = valid[q] ? a.w : 0.f; w[q][4] = valid[q] ? b.x : 0.f; w[q][5] = valid[q] ? b.y : 0.f;
  }
  const float scale = 1.f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  const float tol = m->tolerance;
  int rowblk = 0;
#pragma unroll
  for (int q = 0; q < NR; q++) rowblk |= (con[q] >> 16) & 7;
  const int rowblocks = (wave_ballot(rowblk & 1) ? 1 : 0) | (wave_ballot(rowblk & 2) ? 2 : 0) | (wave_ballot(rowblk & 4) ? 4 : 0);
  const bool mine = lane < nv && ((lane < JB0 ? 1 : (lane < JB1 ? 2 : 4)) & rowblocks) != 0;
  float afree;
  if ((rowblocks & 1) == 0 && m->has_damping == 1) {
    afree = ldl_block0_dual(h0, smooth, hd, lane, &out.qdamped);
    out.have_qdamped = true;
    afree += ldl_solve_blocks(h0, smooth, lane, ~rowblocks & 6 & blockmask);
  } else afree = ldl_solve_blocks(h0, smooth, lane, ~rowblocks & 7 & blockmask);
  float a = mine ? s.qacc_ws[lane] : afree;
  float Ma = mat_vec(mrow, a) - smooth;
  Ma = mine ? Ma : 0.f;
  rows_dot_w<NR>(s, w, con, a, lane, nv, ncon, x);
#pragma unroll
  for (int q = 0; q < NR; q++) x[q] = valid[q] ? x[q] - ar[q] : 0.f;
  JSTAMP_NEWTON(11);
  // the dof columns of the matrix-core pass: lane -> column lane & 31 of the 32 x 32 tile, or dof JB0 + (lane & 15) of the 16 x 16 one
  const bool small_tile = (rowblocks & 1) == 0;
  const int col = small_tile ? (lane & 15) : (lane & 31);
  const int cdof_ = small_tile ? (col < JNV - JB0 ? JB0 + col : JNV) : col;
  const WCol cc = wcol_load(s, cdof_ < JNV ? cdof_ : nv, nv);
  const float cx = (small_tile ? col == JNV - JB0 : col == JNV) ? -1.f : 0.f;
  int it = 0, nls = 0;
  for (; it < m->iterations; it++) {
    lane = wave_opaque_i(lane);
    bool coupled = false;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      f[q] = x[q] < 0.f ? -D[q] * x[q] : 0.f;
      const int bk = (con[q] >> 16) & 7;
      coupled = coupled || (x[q] < 0.f && (bk & (bk - 1)) != 0);
    }
    const bool full = wave_ballot(coupled) != 0ull;
#pragma unroll
    for (int q = 0; q < NR; q++) if (valid[q]) { float* R = s.J + 8 * (lane + 64 * q); R[JW_WT] = x[q] < 0.f ? D[q] : 0.f; R[JW_X] = x[q]; }
    wave_sync();
    float h[JNV], jtf;
    const int last = ne - 1;
    if (small_tile) {
      acc16x16 C;
      acc_zero(C);
      const int uq = lane >> 4;
      for (int r0 = 0; r0 < ne; r0 += 16) {
        float jv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int r = r0 + 4 * u + uq, rc = r < last ? r : last;
          const v4 ra = ld4(s.J + 8 * rc), rb = ld4(s.J + 8 * rc + 4);
          jv[u] = wrow_entry(cc, cdof_, ra, rb, s.e_con[rc]) + cx * rb.w;
          wv[u] = r < ne ? rb.z : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) wave_mfma_16x16x4(jv[u], wv[u] * jv[u], C);
      }
      JSTAMP_NEWTON(12);
      const int jd = (lane >= JB0 && lane < JNV) ? lane - JB0 : 0;
#pragma unroll
      for (int j2 = 0; j2 < JNV; j2++) {
        float cv = 0.f;
        if (j2 >= JB0) { const int c = j2 - JB0; cv = wave_shfl(C.v[c & 3], jd + 16 * (c >> 2)); }
        h[j2] = lane < nv ? mrow[j2] + (lane >= JB0 ? cv : 0.f) : (lane == j2 ? 1.f : 0.f);
      }
      jtf = wave_shfl(C.v[(JNV - JB0) & 3], jd + 16 * ((JNV - JB0) >> 2));
      jtf = lane >= JB0 ? jtf : 0.f;
    } else {
      acc32x32 C;
      acc_zero(C);
      const int uh = lane >> 5;
      for (int r0 = 0; r0 < ne; r0 += 8) {
        float jv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int r = r0 + 2 * u + uh, rc = r < last ? r : last;
          const v4 ra = ld4(s.J + 8 * rc), rb = ld4(s.J + 8 * rc + 4);
          jv[u] = wrow_entry(cc, cdof_, ra, rb, s.e_con[rc]) + cx * rb.w;
          wv[u] = r < ne ? rb.z : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) wave_mfma_32x32x2(jv[u], wv[u] * jv[u], C);
      }
      JSTAMP_NEWTON(12);
#pragma unroll
      for (int j2 = 0; j2 < JNV; j2++) {
        const int reg = (j2 & 3) + 4 * (j2 >> 3);
        float cv = ((j2 >> 2) & 1) ? wave_shfl(C.v[reg], (lane + 32) & 63) : C.v[reg];
        h[j2] = lane < nv ? mrow[j2] + cv : (lane == j2 ? 1.f : 0.f);
      }
      jtf = ((JNV >> 2) & 1) ? wave_shfl(C.v[(JNV & 3) + 4 * (JNV >> 3)], (lane + 32) & 63) : C.v[(JNV & 3) + 4 * (JNV >> 3)];
    }
    float grad = Ma - (lane < nv ? jtf : 0.f);
    float gn = sqrtf(wave_sum(grad * grad));
    if (gn * scale < tol) break;
    float p = !full ? ldl_solve_blocks(h, -grad, lane, rowblocks) : ((rowblocks & 1) == 0 ? ldl_block<JB0, JNV>(h, -grad, lane) : ldl_solve<true>(h, -grad, lane));
    p = lane < nv ? p : 0.f;
    JSTAMP_NEWTON(13);
    float Mp = mat_vec(mrow, p);
    float pMp, pMa;
    wave_sum2(p * Mp, p * Ma, &pMp, &pMa);
    rows_dot_w<NR>(s, w, con, p, lane, nv, ncon, jp);
#pragma unroll
    for (int q = 0; q < NR; q++) jp[q] = valid[q] ? jp[q] : 0.f;
    float al = 0.f, lo = 0.f, hi = 3.0e38f, d10 = 0.f, dlo = 0.f, dhi = 0.f;
    for (int ls = 0; ls < m->ls_iterations; ls++) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < NR; q++) {
        float xa = x[q] + al * jp[q];
        s1 += xa < 0.f ? D[q] * xa * jp[q] : 0.f;
        s2 += xa < 0.f ? D[q] * jp[q] * jp[q] : 0.f;
      }
      float w1, w2;
      wave_sum2(s1, s2, &w1, &w2);
      float d1 = pMa + al * pMp + w1, d2 = pMp + w2;
      if (ls == 0) d10 = fabsf(d1);
      if (ls > 0 && fabsf(d1) <= 1e-6f * d10) break;
      if (d1 < 0.f) { lo = al; dlo = -d1; } else { hi = al; dhi = d1; }
      if (hi < 1.0e38f && fmaxf(dlo, dhi) * (hi - lo) * scale < 1e-3f * tol) break;
      const float stepn = d1 / d2;
      float nx = al - stepn;
      if (!(nx > lo && nx < hi)) nx = hi < 1.0e38f ? 0.5f * (lo + hi) : 2.f * al + 1.f;
      if (nx == al) break;
      al = nx;
      nls++;
    }
    float dc = 0.f;
    bool switched = false;
#pragma unroll
    for (int q = 0; q < NR; q++) {
      float dx = al * jp[q], xn = x[q] + dx;
      bool was = x[q] < 0.f, is = xn < 0.f;
      switched = switched || was != is;
      dc += (was && is) ? 0.5f * D[q] * dx * (2.f * x[q] + dx) : (is ? 0.5f * D[q] * xn * xn : (was ? -0.5f * D[q] * x[q] * x[q] : 0.f));
      x[q] = xn;
    }
    const bool any_switched = wave_ballot(switched) != 0ull;
    float improvement = -(al * pMa + 0.5f * al * al * pMp + wave_sum(dc));
    JSTAMP_NEWTON(14);
    a += al * p; Ma += al * Mp;
    if (improvement * scale < tol || newton_next_round_is_idle(any_switched, al, improvement, scale, tol)) { it++; break; }
  }
#pragma unroll
  for (int q = 0; q < NR; q++) f[q] = x[q] < 0.f ? -D[q] * x[q] : 0.f;
  out.qacc = a;
  out.iters = it | (nls << 8);
  // J^T f: row forces -> one wrench per contact (its rows share the body pair; kept in the contact frame's tangent slots, dead since the row
  // builder) -> per dof the wrenches of the contacts it moves, dotted with its motion subspace; joint-limit rows add +-f to their dof
  wave_sync();
#pragma unroll
  for (int q = 0; q < NR; q++) if (valid[q]) { s.e_f[lane + 64 * q] = f[q]; s.J[8 * (lane + 64 * q) + JW_WT] = f[q]; }
  wave_sync();
  for (int ci = lane; ci < ncon; ci += 64) {
    const int cd = s.c_dim[ci], nrow = cd == 1 ? 1 : 2 * (cd - 1), r0 = s.c_efc[ci];
    float F[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < nrow; e++) {
      const v4 ra = ld4(s.J + 8 * (r0 + e)), rb = ld4(s.J + 8 * (r0 + e) + 4);
      F[0] += rb.z * ra.x; F[1] += rb.z * ra.y; F[2] += rb.z * ra.z; F[3] += rb.z * ra.w; F[4] += rb.z * rb.x; F[5] += rb.z * rb.y;
    }
#pragma unroll
    for (int c = 0; c < 6; c++) s.c_frame[ci][3 + c] = F[c];
  }
  wave_sync();
  {
    const WCol own = wcol_load(s, lane, nv);
    float acc = 0.f;
    for (int c = 0; c < ncon; c++) {
      const int obs = s.c_ob[c];
      const float sgn = (float)((int)((own.mv >> ((obs >> 24) & 31)) & 1u) - (int)((own.mv >> ((obs >> 8) & 31)) & 1u));
      const float* F = s.c_frame[c] + 3;
      acc += sgn * (F[0] * own.S.a.x + F[1] * own.S.a.y + F[2] * own.S.a.z + F[3] * own.S.b.x + F[4] * own.S.b.y + F[5] * own.S.b.z);
    }
    for (int r = 0; r < nlim; r++) {
      const int cn = s.e_con[r];
      if (((cn >> 19) & 31) == lane) acc += s.J[8 * r] * s.J[8 * r + JW_WT];
    }
    out.qfrc_con = lane < nv ? acc : 0.f;
  }
  return out;
}

// Stage S of the contact-free instantiation: every row is a joint-limit row, J_r = +-e_d, at most one per dof.  The same primal Newton
// iteration as stage_newton (same start point, tolerances, exact line search) with the row of dof k held by lane k: the Hessian is
// M + diag(D_k [row k active]) and J^T f = +-f_k -- no matrix-core pass, no LDS traffic inside the loop, one 9 x 9 block factorisation per
// iteration (limits only exist on hinge dofs: block 0).
template <class L>
JDEV NewtonOut stage_newton_limits(const JacoModelDev* m, L& s, const float (&mrow)[JNV], float smooth, float hd, int lane) {
  NewtonOut out;
  const int nv = m->nv, ne = wave_uniform_i(s.nefc);
  out.qfrc_con = 0.f; out.iters = 0; out.qdamped = 0.f; out.have_qdamped = false;
  float h0[JNV];
#pragma unroll
  for (int j = 0; j < JNV; j++) h0[j] = lane < nv ? mrow[j] : (lane == j ? 1.f : 0.f);
  const int blockmask = 1 | (nv > JB0 ? 2 : 0) | (nv > JB1 ? 4 : 0);
  if (ne == 0) {
    if (m->has_damping == 1) {
      out.qacc = ldl_block0_dual(h0, smooth, hd, lane, &out.qdamped) + ldl_solve_blocks(h0, smooth, lane, blockmask & 6);
      out.have_qdamped = true;
    } else out.qacc = ldl_solve_blocks(h0, smooth, lane, blockmask);
    return out;
  }
  // my row: the limit row of dof `lane`, if there is one (rows are few: a scan)
  float sg = 0.f, ar = 0.f, D = 0.f;
  int myrow = -1;
  const int kk = lane < nv ? lane : 0;
  for (int r = 0; r < ne; r++) {
    const float j = s.J[r * JLD + kk];
    if (lane < JB0 && j != 0.f) { sg = j; ar = s.e_aref[r]; D = s.e_D[r]; myrow = r; }
  }
  const bool has = myrow >= 0;
  const float scale = 1.f / (m->meaninertia * (float)(nv > 1 ? nv : 1)), tol = m->tolerance;
  const bool mine = lane < JB0 && lane < nv;
  const float afree = ldl_solve_blocks(h0, smooth, lane, 6 & blockmask);
  float a = mine ? s.qacc_ws[lane] : afree;
  float Ma = mat_vec(mrow, a) - smooth;
  Ma = mine ? Ma : 0.f;
  float x = has ? sg * a - ar : 0.f;
  int it = 0, nls = 0;
  for (; it < m->iterations; it++) {
    lane = wave_opaque_i(lane);
    const bool act = has && x < 0.f;
    float h[JNV];
#pragma unroll
    for (int j = 0; j < JNV; j++) h[j] = lane < nv ? mrow[j] + ((lane == j && act) ? D : 0.f) : (lane == j ? 1.f : 0.f);
    const float jtf = act ? sg * (-D * x) : 0.f;
    const float grad = Ma - jtf;
    const float gn = sqrtf(wave_sum(grad * grad));
    if (gn * scale < tol) break;
    float p = ldl_block<0, JB0>(h, -grad, lane);
    p = lane < nv ? p : 0.f;
    const float Mp = mat_vec(mrow, p);
    float pMp, pMa;
    wave_sum2(p * Mp, p * Ma, &pMp, &pMa);
    const float jp = has ? sg * p : 0.f;
    float al = 0.f, lo = 0.f, hi = 3.0e38f, d10 = 0.f, dlo = 0.f, dhi = 0.f;
    for (int ls = 0; ls < m->ls_iterations; ls++) {   // exact line search, as in stage_newton
      const float xa = x + al * jp;
      const float s1 = (has && xa < 0.f) ? D * xa * jp : 0.f, s2 = (has && xa < 0.f) ? D * jp * jp : 0.f;
      float w1, w2;
      wave_sum2(s1, s2, &w1, &w2);
      const float d1 = pMa + al * pMp + w1, d2 = pMp + w2;
      if (ls == 0) d10 = fabsf(d1);
      if (ls > 0 && fabsf(d1) <= 1e-6f * d10) break;
      if (d1 < 0.f) { lo = al; dlo = -d1; } else { hi = al; dhi = d1; }
      if (hi < 1.0e38f && fmaxf(dlo, dhi) * (hi - lo) * scale < 1e-3f * tol) break;
      float nx = al - d1 / d2;
      if (!(nx > lo && nx < hi)) nx = hi < 1.0e38f ? 0.5f * (lo + hi) : 2.f * al + 1.f;
      if (nx == al) break;
      al = nx;
      nls++;
    }
    const float dx = al * jp, xn = x + dx;
    const bool was = has && x < 0.f, is = has && xn < 0.f;
    const bool any_switched = wave_ballot(was != is) != 0ull;
    const float dc = (was && is) ? 0.5f * D * dx * (2.f * x + dx) : (is ? 0.5f * D * xn * xn : (was ? -0.5f * D * x * x : 0.f));
    x = xn;
    const float improvement = -(al * pMa + 0.5f * al * al * pMp + wave_sum(dc));
    a += al * p; Ma += al * Mp;
    if (improvement * scale < tol || newton_next_round_is_idle(any_switched, al, improvement, scale, tol)) { it++; break; }
  }
  const float f = (has && x < 0.f) ? -D * x : 0.f;
  out.qacc = a;
  out.qfrc_con = sg * f;
  out.iters = it | (nls << 8);
  if (has) s.e_f[myrow] = f;
  return out;
}

// ---------------------------------------------------------------- stage E: integration
template <class L>
JDEV void stage_integrate_pos(const JacoModelDev* m, L& s, int lane) {
  const float h = m->timestep, hl = m->timestep_lo;
  const bool comp = m->compensated != 0;
  // lane = position coordinate: hinge angles and free-body translations advance by h * (their dof's velocity)
  if (lane < m->nq) {
    const int d = s.mc.q_dof[lane];
    if (d >= 0) {
      if (comp) { const f2 r = comp_advance(s.qpos[lane], s.qpos_lo[lane], h, hl, s.qvel[d], s.qvel_lo[d]); s.qpos[lane] = r.hi; s.qpos_lo[lane] = r.lo; }
      else s.qpos[lane] += h * s.qvel[d];
    }
  }
  // lane = free body: quaternion integration with the body-frame angular velocity, then normalisation
  if (lane < m->nbody && s.mc.b_jtype[lane] != JJ_HINGE) {
    const int qa = s.mc.b_qadr[lane], da = s.mc.b_dadr[lane];
    const v3 w = ld3(&s.qvel[da + 3]);
    const float ww = dot(w, w), x2 = 0.25f * h * h * ww;   // (ang / 2)^2, ang = h |w|
    float q0 = s.qpos[qa + 3], q1 = s.qpos[qa + 4], q2 = s.qpos[qa + 5], q3 = s.qpos[qa + 6];
    float l0 = comp ? s.qpos_lo[qa + 3] : 0.f, l1 = comp ? s.qpos_lo[qa + 4] : 0.f, l2 = comp ? s.qpos_lo[qa + 5] : 0.f, l3 = comp ? s.qpos_lo[qa + 6] : 0.f;
    if (ww > 0.f) {
      // q <- q (x) dq,  dq = (cos(ang/2), axis sin(ang/2)), written as the increment  q (cos(ang/2) - 1) + q (x) (0, d)  with
      // d = w (h/2) sinc(ang/2): no axis normalisation, and cos - 1 without cancellation
      float sc, cm1;   // sin(x) / x and cos(x) - 1 at x = ang / 2
      if (x2 < 0.01f) {   // |x| < 0.1 (angular velocity < 200 rad/s at h = 1 ms): series, truncation < 1e-13
        sc = 1.f + x2 * (-1.f / 6.f + x2 * (1.f / 120.f - x2 * (1.f / 5040.f)));
        cm1 = x2 * (-0.5f + x2 * (1.f / 24.f - x2 * (1.f / 720.f)));
      } else {
        const float x = sqrtf(x2), s4 = sinf(0.5f * x);
        sc = sinf(x) / x; cm1 = -2.f * s4 * s4;
      }
      const float k = 0.5f * h * sc, d1 = w.x * k, d2 = w.y * k, d3 = w.z * k;
      const float e0 = q0 * cm1 - q1 * d1 - q2 * d2 - q3 * d3;
      const float e1 = q1 * cm1 + q0 * d1 + q2 * d3 - q3 * d2;
      const float e2 = q2 * cm1 + q0 * d2 - q1 * d3 + q3 * d1;
      const float e3 = q3 * cm1 + q0 * d3 + q1 * d2 - q2 * d1;
      if (comp) {
        f2 r;
        r = comp_add(q0, l0, e0, 0.f); q0 = r.hi; l0 = r.lo;
        r = comp_add(q1, l1, e1, 0.f); q1 = r.hi; l1 = r.lo;
        r = comp_add(q2, l2, e2, 0.f); q2 = r.hi; l2 = r.lo;
        r = comp_add(q3, l3, e3, 0.f); q3 = r.hi; l3 = r.lo;
      } else { q0 += e0; q1 += e1; q2 += e2; q3 += e3; }
    }
    // |q|^2 - 1 (fma chain: ~1e-7 absolute, which is all a norm needs -- a common scale factor does not touch the rotation, and the
    // scaling itself is applied to the compensated pair, so the direction of q keeps its ~48 bits)
    const float r2 = fmaf(q3, q3, fmaf(q2, q2, fmaf(q1, q1, fmaf(q0, q0, -1.f)))) + 2.f * (q0 * l0 + q1 * l1 + q2 * l2 + q3 * l3);
    if (comp && fabsf(r2) < 1e-3f) {
      const float k = r2 * (-0.5f + 0.375f * r2);   // 1 / sqrt(1 + r2) - 1
      f2 r;
      r = comp_add(q0, l0, q0 * k, 0.f); q0 = r.hi; l0 = r.lo;
      r = comp_add(q1, l1, q1 * k, 0.f); q1 = r.hi; l1 = r.lo;
      r = comp_add(q2, l2, q2 * k, 0.f); q2 = r.hi; l2 = r.lo;
      r = comp_add(q3, l3, q3 * k, 0.f); q3 = r.hi; l3 = r.lo;
    } else {   // plain state, or a quaternion handed in un-normalised (set_state; the reference's zero quaternion, env_mujoco_util.py:119-121)
      const float n = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
      if (n < JMINVAL) { q0 = 1.f; q1 = q2 = q3 = 0.f; } else { const float in = 1.f / n; q0 *= in; q1 *= in; q2 *= in; q3 *= in; }
      l0 = l1 = l2 = l3 = 0.f;
    }
    s.qpos[qa + 3] = q0; s.qpos[qa + 4] = q1; s.qpos[qa + 5] = q2; s.qpos[qa + 6] = q3;
    if (comp) { s.qpos_lo[qa + 3] = l0; s.qpos_lo[qa + 4] = l1; s.qpos_lo[qa + 5] = l2; s.qpos_lo[qa + 6] = l3; }
  }
}

#include "collision.h"
#include "env_logic.h"

// ---------------------------------------------------------------- stage S, side solve (light tier, split mode: collision.h "Side rows")
// The pedestal block's sub-problem on its own: six dofs (lanes JB1 .. JNV-1), up to JSIDE_ROWS rows (lane = row), same primal Newton with
// exact line search as stage_newton; the Hessian is built by a loop over the rows (a handful of rank-1 updates of a 6 x 6 matrix: not
// worth a matrix-core pass).  Overwrites the block's entries of the main solve's output (there they are the unconstrained solution).
template <class L>
JDEV void newton_side(const JacoModelDev* m, L& s, const float (&mrow)[JNV], float smooth, int lane, NewtonOut& out) {
  float* sd = side_buf(s);
  int ns = wave_uniform_i(s.nside);
  ns = ns < 0 ? 0 : (ns > JSIDE_ROWS ? JSIDE_ROWS : ns);
  const int nv = m->nv;
  const bool mine = lane >= JB1 && lane < JNV, vr = lane < ns;
  const int kk = mine ? lane - JB1 : 0, rr = vr ? lane : 0;
  float js[6];
#pragma unroll
  for (int k = 0; k < 6; k++) js[k] = vr ? sd[JSIDE_J + rr * 6 + k] : 0.f;
  const float D = vr ? sd[JSIDE_D + rr] : 0.f, ar = vr ? sd[JSIDE_AREF + rr] : 0.f;
  const float scale = 1.f / (m->meaninertia * (float)(nv > 1 ? nv : 1)), tol = m->tolerance;
  wave_sync();   // (aref / D are in registers: their slots become the staging area of the loop)
  float a = mine ? s.qacc_ws[lane] : 0.f;
  float Ma = 0.f, x = 0.f;
#pragma unroll
  for (int j = 0; j < 6; j++) { const float aj = wave_bcast(a, JB1 + j); Ma += mrow[JB1 + j] * aj; x += js[j] * aj; }
  Ma = mine ? Ma - smooth : 0.f;
  x = vr ? x - ar : 0.f;
  float jtf = 0.f;
  int it = 0;
  for (; it < m->iterations; it++) {
    lane = wave_opaque_i(lane);   // (as in stage_newton)
    if (vr) { sd[JSIDE_F + lane] = x < 0.f ? D : 0.f; sd[JSIDE_AREF + lane] = x < 0.f ? -D * x : 0.f; }
    wave_sync();
    float h[JNV];
#pragma unroll
    for (int j = 0; j < JNV; j++) h[j] = (j >= JB1 && mine) ? mrow[j] : 0.f;
    jtf = 0.f;
    for (int r = 0; r < ns; r++) {
      const float* Jr = sd + JSIDE_J + r * 6;
      const float jk = Jr[kk], t = sd[JSIDE_F + r] * jk;
      jtf += jk * sd[JSIDE_AREF + r];
#pragma unroll
      for (int j = 0; j < 6; j++) h[JB1 + j] += t * Jr[j];
    }
    wave_sync();
    const float grad = mine ? Ma - jtf : 0.f;
    const float gn = sqrtf(wave_sum(grad * grad));
    if (gn * scale < tol) break;
    const float p = ldl_block<JB1, JNV>(h, -grad, lane);
    float Mp = 0.f, jp = 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) { const float pj = wave_bcast(p, JB1 + j); Mp += mrow[JB1 + j] * pj; jp += js[j] * pj; }
    Mp = mine ? Mp : 0.f;
    jp = vr ? jp : 0.f;
    float pMp, pMa;
    wave_sum2(p * Mp, p * Ma, &pMp, &pMa);
    float al = 0.f, lo = 0.f, hi = 3.0e38f, d10 = 0.f, dlo = 0.f, dhi = 0.f;
    for (int ls = 0; ls < m->ls_iterations; ls++) {   // exact line search, as in stage_newton
      const float xa = x + al * jp;
      const float s1 = xa < 0.f ? D * xa * jp : 0.f, s2 = xa < 0.f ? D * jp * jp : 0.f;
      float w1, w2;
      wave_sum2(s1, s2, &w1, &w2);
      float d1 = pMa + al * pMp + w1, d2 = pMp + w2;
      if (ls == 0) d10 = fabsf(d1);
      if (ls > 0 && fabsf(d1) <= 1e-6f * d10) break;
      if (d1 < 0.f) { lo = al; dlo = -d1; } else { hi = al; dhi = d1; }
      if (hi < 1.0e38f && fmaxf(dlo, dhi) * (hi - lo) * scale < 1e-3f * tol) break;
      float nx = al - d1 / d2;
      if (!(nx > lo && nx < hi)) nx = hi < 1.0e38f ? 0.5f * (lo + hi) : 2.f * al + 1.f;
      if (nx == al) break;
      al = nx;
    }
    const float dx = al * jp, xn = x + dx;
    const bool was = x < 0.f, is = xn < 0.f;
    const bool any_switched = wave_ballot(was != is) != 0ull;
    const float dc = (was && is) ? 0.5f * D * dx * (2.f * x + dx) : (is ? 0.5f * D * xn * xn : (was ? -0.5f * D * x * x : 0.f));
    x = xn;
    const float improvement = -(al * pMa + 0.5f * al * al * pMp + wave_sum(dc));
    a += al * p; Ma += al * Mp;
    if (improvement * scale < tol || newton_next_round_is_idle(any_switched, al, improvement, scale, tol)) { it++; break; }
  }
  // row forces (touch stage) and J^T f of the block
  const float f = x < 0.f ? -D * x : 0.f;
  wave_sync();
  if (vr) sd[JSIDE_F + lane] = f;
  wave_sync();
  float qf = 0.f;
  for (int r = 0; r < ns; r++) qf += sd[JSIDE_J + r * 6 + kk] * sd[JSIDE_F + r];
  out.qacc = mine ? a : out.qacc;
  out.qfrc_con = mine ? qf : out.qfrc_con;
  if (it > (out.iters & 255)) out.iters = (out.iters & ~255) | (it & 255);
}

// Append `env` to tier queue `t` from inside a running launch.  Everything the consumer will read (state rows, task row,
// remaining, ...) must already have gone out with write-through stores; the entry is published after they are acknowledged.
JDEV void queue_push(const JacoStepArgs& A, int t, int env, int left, int lane) {
  if (lane == 0) st_wt_i(&A.remaining[env], left);
  dev_stores_done();
  wave_sync();
  if (lane == 0) {
    int slot = jaco_atomic_inc(A.q[t].count);
    st_wt_i(&A.q[t].list[slot], env);
    dev_stores_done();   // (the entry is out before the count of running light workgroups drops)
  }
}
// a tier (1..3) that really had to be used in this step is remembered for the env's next step
JDEV void hint_raise(const JacoStepArgs& A, int env, int tier, int lane) {
  if (lane == 0 && A.hint) atomicMax(&A.hint[env], tier);
}

// ---------------------------------------------------------------- the kernels
// One substep loop for one env; returns the number of substeps NOT done (light tier bail-out) or 0.
// TIER: 0 light, 1 medium, 2 heavy, 3 huge.  Tiers below 3 stop at a capacity overflow (*why = 1) and leave the env to the next
// tier; tiers above 0 can give the env back to the tier below once it would fit again (handback; *why = 2).
// FULL = false: the instantiation of the step kernel proper (jaco_physics_kernel), which is only ever launched in modes 0 (ctrl level)
// and 1 (env step): the reset-time modes (forward pass, placing hold, grasping pre-reach, take_action / terminal_inspection on their own)
// fold away at compile time and stay out of the hot kernel's code and register budget; their launches use jaco_physics_kernel_listed.
template <class C, int TIER, bool FULL = true>
JDEV int run_env(const JacoStepArgs& A_, JacoLDS<C>& s, int env, int nsub, int lane, bool handback = false, int* why = nullptr) {
  const JacoStepArgs* Ap = args_view(A_);
#define A (*Ap)
  constexpr bool LIGHT = TIER == 0;
  bool bailed = false;
  const JacoModelDev* m = opaque_ptr(A.model);
  const int nq = m->nq, nv = m->nv, nu = m->nu, ns = m->nsensor;
  if (FULL && (A.env_mode == 3 || A.env_mode == 2 || A.env_mode == 6) && A.mask && !A.mask[env]) return 0;   // masked reset: the other envs are not touched
  if (lane < nq) { s.qpos[lane] = A.qpos[(size_t)env * nq + lane]; s.qpos_lo[lane] = A.qpos_lo ? A.qpos_lo[(size_t)env * nq + lane] : 0.f; }
  if (lane < nv) { s.qvel[lane] = A.qvel[(size_t)env * nv + lane]; s.qacc_ws[lane] = A.qacc_ws[(size_t)env * nv + lane]; s.qvel_lo[lane] = A.qvel_lo ? A.qvel_lo[(size_t)env * nv + lane] : 0.f; }
  // ctrl-level launches read the caller's controls.  Env-level launches compute theirs (controller + gripper ramp, or the interrupted substep's
  // from the task row); a forward pass alone (mode 2: sim.forward() after sim.reset(), which zeroes data.ctrl) runs on ZERO controls -- A.ctrl is
  // only a readable placeholder there (the qvel buffer: reading it made a reset's first touch reading depend on other envs' velocities;
  // found by the auto-reset bit-identity test in round 5, 1 env of 8 192 x 6 steps)
  if (lane < nu) s.ctrl[lane] = A.env_mode == 0 ? A.ctrl[(size_t)env * nu + lane] : 0.f;
  stage_model(m, s, lane);
  if (lane == 0) { s.ncon = 0; s.nefc = 0; s.ncand = 0; s.nlimit = 0; s.nsphere = 0; s.nside = 0; s.nside_cand = 0; }   // (LDS is not zeroed between workgroups)
  unsigned flags = LIGHT ? 0u : JFLAG_HEAVY_TIER;
  float sens = 0.f;
  int iters = 0, left = 0, sub0 = 0, nls_last = 0, calm = 0;
  bool tier_used = false;   // this tier's extra capacity was really needed in at least one substep
  const int emode = FULL ? A.env_mode : (A.env_mode != 0 ? 1 : 0);
  if (emode == 4 || emode == 5) nsub = 0;   // take_action / terminal_inspection on their own: no substep runs
  // poses of the two task-layer markers: LDS copy for this launch (their geoms are re-posed every substep)
  if (lane < 24) s.mk[lane] = A.marker ? A.marker[(size_t)env * 24 + lane] : m->marker_rest[lane / 12][lane % 12];
  if (emode) {   // task row + the one-substep-stale quantities the controller reads (env_logic.h)
    if (lane < JTASK_N) s.task[lane] = A.task[(size_t)env * JTASK_N + lane];
    const float* CR = A.cache + (size_t)env * JCACHE_N;
    if (lane < 36) { s.M[m_index(lane / 6, lane % 6)] = CR[JC_M + lane]; s.cdof[lane / 6][lane % 6] = CR[JC_CDOF + lane]; }
    if (lane < 6) s.bias[lane] = CR[JC_BIAS + lane];
    if (lane < 3) { s.xpos[m->ee_body][lane] = CR[JC_EEPOS + lane]; s.xpos[m->obj_body >= 0 ? m->obj_body : 0][lane] = CR[JC_OBJPOS + lane]; }
    if (lane < 9) s.xmat[m->ee_body][lane] = CR[JC_EEMAT + lane];
  }
  wave_sync();
  if (emode == 1 || emode == 4 || emode == 5) {
    if (s.task[JT_DONE] != 0.f) {   // finished (or quarantined) and not yet reset: frozen (no auto-reset); counters untouched
      if (lane == 0 && emode != 4) { A.reward[env] = 0.f; A.done[env] = 1; }
      return 0;
    }
  }
  bool fwd = false;   // auto-reset: this pass is the reset env's sim.forward() + observation (one substep's derived quantities, no integration)
  if (emode == 1 && s.task[JT_FWD] != 0.f) {   // ... resumed in a bigger tier after the forward pass overflowed the tier below
    fwd = true; nsub = 1;
    if (lane < nu) s.ctrl[lane] = 0.f;
    if (lane == 0) { s.task[JT_SUB] = 0.f; s.task[JT_PENDING] = 0.f; }
    wave_sync();
  }
  if ((emode == 1 && !fwd) || emode == 4) {
    sub0 = wave_uniform_i((int)s.task[JT_SUB]);
    // (a step that overflowed this tier in its very first substep comes back with JT_SUB = 0 and the interrupted substep pending:
    // its action has been taken already -- taking it again would add the gripper increment twice and shift the draw counter)
    if (sub0 == 0 && s.task[JT_PENDING] == 0.f) {
      // _take_action: the marker placement consumes 6 draws (a8), then the EE target and the gripper ramp
      // the previous target (subgoal_reach = policy sub-goal + self.target_pos, :609, reads it before the new one is set)
      float prev_tg[6];
      for (int k = 0; k < 6; k++) prev_tg[k] = s.task[JT_TARGET + k];
      take_action(m, s, A.action + (size_t)env * A.nact, A.nact, lane);
      const unsigned cnt0 = rng_count(s.task[JT_RNG]);
      wave_sync();
      if (A.marker) {   // set_mocap_*("subgoal_reach", rule-based sub-goal) and set_mocap_*("hand", new target) (:613-615,:644-646)
        float nz[6], spos[3], sori[3];
        for (int k = 0; k < 6; k++) nz[k] = A.noise ? A.noise[(size_t)env * 12 + k] : rng_uniform(A.seed, (unsigned)env, cnt0 + k);
        v3 pe; m3 Re;
        ee_frame(m, s, &pe, &Re);
        const int ob = m->obj_body >= 0 ? m->obj_body : 0;
        rulebased_subgoal(A.task_id, pe, ld3(s.task + JT_OBJGOAL), s.xpos[ob][1], ld3(s.task + JT_DESTGOAL), nz, spos, sori);
        bool move_sub = true;
        if (A.obs_mode == 1) {   // no rule-based sub-goal: the marker follows the policy's sub-goal, if the caller passes one
          move_sub = A.subgoal != nullptr;
          if (move_sub) for (int k = 0; k < 3; k++) { spos[k] = A.subgoal[(size_t)env * 6 + k] + prev_tg[k]; sori[k] = A.subgoal[(size_t)env * 6 + 3 + k] + prev_tg[3 + k]; }
        }
        // every lane whose geom rides on a marker stores that marker's pose itself (it is also the lane that reads it back)
        const int km = lane < m->ngeom ? m->g_marker[lane] : -1;
        if (km >= 0 && (km == 0 || move_sub)) {
          const float* tg = s.task + JT_TARGET;
          v3 mp = km == 0 ? ld3(tg) : mk3(spos[0], spos[1], spos[2]);
          m3 MR = euler_rxyz_to_mat(km == 0 ? tg[3] : sori[0], km == 0 ? tg[4] : sori[1], km == 0 ? tg[5] : sori[2]);
          // write-through: if this env is handed to a heavy-tier workgroup later in the step, that workgroup (possibly on
          // another XCD) reads the pose from memory
          float* P = A.marker + (size_t)env * 24 + 12 * km;
          float* Q = s.mk + 12 * km;   // (every geom lane of the marker writes the same values)
          st_wt(P + 0, mp.x); st_wt(P + 1, mp.y); st_wt(P + 2, mp.z);
          Q[0] = mp.x; Q[1] = mp.y; Q[2] = mp.z;
#pragma unroll
          for (int k = 0; k < 9; k++) { st_wt(P + 3 + k, MR.m[k]); Q[3 + k] = MR.m[k]; }
        }
      }
      if (lane == 0 && A.obs_mode == 0) s.task[JT_RNG] = rng_slot(cnt0 + 6u);
      wave_sync();
    }
    osc_target_quat(s, lane);
    wave_sync();
  }
  float pinv = 0.f;   // mode 3, lanes 9..15: the pinned object pose
  if (emode == 3) {
    // placing reset (env_mujoco_util.py:106-117): the object goes to the grasp frame EE_obj, 4 cm back along its x axis, and
    // is re-pinned there (zero velocity for every free body) after each of the nsub substeps, while the controller
    // holds the EE at its reset pose and the finger servos close onto the object (gripper command 0.6).
    sub0 = wave_uniform_i((int)s.task[JT_SUB]);
    float* PIN = A.cache + (size_t)env * JCACHE_N + JC_PIN;
    if (sub0 == 0) {
      stage_walk(m, s, lane, false);
      wave_sync();
      v3 po, pe; m3 Ro, Re;
      eeobj_frame(m, s, &po, &Ro);
      ee_frame(m, s, &pe, &Re);
      po = po - col(Ro, 0) * 0.04f;
      float q[4], eul[3];
      mat_to_quat(Ro, q);
      mat_to_euler_rxyz(Re, eul);
      float pin[7] = {po.x, po.y, po.z, q[0], q[1], q[2], q[3]};
#pragma unroll
      for (int k = 0; k < 7; k++) if (lane == 9 + k) pinv = pin[k];
      if (lane >= 9 && lane < 16) st_wt(&PIN[lane - 9], pinv);   // (write-through: read back by the heavy tier after a hand-off)
      if (lane == 0) {
        float* t = s.task;
        t[JT_TARGET + 0] = pe.x; t[JT_TARGET + 1] = pe.y; t[JT_TARGET + 2] = pe.z;
        t[JT_TARGET + 3] = eul[0]; t[JT_TARGET + 4] = eul[1]; t[JT_TARGET + 5] = eul[2];
        t[JT_GRIP] = 0.6f; t[JT_GRIP_PREV] = 0.6f;
      }
      wave_sync();
      if (lane >= 9 && lane < 16) { s.qpos[lane] = pinv; s.qpos_lo[lane] = 0.f; }
      if (lane >= 9 && lane < nv) { s.qvel[lane] = 0.f; s.qvel_lo[lane] = 0.f; }
    } else if (lane >= 9 && lane < 16) pinv = PIN[lane - 9];   // resumed by the heavy tier
    wave_sync();
    osc_target_quat(s, lane);
    wave_sync();
  }
  if (emode == 6) {
    // grasping reset (env_mujoco_util.py:123-170): EE target = [object goal, orientation looking along EE -> object] (float16 angles,
    // yaw drawn), then loop 1: { stop_obj (free-body velocities zeroed, sim.forward), controller + sim.step } until the EE is within
    // 0.2 m of the object goal (the target then becomes the EE's own position) or its orientation within pi/6 of the sampled reaching
    // goal's; loop 2: { controller + sim.step, target back to the object goal } until the EE is within 0.15 m.  The reference's
    // loops are unbounded; here `nsub` caps them (JFLAG_PREREACH_CAP).  JT_PHASE: 1 = loop 1, 2 = loop 2, 3 = done.
    sub0 = wave_uniform_i((int)s.task[JT_SUB]);
    if (s.task[JT_PHASE] == 0.f) {
      stage_walk(m, s, lane, false);
      wave_sync();
      v3 pe; m3 Re;
      ee_frame(m, s, &pe, &Re);
      const unsigned cnt0 = rng_count(s.task[JT_RNG]);
      const float gamma = -0.1f + 0.2f * (A.noise ? A.noise[(size_t)env * 12] : rng_uniform(A.seed, (unsigned)env, cnt0));
      float ori[3];
      grasp_reach_ori(pe, ld3(s.task + JT_OBJGOAL), gamma, ori);
      wave_sync();
      if (lane == 0) {
        float* t = s.task;
        for (int k = 0; k < 3; k++) { t[JT_TARGET + k] = t[JT_OBJGOAL + k]; t[JT_TARGET + 3 + k] = ori[k]; }
        t[JT_GRIP] = 0.6f; t[JT_GRIP_PREV] = 0.6f; t[JT_PHASE] = 1.f;
        if (!A.noise) t[JT_RNG] = rng_slot(cnt0 + 1u);
      }
      wave_sync();
    }
    osc_target_quat(s, lane);
    wave_sync();
    if (s.task[JT_PHASE] >= 3.f) nsub = sub0;   // (already there: only the observation is left)
  }
  const bool markers = A.marker != nullptr;
  JProfCtx pc;
  pc.row = nullptr;
  pc.tprev = 0;
#ifdef JACO_PROFILE_STAGES
  pc.row = A.prof ? A.prof + (size_t)env * JPROF_N : nullptr;
  pc.tprev = __builtin_amdgcn_s_memtime();
#endif
  PairList<C> pl;   // broadphase pair list, carried from substep to substep (collision.h)
again:
  pl.n = -1;
  for (int sub = sub0; sub < nsub; sub++) {
    Ap = args_view(A_);
    m = opaque_ptr(A.model);
    // Same for the lane id: everything a lane addresses (LDS offsets, "lane < n" masks, model table slots) derives from it, and the
    // optimiser would otherwise compute ~140 such values once, before the loop, and keep them alive across all of it -- in scratch
    // memory (556 B of spills per lane, every use a scratch load).  Recomputing them where they are used is one or two VALU
    // instructions each; the kernel's remaining spills (156 B) all sit inside the MPR routine.
    lane = wave_opaque_i(lane);
    bool held_pending = false;
    const int phase = emode == 6 ? wave_uniform_i((int)s.task[JT_PHASE]) : 0;
    if ((emode == 3 || emode == 6) && s.task[JT_PENDING] != 0.f) {   // resume of an interrupted held substep: its ctrl was saved
      if (lane < nu) s.ctrl[lane] = s.task[JT_CTRL + lane];
      wave_sync();
      if (lane == 0) s.task[JT_PENDING] = 0.f;
      held_pending = true;
    }
    if (emode == 6 && phase == 2 && !held_pending) {   // _step_simulation() with what mjData holds from the previous sim.step (one substep stale)
      stage_osc(m, s, lane, flags);
      if (lane >= 6 && lane < nu) s.ctrl[lane] = 0.6f;
      wave_sync();
    }
    if (emode == 6 && phase == 1 && !held_pending) {   // stop_obj (mujoco.py:239-246): free-body velocities zeroed before the forward pass
      if (lane >= 9 && lane < nv) { s.qvel[lane] = 0.f; s.qvel_lo[lane] = 0.f; }
      wave_sync();
    }
    if (emode == 1 && !fwd) {
      if (s.task[JT_PENDING] != 0.f) {   // resume of a substep interrupted by the light tier: its ctrl was saved
        if (lane < nu) s.ctrl[lane] = s.task[JT_CTRL + lane];
        wave_sync();
        if (lane == 0) s.task[JT_PENDING] = 0.f;
      } else {
        stage_osc(m, s, lane, flags);   // _step_simulation: feedback -> OSC torque (env_mujoco_util.py:73-90)
        JSTAMP(15);
        if (lane >= 6 && lane < nu) {   // gripper ramp np.linspace(prev, new, skip_frames)[gripper_iter] (:631,:78)
          float g0 = s.task[JT_GRIP_PREV], g1 = s.task[JT_GRIP];
          s.ctrl[lane] = nsub > 1 ? g0 + (g1 - g0) * ((float)sub / (float)(nsub - 1)) : g1;
        }
        wave_sync();
      }
    }
#ifdef JACO_WALK_PROFILE
    stage_walk(m, s, lane, markers, &pc);
#else
    stage_walk(m, s, lane, markers);
#endif
    for (int i = lane; i < JMBLK; i += 64) s.M[i] = 0.f;
    wave_sync();
    if constexpr (C::CONTACT) if (A.dbg && env == A.dbg_env && sub == nsub - 1 && lane < m->ngeom) for (int k = 0; k < 3; k++) A.dbg[JDBG_GPOS + 3 * lane + k] = s.gpos[lane][k];
    JSTAMP(0);
    JSTAMP(1);
    const ActParams actp = act_fetch(m, lane);
    const StagePrefetch pf = stage_prefetch(m, lane);
    stage_accumulate(m, s, lane);
    wave_sync();
    stage_mass_bias(m, s, lane, pf);
    wave_sync();
    if ((emode == 3 || (emode == 6 && phase == 1)) && !held_pending) {   // the substep follows a sim.forward(): the controller sees *this* state's M, J, bias
      stage_osc(m, s, lane, flags);
      if (lane >= 6 && lane < nu) s.ctrl[lane] = 0.6f;
      wave_sync();
    }
    stage_actuation(m, s, lane, actp);
    JSTAMP(2);
    // collision first, then the row builders: the constraint rows share LDS with the geom poses and the broadphase survivors,
    // which are dead once the contact list exists.  (Actuation touches neither: no synchronisation of its own.)
    unsigned cflags = 0;
    if (C::CONTACT && !A.disable_contact) {
      stage_collision(A, m, s, env, lane, cflags, pc, pl);
      wave_sync();
      JSTAMP(4);
      stage_limit_rows(m, s, lane, pf);
      wave_sync();
      JSTAMP(3);
      stage_contact_rows(m, s, lane, cflags);
      wave_sync();
      JSTAMP(5);
    } else {
      stage_limit_rows(m, s, lane, pf);
      if (lane == 0) { s.ncon = 0; s.ncand = 0; s.nsphere = 0; s.nside = 0; s.nside_cand = 0; }
      wave_sync();
    }
    if (!LIGHT) {   // would the tier below have coped with this substep?  (huge -> heavy, heavy -> medium, medium -> light)
      constexpr int LCON = TIER == 3 ? JacoHeavy::MAXCON : (TIER == 2 ? JacoMedium::MAXCON : JacoLight::MAXCON);
      constexpr int LEFC = TIER == 3 ? JacoHeavy::MAXEFC : (TIER == 2 ? JacoMedium::MAXEFC : JacoLight::MAXEFC);
      constexpr int LCAND = TIER == 3 ? JacoHeavy::MAXCAND : (TIER == 2 ? JacoMedium::MAXCAND : JacoLight::MAXCAND);
      // (the light tier also copes when the rows fit without the pedestal's, which it solves on the side: collision.h "Side rows")
      const bool rows_fit = s.nefc <= LEFC || (TIER == 1 && SideRows<JacoLDS<JacoLight>>::on && s.nside_cand > 0 && s.nefc - s.nside_cand <= LEFC);
      const bool fits = s.ncon <= LCON && rows_fit && s.nsphere <= LCAND;   // (head room below the capacity was tried: no gain)
      calm = fits ? calm + 1 : 0;
      if (!fits && !tier_used) { tier_used = true; if (A.hint_mode != 2) hint_raise(A, env, TIER, lane); }
    }
    if (TIER == 3) flags |= cflags;   // the last tier has nobody to hand over to: contacts / rows beyond its capacity were dropped, say so
    if (TIER < 3 && cflags) {   // capacity exceeded: leave this substep (and the rest) to the next tier; nothing was mutated
      left = nsub - sub;
      bailed = true;
      flags |= (cflags & 7u) << (JFLAG_BAIL_CAUSE_SHIFT + 3 * TIER);   // informational: which capacity of which tier sent the env on
      if (emode == 1 || emode == 3 || emode == 6) {
        if (lane < nu) s.task[JT_CTRL + lane] = s.ctrl[lane];
        if (lane == 0) { s.task[JT_PENDING] = 1.f; s.task[JT_SUB] = (float)sub; }
        wave_sync();
      }
      break;
    }
    float mrow[JNV], h[JNV];   // M[lane][:] stays in registers for the rest of the substep
    {   // row `lane` of the block-diagonal mass matrix: only the lane's own dof block is stored
      const int blo = lane < JB0 ? 0 : (lane < JB1 ? JB0 : JB1), bn = lane < JB0 ? JB0 : (lane < JB1 ? JB1 - JB0 : JNV - JB1);
      const int rbase = lane < nv ? m_index(lane, blo) : 0;
#pragma unroll
      for (int j = 0; j < JNV; j++) { const bool in = lane < nv && j >= blo && j < blo + bn; mrow[j] = in ? s.M[in ? rbase + j - blo : 0] : 0.f; }
    }
    float smooth = lane < nv ? s.smooth[lane] : 0.f;
    float qas = 0.f;   // (qacc_smooth is no longer formed; kept in the dump layout)
    wave_sync();
    const float hdamp = (m->has_damping && lane < nv) ? m->timestep * pf.damping : 0.f;
    NewtonOut nw;
    if constexpr (C::WRENCH) nw = stage_newton_w(m, s, mrow, smooth, hdamp, lane, pc);
    else if constexpr (C::CONTACT) nw = stage_newton(m, s, mrow, smooth, hdamp, lane, pc);
    else nw = stage_newton_limits(m, s, mrow, smooth, hdamp, lane);
    if (SideRows<JacoLDS<C>>::on) { if (wave_uniform_i(s.nside) > 0) newton_side(m, s, mrow, smooth, lane, nw); }
    JSTAMP(6);
    iters = nw.iters & 255;
    int nls_dbg = nw.iters >> 8;
    nls_last = nls_dbg;
    if (iters >= m->iterations) flags |= JFLAG_SOLVER_MAXITER;
    wave_sync();
    if (C::CONTACT) stage_touch(m, s, lane, &sens);
    JSTAMP(7);
    // Euler with implicit joint damping
    float total = smooth + nw.qfrc_con, qacc_e = nw.qacc;
    if (m->has_damping) {   // (M + h D) qacc = total; blocks without damping keep the solver's qacc (M qacc = total there)
      float qd;
      if (nw.have_qdamped) qd = nw.qdamped;   // (already solved next to M^-1 qfrc_smooth: the block carried no row)
      else {
#pragma unroll
        for (int j = 0; j < JNV; j++) h[j] = (lane < nv ? mrow[j] : 0.f) + (lane == j ? (lane < nv ? hdamp : 1.f) : 0.f);
        qd = ldl_solve_blocks(h, total, lane, JDAMPED_BLOCKS);
      }
      qacc_e = lane < JB0 ? qd : nw.qacc;
    }
    if (A.dbg && env == A.dbg_env && sub == nsub - 1) {
      float* D = A.dbg;
      if (lane < m->nbody) {
        for (int k = 0; k < 3; k++) D[JDBG_XPOS + 3 * lane + k] = s.xpos[lane][k];
        for (int k = 0; k < 9; k++) D[JDBG_XMAT + 9 * lane + k] = s.xmat[lane][k];
      }
      for (int i = lane; i < JNV * JNV; i += 64) {
        const int d = i / JNV, j = i % JNV;
        const bool same = (d < JB0) == (j < JB0) && (d < JB1) == (j < JB1);
        D[JDBG_M + i] = same ? s.M[m_index(d, j)] : 0.f;
      }
      if (lane < nv) {
        D[JDBG_BIAS + lane] = s.bias[lane]; D[JDBG_SMOOTH + lane] = smooth; D[JDBG_QACC_SMOOTH + lane] = qas;
        D[JDBG_QACC + lane] = nw.qacc; D[JDBG_QFRC_CON + lane] = nw.qfrc_con;
      }
      if (lane == 0) { D[JDBG_NCON] = (float)s.ncon; D[JDBG_NCON + 1] = (float)(s.nefc + s.nside); D[JDBG_NCON + 2] = (float)iters; D[JDBG_NCON + 3] = (float)s.ncand; }
      for (int c = lane; c < s.ncon && c < JDBG_MAXCON; c += 64) {
        float* o = D + JDBG_CONTACT + 8 * c;
        o[0] = s.c_dist[c]; o[1] = s.c_pos[c][0]; o[2] = s.c_pos[c][1]; o[3] = s.c_pos[c][2];
        o[4] = s.c_frame[c][0]; o[5] = s.c_frame[c][1]; o[6] = s.c_frame[c][2]; o[7] = (float)s.c_pair[c];
      }
      for (int r = lane; r < s.nefc && r < JDBG_MAXEFC; r += 64) {
        float* o = D + JDBG_EFC + 4 * r;
        o[0] = s.e_aref[r]; o[1] = 1.f / s.e_D[r]; o[2] = 0.f; o[3] = s.e_f[r];
      }
      if (lane < s.ncon) D[JDBG_CFN + lane] = s.c_fn[lane];
      if (lane < m->nsensor) D[JDBG_SENS + lane] = sens;
    }
    wave_sync();
    if (emode == 2 || fwd) break;   // sim.forward(): derived quantities only, no integration
    if (lane < nv) {
      float v;
      if (m->compensated) {   // qvel += h qacc on the compensated pair
        const f2 nvl = comp_advance(s.qvel[lane], s.qvel_lo[lane], m->timestep, m->timestep_lo, qacc_e, 0.f);
        v = nvl.hi; s.qvel_lo[lane] = nvl.lo;
      } else v = s.qvel[lane] + m->timestep * qacc_e;
      s.qvel[lane] = v;
      s.qacc_ws[lane] = nw.qacc;
      if (!(v == v) || fabsf(v) > 1e10f) flags |= JFLAG_NAN;
    }
    wave_sync();
    stage_integrate_pos(m, s, lane);
    wave_sync();
    // Sensitivity probes (diagnostic builds only, tools/build_variant.sh; profiles/r03_ab_probe.txt): what paces the kernel?
    // +640 independent fma per substep (+10 % VALU) cost +3.7 %; 12 extra dependent LDS round trips per substep cost +0.6 %.
#ifdef JACO_PROBE_VALU
    {
      float pa[8];
#pragma unroll
      for (int k = 0; k < 8; k++) pa[k] = s.qvel[(lane + k) & 15];
      for (int i = 0; i < JACO_PROBE_VALU / 8; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) pa[k] = fmaf(pa[k], 1.0000001f, 1e-30f);
      }
      float pr = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) pr += pa[k];
      if (pr == 123.456f) s.smooth[40] = pr;   // (never true: keeps the chains alive)
    }
#endif
#ifdef JACO_PROBE_LDS
    {
      float pv = s.qvel[lane & 15];
      for (int i = 0; i < JACO_PROBE_LDS; i++) {
        s.smooth[48 + (lane & 15)] = pv;
        wave_sync();
        pv = s.smooth[48 + ((lane + 1) & 15)] + 1e-30f;
        wave_sync();
      }
      if (pv == 123.456f) s.smooth[40] = pv;
    }
#endif
    if (emode == 3) {   // set_obj_xyz (mujoco.py:217-227): object back to the pinned pose, all free-body velocities zeroed
      if (lane >= 9 && lane < 16) { s.qpos[lane] = pinv; s.qpos_lo[lane] = 0.f; }
      if (lane >= 9 && lane < nv) { s.qvel[lane] = 0.f; s.qvel_lo[lane] = 0.f; }
      wave_sync();
    }
    if (emode == 6) {   // the loops' exit tests, on the poses of this substep's forward pass (what mjData holds after sim.step)
      v3 pe; m3 Re;
      ee_frame(m, s, &pe, &Re);
      const v3 og = ld3(s.task + JT_OBJGOAL);
      const float dist = norm(pe - og);
      int nph = phase;
      bool hold = false;
      if (phase == 1) {
        float eul[3];
        mat_to_euler_rxyz(Re, eul);
        if (dist < 0.2f) { nph = 2; hold = true; }
        else if (grasp_ang_diff(eul, s.task + JT_REACHGOAL + 3) < 0.52359877559829887f) nph = 2;
      } else if (dist < 0.15f) nph = 3;
      wave_sync();
      if (lane == 0) {
        float* t = s.task;
        if (phase == 2 || !hold) { t[JT_TARGET] = og.x; t[JT_TARGET + 1] = og.y; t[JT_TARGET + 2] = og.z; }
        if (hold) { t[JT_TARGET] = pe.x; t[JT_TARGET + 1] = pe.y; t[JT_TARGET + 2] = pe.z; }
        t[JT_PHASE] = (float)nph;
      }
      wave_sync();
      if (nph == 3) { sub++; break; }   // (left stays 0: the observation follows below)
    }
    JSTAMP(8);
    // heavy tier: the burst that overflowed the light capacities is over (two substeps in a row would have fitted): give
    // the env back to the light code for the rest of the step (the caller alternates the two tiers)
    if (!LIGHT && handback && calm >= 2 && sub + 1 < nsub && emode != 2) {
      left = nsub - (sub + 1);
      flags |= JFLAG_TIER_RETURN;
      if (emode == 1 || emode == 3 || emode == 6) { if (lane == 0) { s.task[JT_PENDING] = 0.f; s.task[JT_SUB] = (float)(sub + 1); } wave_sync(); }
      break;
    }
  }
  Ap = args_view(A_);   // (the epilogue reads the argument block afresh: nothing of it was carried through the substep loop)
  bool reset_now = false, term_now = false;
  // Write-through stores for everything another workgroup may read or REWRITE before this launch set is over: an env that is handed over
  // (bailed), and -- option auto_reset -- every normal step's outputs, because a step that ends its episode is followed by the in-kernel
  // reset, whose forward pass may overflow the tier and be finished by a resident worker on another XCD: that workgroup rewrites the
  // observation, sensor, cache and state rows, and two XCDs holding dirty copies of one line write back in an undefined order (the
  // GPU determinism test of the policy-driven regime caught this as one env in 8 192 differing from run to run).
  const bool wt = bailed || (emode == 1 && A.auto_reset != 0 && !fwd);
  if (emode != 2) {
    if (wt) {   // handed over to another workgroup (possibly on another XCD): write-through stores
      if (lane < nq) { st_wt(&A.qpos[(size_t)env * nq + lane], s.qpos[lane]); if (A.qpos_lo) st_wt(&A.qpos_lo[(size_t)env * nq + lane], s.qpos_lo[lane]); }
      if (lane < nv) { st_wt(&A.qvel[(size_t)env * nv + lane], s.qvel[lane]); st_wt(&A.qacc_ws[(size_t)env * nv + lane], s.qacc_ws[lane]); if (A.qvel_lo) st_wt(&A.qvel_lo[(size_t)env * nv + lane], s.qvel_lo[lane]); }
    } else {
      if (lane < nq) { A.qpos[(size_t)env * nq + lane] = s.qpos[lane]; if (A.qpos_lo) A.qpos_lo[(size_t)env * nq + lane] = s.qpos_lo[lane]; }
      if (lane < nv) { A.qvel[(size_t)env * nv + lane] = s.qvel[lane]; A.qacc_ws[(size_t)env * nv + lane] = s.qacc_ws[lane]; if (A.qvel_lo) A.qvel_lo[(size_t)env * nv + lane] = s.qvel_lo[lane]; }
    }
  }
  if (left == 0 && lane < ns && A.sensordata && (emode < 4 || emode == 6)) { if (wt) st_wt(&A.sensordata[(size_t)env * ns + lane], sens); else A.sensordata[(size_t)env * ns + lane] = sens; }
  if (emode == 6 && left == 0 && s.task[JT_PHASE] < 3.f) flags |= JFLAG_PREREACH_CAP;
  if (emode == 5) sens = (lane < ns && A.sensordata) ? A.sensordata[(size_t)env * ns + lane] : 0.f;   // touch of the last forward pass
  if (emode) {
    if ((left == 0 && emode != 3 && (emode < 4 || emode == 6)) || (!LIGHT && left > 0 && !bailed && (emode == 1 || emode == 6))) {
      // what the controller reads one substep late, for the next launch -- or, on a heavy -> light hand-back in the middle
      // of a step, for the light code's next substep
      float* CW = A.cache + (size_t)env * JCACHE_N;
      int ob = m->obj_body >= 0 ? m->obj_body : 0;
      if (wt) {
        if (lane < 36) { st_wt(&CW[JC_M + lane], s.M[m_index(lane / 6, lane % 6)]); st_wt(&CW[JC_CDOF + lane], s.cdof[lane / 6][lane % 6]); }
        if (lane < 6) st_wt(&CW[JC_BIAS + lane], s.bias[lane]);
        if (lane < 3) { st_wt(&CW[JC_EEPOS + lane], s.xpos[m->ee_body][lane]); st_wt(&CW[JC_OBJPOS + lane], s.xpos[ob][lane]); }
        if (lane < 9) st_wt(&CW[JC_EEMAT + lane], s.xmat[m->ee_body][lane]);
      } else {
        if (lane < 36) { CW[JC_M + lane] = s.M[m_index(lane / 6, lane % 6)]; CW[JC_CDOF + lane] = s.cdof[lane / 6][lane % 6]; }
        if (lane < 6) CW[JC_BIAS + lane] = s.bias[lane];
        if (lane < 3) { CW[JC_EEPOS + lane] = s.xpos[m->ee_body][lane]; CW[JC_OBJPOS + lane] = s.xpos[ob][lane]; }
        if (lane < 9) CW[JC_EEMAT + lane] = s.xmat[m->ee_body][lane];
      }
    }
    if (left == 0 && emode != 3 && emode != 4) {
      // observation, reward, termination from the poses / sensors of the last forward pass (one substep stale, as in
      // the reference) -- make_observation, _get_reward, terminal_inspection (env_mujoco.py:122-126)
      int ob = m->obj_body >= 0 ? m->obj_body : 0;
      v3 pe; m3 Re;
      ee_frame(m, s, &pe, &Re);
      float eul[3];
      mat_to_euler_rxyz(Re, eul);
      v3 obj = ld3(s.xpos[ob]);
      v3 objgoal = ld3(s.task + JT_OBJGOAL), destgoal = ld3(s.task + JT_DESTGOAL);
      int touch = touch_class(sens, lane);
      float nz[6];
      unsigned cnt = rng_count(s.task[JT_RNG]);
      for (int k = 0; k < 6; k++) nz[k] = A.noise ? A.noise[(size_t)env * 12 + 6 + k] : rng_uniform(A.seed, (unsigned)env, cnt + k);
      float spos[3], sori[3];
      rulebased_subgoal(A.task_id, pe, objgoal, obj.y, destgoal, nz, spos, sori);
      float rew = 0.f, bonus = 0.f, wb = 0.f;
      int succ = 0;
      bool done = false;
      const float PI = 3.14159265358979323846f;
      if ((emode == 1 && !fwd) || emode == 5) {
        if (emode == 1) rew = A.task_id == JTASK_PICKING ? reward_picking(pe, eul, obj, touch)
                            : (A.task_id == JTASK_REACHING ? reward_reaching(pe, eul, s.task + JT_REACHGOAL, ld3(m->base_pos))
                            : (A.task_id == JTASK_GRASPING ? reward_picking(pe, eul, obj, touch, 0.05f) : 0.f));   // (placing, pickAndplace: 0 in the reference)
        float trow[4] = {0.f, s.task[JT_STEPS], s.task[JT_EPISODES], 0.f};
        float picked = s.task[JT_PICKED];
        const float objvel = nv >= 12 ? norm(ld3(&s.qvel[9])) : 0.f;   // |get_obj_vel()| (mujoco.py:212-215): the object's linear velocity, current state
        done = terminal_inspection(A.task_id, trow, s.qpos[2], pe, ld3(m->base_pos), obj, destgoal, touch, &bonus, &succ, &wb, eul, s.task + JT_REACHGOAL, &picked, objvel);
        // quarantine (failure row): a state that went non-finite ends the episode with no reward and stays
        // frozen until it is reset -- what MuJoCo's own bad-state check does with mj_resetData, made visible to the learner
        const bool bad = wave_ballot((flags & JFLAG_NAN) != 0u || !(rew == rew)) != 0ull;
        if (bad) { done = true; rew = 0.f; bonus = 0.f; succ = 0; flags |= JFLAG_NAN; }
        wave_sync();
        if (lane == 0) {
          s.task[JT_STEPS] = trow[JT_STEPS]; s.task[JT_EPISODES] = trow[JT_EPISODES]; s.task[JT_DONE] = done ? 1.f : 0.f;
          s.task[JT_SUCC] = (float)succ; s.task[JT_WB] = wb; s.task[JT_SUB] = 0.f; s.task[JT_PENDING] = 0.f; s.task[JT_PICKED] = picked;
          A.reward[env] = rew + bonus;
          A.done[env] = done ? 1 : 0;
          if (done && A.terminal) { A.terminal[2 * (size_t)env] = (float)succ; A.terminal[2 * (size_t)env + 1] = wb; }   // (survives the in-kernel reset)
        }
        term_now = emode == 1 && wave_ballot(done) != 0ull;
        reset_now = term_now && A.auto_reset != 0;
      }
      if (fwd && lane == 0) { s.task[JT_FWD] = 0.f; s.task[JT_SUB] = 0.f; s.task[JT_PENDING] = 0.f; }
      if (lane == 0 && emode != 5 && A.obs_mode == 0) s.task[JT_RNG] = rng_slot(cnt + 6u);
      if (lane < 26 && emode != 5) {
        float o;
        if (lane == 0) o = (float)touch;
        else if (lane < 4) o = lane == 1 ? pe.x : (lane == 2 ? pe.y : pe.z);
        else if (lane < 7) o = eul[lane - 4] / PI;
        else if (lane == 7) o = (s.task[JT_GRIP] - 0.8f) / 0.2f;
        else if (lane < 11) o = lane == 8 ? obj.x : (lane == 9 ? obj.y : obj.z);
        else if (lane < 14) o = 0.f;
        else if (lane < 17) o = s.task[JT_DESTGOAL + lane - 14];
        else if (lane < 20) o = A.obs_mode == 1 ? s.task[JT_REACHGOAL + lane - 17] : spos[lane - 17];
        else if (lane < 23) o = (A.obs_mode == 1 ? s.task[JT_REACHGOAL + lane - 17] : sori[lane - 20]) / PI;
        else o = lane == 24 ? PI / 2.f : 0.f;
        if (!(fabsf(o) <= 3.0e38f)) o = 0.f;   // (quarantined env: the observation row stays finite)
        if (wt) st_wt(&A.obs[(size_t)env * 26 + lane], o); else A.obs[(size_t)env * 26 + lane] = o;
        if (term_now && A.terminal_obs) A.terminal_obs[(size_t)env * 26 + lane] = o;   // latched with (success, wb) by every terminal step, whoever resets the env (what the learner's value bootstrap wants after a time-out)
      }
    }
    if (left == 0 && (emode == 3 || emode == 6) && lane == 0) { s.task[JT_SUB] = 0.f; s.task[JT_PENDING] = 0.f; }
    wave_sync();
    if (lane < JTASK_N) { if (wt) st_wt(&A.task[(size_t)env * JTASK_N + lane], s.task[lane]); else A.task[(size_t)env * JTASK_N + lane] = s.task[lane]; }
  }
  if (reset_now) {
    // auto-reset (option "auto_reset"; tasks whose reset is draws + sim.forward(): picking, reaching, pickAndplace): the episode has
    // ended and its reward / done flag are out; the same wave now does what jaco_reset(mask) would do in a launch chain of its own
    // (0.6 ms of mostly idle GPU per step; profiles/r03_trace_policy.txt is the step's timeline without it): sim.reset(), the draws (env_logic.h reset_draws: the code and
    // RNG stream of jaco_reset_kernel), then one more pass through the substep body as sim.forward() and the new episode's observation.
    wave_sync();
    if (lane < nq) { s.qpos[lane] = A.qpos0[lane]; s.qpos_lo[lane] = 0.f; }
    if (lane < nv) { s.qvel[lane] = 0.f; s.qvel_lo[lane] = 0.f; s.qacc_ws[lane] = 0.f; }
    if (lane < nu) s.ctrl[lane] = 0.f;
    if (lane < 24) {
      const float r = m->marker_rest[lane / 12][lane % 12];
      s.mk[lane] = r;
      if (A.marker) st_wt(&A.marker[(size_t)env * 24 + lane], r);   // (write-through: a bigger tier may pick the forward pass up)
    }
    wave_sync();
    if (lane == 0) {
      reset_draws(A.task_id, A.seed, (unsigned)env, nq >= 23, m->base_pos, s.qpos, s.task, GoalBuffer{A.goal_buf, A.goal_n, A.goal_stride});
      s.task[JT_FWD] = 1.f;
      if (A.hint) st_wt_i(&A.hint[env], 0);   // a reset env starts from scratch, as under jaco_reset (a bigger tier that finishes the forward pass sets it again)
    }
    wave_sync();
    fwd = true; sub0 = 0; nsub = 1; left = 0; calm = 0;
    goto again;
  }
  unsigned long long anyf = wave_ballot(flags != 0);
  if (anyf) {
    unsigned f = flags;
    for (int o = 1; o < 64; o <<= 1) f |= (unsigned)wave_shfl_i((int)f, lane ^ o);
    if (lane == 0 && A.flags) { if (bailed) or_wt(&A.flags[env], f); else A.flags[env] |= f; }
  }
  if (!LIGHT && left == 0 && nsub > 0 && lane == 0 && A.hint && A.hint_mode == 2) {   // the tier the state the next step starts from needs
    const int nc = s.ncon, ne = s.nefc, ns = s.nsphere;
    const bool light_rows = ne <= JacoLight::MAXEFC || (SideRows<JacoLDS<JacoLight>>::on && s.nside_cand > 0 && ne - s.nside_cand <= JacoLight::MAXEFC);
    A.hint[env] = (nc <= JacoLight::MAXCON && light_rows && ns <= JacoLight::MAXCAND) ? 0
                : ((nc <= JacoMedium::MAXCON && ne <= JacoMedium::MAXEFC && ns <= JacoMedium::MAXCAND) ? 1
                : ((nc <= JacoHeavy::MAXCON && ne <= JacoHeavy::MAXEFC && ns <= JacoHeavy::MAXCAND) ? 2 : 3));
  }
  if (left == 0 && lane == 0 && A.stats) {
    A.stats[4 * env] = s.ncon; A.stats[4 * env + 1] = s.nefc + s.nside; A.stats[4 * env + 2] = iters; A.stats[4 * env + 3] = s.ncand | (nls_last << 16);
  }
  wave_sync();
  if (why) *why = left <= 0 ? 0 : (bailed ? 1 : 2);
  return left;
#undef A
}

// The library is built from several translation units so that the long device compiles run side by side (__graft_entry__.build):
// kernels.hip is compiled once per kernel with -DJACO_TU=<n> (JACO_TU_HAS(n): this unit defines kernel n), jaco_env.hip (host side,
// -DJACO_TU=-1) defines none and launches them through the jaco_launch_* functions below.  Without JACO_TU (CPU emulator build)
// every kernel is defined.
#ifdef JACO_TU
#define JACO_TU_HAS(n) (JACO_TU == (n))
#else
#define JACO_TU_HAS(n) 1
#endif
// light tier: one workgroup (= one wavefront) per env
#ifndef JACO_LIGHT_WAVES
#define JACO_LIGHT_WAVES 3   // waves per SIMD the light kernel is compiled for: 13.3 KB of LDS per env allow 12 envs per CU, 168 VGPRs each
#endif
template <bool FULL, class C = JacoLight>
JDEV void light_grid(const JacoStepArgs& A, JacoLDS<C>& s) {
  const int lane = lane_id();
  if (env_id() >= A.nenv) return;
  const int nslots = A.nslots ? *A.nslots : A.nenv;
  for (int slot = env_id(); slot < nslots; slot += grid_size()) {   // (one pass, except for the small grid of a masked reset)
  const int env = A.order ? A.order[slot] : slot;
  // (envs whose previous step ended in a bigger tier were queued there before the launch: not this grid's, and not counted in light_left)
  if (A.routed_mark && A.routed_mark[env] == A.launch_id) return;
  const bool masked_out = FULL && (A.env_mode == 2 || A.env_mode == 3 || A.env_mode == 6) && A.mask && !A.mask[env];
  int left = 0;
  if (!masked_out) {
    if (FULL && A.hint && (A.env_mode == 2 || A.env_mode == 3 || A.env_mode == 6) && lane == 0) st_wt_i(&A.hint[env], 0);   // a reset env starts from scratch (its forward pass may raise it again)
    const unsigned long long t_start = wave_clock();
    left = run_env<C, 0, FULL>(A, s, env, A.nsub, lane);
    // (only real steps record their cost: the masked forward pass of a reset must not wipe the launch-order heuristic's input)
    if (lane == 0 && A.cost && A.env_mode <= 1) { unsigned c = (unsigned)((wave_clock() - t_start) >> 4); if (left > 0) st_wt_u(&A.cost[env], c); else A.cost[env] = c; }
    // hand-off: the env's state went to memory with write-through stores (run_env); once they are acknowledged the env
    // is appended to the medium tier's queue.  Its workgroups run concurrently (jaco_env.hip) and poll the queue.
    // (a reset's forward pass goes straight to the last tier: one more launch in the chain instead of three, each of which would
    // redo the narrowphase of a hand-inside-the-pedestal pose up to its own capacity)
    if (left > 0) queue_push(A, (FULL && A.env_mode == 2) ? 2 : 0, env, left, lane);
  }
  wave_sync();
  }
  if (lane == 0 && A.light_left) jaco_atomic_dec(A.light_left, false);
}
#if JACO_TU_HAS(0)
__global__ __launch_bounds__(64, JACO_LIGHT_WAVES) void jaco_physics_kernel(JacoStepArgs A) {   // modes 0 and 1 only
  __shared__ JacoLDS<JacoLight> s;
  JEMU_POISON(s);
  light_grid<false>(A, s);
}
#endif
// the full code under its own name for reset-time launches (forward pass, placing hold; for a masked reset a small grid that walks
// the list of reset envs): the step kernel's launch statistics (rocprofv3 --stats, bench.py's kernel_ms) then hold step launches only
#if JACO_TU_HAS(1)
__global__ __launch_bounds__(64, JACO_LIGHT_WAVES) void jaco_physics_kernel_listed(JacoStepArgs A) {   // every mode
  __shared__ JacoLDS<JacoLight> s;
  JEMU_POISON(s);
  light_grid<true>(A, s);
}
#endif
// The contact-free instantiation (modes 0 and 1): arm-only models / option disable_contact.  No collision, contact-row or touch code, LDS
// without the geom, candidate and contact arrays; 128 registers: BASELINE config 2's 4 096 envs are resident at once (16 per CU).  Nothing
// can overflow a capacity here (at most one limit row per joint), so the host launches no tier workers or drains next to it.
#if JACO_TU_HAS(8)
__global__ __launch_bounds__(64, 4) void jaco_physics_kernel_arm(JacoStepArgs A) {
  __shared__ JacoLDS<JacoArm> s;
  JEMU_POISON(s);
  light_grid<false, JacoArm>(A, s);
}
#endif
// One handed-over env on a bigger-tier workgroup: the big code (medium: TB = 1, heavy: TB = 2) runs while the overflow
// lasts, the light code in between, until the env's step is complete.  Returns 0, or -- medium only -- the substeps left
// when the env overflowed the medium capacities too (the caller passes it on to the heavy tier).
template <class BIG, int TB, class U>
JDEV int run_env_tiers(const JacoStepArgs& A, U& u, int env, int lane) {
  const unsigned long long t_start = wave_clock();
  const bool stepmode = A.env_mode == 1 || A.env_mode == 3 || A.env_mode == 6;
  int left = stepmode ? A.nsub : A.remaining[env], why = 0;
  for (;;) {
    left = run_env<BIG, TB>(A, u.big, env, stepmode ? A.nsub : left, lane, !A.no_tier_return, &why);
    if (left <= 0 || why == 1) break;
    wave_sync();
    left = run_env<JacoLight, 0>(A, u.light, env, stepmode ? A.nsub : left, lane);
    if (left <= 0) break;
    wave_sync();
  }
  if (lane == 0 && A.cost) A.cost[env] += (unsigned)((wave_clock() - t_start) >> 4);
  return left > 0 ? left : 0;
}
union JacoMediumLDS { JacoLDS<JacoMedium> big; JacoLDS<JacoLight> light; };
// medium tier: an env that outgrew it as well goes on the heavy tier's queue
JDEV void medium_env(const JacoStepArgs& A, JacoMediumLDS& u, int env, int lane) {
  int left = run_env_tiers<JacoMedium, 1>(A, u, env, lane);
  if (left > 0) queue_push(A, 1, env, left, lane);
}
// heavy tier: heavy code while the env needs more than the medium capacities, then medium <-> light as above.
// Returns 0 when the env's step is complete, or the substeps left when it outgrew the heavy capacities too (huge tier's turn).
union JacoAllLDS { JacoLDS<JacoHeavy> heavy; JacoMediumLDS ml; };
#ifndef JACO_HANDDOWN_MIN
#define JACO_HANDDOWN_MIN 8
#endif                      // substeps that must be left for a hand-down to pay (it costs a queue round trip and a fresh model staging)
JDEV int heavy_env_run(const JacoStepArgs& A, JacoAllLDS& u, int env, int lane, bool allow_down = false) {
  const bool stepmode = A.env_mode == 1 || A.env_mode == 3 || A.env_mode == 6;
  int left = stepmode ? A.nsub : A.remaining[env], why = 0;
  for (;;) {
    left = run_env<JacoHeavy, 2>(A, u.heavy, env, stepmode ? A.nsub : left, lane, !A.no_tier_return, &why);
    if (left <= 0) return 0;
    if (why == 1) return left;
    if (allow_down && A.handdown && A.env_mode == 1 && left >= JACO_HANDDOWN_MIN) return -left;   // calm again: the medium tier's turn (caller queues it)
    wave_sync();
    if (!stepmode && lane == 0) A.remaining[env] = left;   // (ctrl level: run_env_tiers reads the substeps left from here)
    wave_sync();
    left = run_env_tiers<JacoMedium, 1>(A, u.ml, env, lane);
    if (left <= 0) return 0;
    wave_sync();
    if (!stepmode && lane == 0) A.remaining[env] = left;
    wave_sync();
  }
}
JDEV void heavy_env(const JacoStepArgs& A, JacoAllLDS& u, int env, int lane) {
  const unsigned long long t_start = wave_clock();
  const int left = heavy_env_run(A, u, env, lane, true);
  if (lane == 0 && A.cost) A.cost[env] += (unsigned)((wave_clock() - t_start) >> 4);
  if (left > 0) queue_push(A, 2, env, left, lane);   // outgrew the heavy capacities too
  if (left < 0) queue_push(A, 0, env, -left, lane);  // handed down (the launch that follows this drain serves the medium queue again)
}
// huge tier: whatever outgrew the heavy tier (a reset with the hand inside the pedestal) runs here while that lasts, then steps
// down through heavy / medium / light like everything else
union JacoHugeLDS { JacoLDS<JacoHuge> huge; JacoAllLDS rest; };
JDEV void huge_env(const JacoStepArgs& A, JacoHugeLDS& u, int env, int lane) {
  const bool stepmode = A.env_mode == 1 || A.env_mode == 3 || A.env_mode == 6;
  const unsigned long long t_start = wave_clock();
  int left = stepmode ? A.nsub : A.remaining[env], why = 0;
  for (;;) {
    left = run_env<JacoHuge, 3>(A, u.huge, env, stepmode ? A.nsub : left, lane, !A.no_tier_return, &why);
    if (left <= 0) break;
    wave_sync();
    if (!stepmode && lane == 0) A.remaining[env] = left;
    wave_sync();
    left = heavy_env_run(A, u.rest, env, lane);
    if (left <= 0) break;
    wave_sync();
    if (!stepmode && lane == 0) A.remaining[env] = left;
    wave_sync();
  }
  if (lane == 0 && A.cost) A.cost[env] += (unsigned)((wave_clock() - t_start) >> 4);
}

// Resident workers of tier queue T (0 medium, 1 heavy, 2 huge), concurrent with the light grid: persistent workgroups claim
// queue slots.  A worker leaves as soon as the light grid has finished (whatever is still queued goes to the drain launch
// that follows in stream order, which brings a full grid instead of these few workgroups), or -- all but a small reserve --
// when the queue has run dry after the start-of-step routing.
#define JACO_WORKER_PATIENCE 6000
template <int T, class LDS>   // (a direct call: handing the serve function over as a lambda trips the gfx950 backend, "illegal VGPR to SGPR copy")
JDEV void tier_serve(const JacoStepArgs& A, LDS& u, int env, int lane) {
  if constexpr (T == 0) medium_env(A, u, env, lane);
  else if constexpr (T == 1) heavy_env(A, u, env, lane);
  else huge_env(A, u, env, lane);
}
template <int T, class LDS>
JDEV void tier_workers(const JacoStepArgs& A, LDS& u, int lane) {
  const JacoStepArgs::Queue& Q = A.q[T];
  const int me = env_id();
  if (Q.limit && me >= *Q.limit) return;
  const int reserve = Q.reserve ? *Q.reserve : 0x7fffffff;
  for (;;) {
    // claim a slot that a producer has already reserved (count > taken): nothing is ever claimed that may stay empty
    int i = -1, last_left = -1, idle = 0;
    for (;;) {
      // nobody stays once the light grid is done: whatever is still queued then belongs to the drain launches, which bring
      // full grids instead of these few workgroups (that is also what carries the load when most envs need a bigger tier)
      const int ll = wave_uniform_i(dev_load_relaxed(A.light_left));
      if (ll <= 0) return;
      int got = -1;
      if (lane == 0) {
        const int t = dev_load_relaxed(Q.taken), c = dev_load_relaxed(Q.count);
        if (t < c) got = atomicCAS(Q.taken, t, t + 1) == t ? t : -2;   // -2: another worker was faster, look again
      }
      got = wave_uniform_i(got);
      if (got >= 0) { i = got; break; }
      if (got == -2) continue;
      // the queue is dry.  What the ordering pass queued before the launch (the envs whose last step ended in a bigger tier) has
      // been handed out: only the reserve stays resident for the odd overflow later on -- an idle worker holds LDS the light
      // grid could use
      if (me >= reserve) return;
      idle = ll == last_left ? idle + 1 : 0;
      last_left = ll;
      if (idle > JACO_WORKER_PATIENCE) return;   // (safety: the light grid makes no progress, e.g. the launches were serialised)
      wave_sleep();
    }
    int env;
    for (;;) {   // the producer publishes the entry right after reserving the slot
      env = wave_uniform_i(dev_load_relaxed(&Q.list[i]));
      if (env >= 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
    dev_acquire();   // the env's state, written by the producing workgroup before it published the entry
    if (lane == 0) st_wt_i(&Q.list[i], -2);   // taken
    tier_serve<T>(A, u, env, lane);
    wave_sync();
  }
}
// drain of tier queue T: after the light grid and the workers have finished (stream order), serve whatever entry is still pending,
// i.e. the slots [taken, count).  Workgroups claim them one at a time (the envs' costs differ by an order of magnitude: a strided
// split leaves most of the grid idle behind the slowest stride); the grid is sized to the tier's full occupancy by the host.
template <int T, class LDS>
JDEV void tier_drain(const JacoStepArgs& A, LDS& u, int lane) {
  const JacoStepArgs::Queue& Q = A.q[T];
  const int count = *Q.count;   // (final: every producer of this queue has finished)
  for (;;) {
    int i = 0;
    if (lane == 0) i = jaco_atomic_inc(Q.taken);
    i = wave_uniform_i(wave_bcast_i(i, 0));
    if (i >= count) break;
    int env = Q.list[i];
    if (env < 0) continue;
    tier_serve<T>(A, u, env, lane);
    wave_sync();
  }
}
// The resident workers run with a raised wave priority (s_setprio): the envs they serve are the step's critical path -- a huge-tier
// env step (hand inside the pedestal after a reset: ~400 rows) takes as long as the whole light grid, and every issue slot it loses
// to the three light waves sharing its SIMD is added to the step's tail.
#if JACO_TU_HAS(2)
__global__ __launch_bounds__(64, 2) void jaco_physics_kernel_medium(JacoStepArgs A) {
  __builtin_amdgcn_s_setprio(1);
  __shared__ JacoMediumLDS u;
  JEMU_POISON(u);
  tier_workers<0>(A, u, lane_id());
}
#endif
#if JACO_TU_HAS(3)
__global__ __launch_bounds__(64, 2) void jaco_physics_kernel_medium_drain(JacoStepArgs A) {
  __shared__ JacoMediumLDS u;
  JEMU_POISON(u);
  tier_drain<0>(A, u, lane_id());
}
#endif
#if JACO_TU_HAS(4)
__global__ __launch_bounds__(64, JACO_HEAVY_WAVES) void jaco_physics_kernel_heavy_workers(JacoStepArgs A) {
  __builtin_amdgcn_s_setprio(2);
  __shared__ JacoAllLDS u;
  JEMU_POISON(u);
  tier_workers<1>(A, u, lane_id());
}
#endif
#if JACO_TU_HAS(5)
__global__ __launch_bounds__(64, JACO_HEAVY_WAVES) void jaco_physics_kernel_heavy_drain(JacoStepArgs A) {
  __shared__ JacoAllLDS u;
  JEMU_POISON(u);
  tier_drain<1>(A, u, lane_id());
}
#endif
#if JACO_TU_HAS(6)
__global__ __launch_bounds__(64) void jaco_physics_kernel_huge_workers(JacoStepArgs A) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ JacoHugeLDS u;
  JEMU_POISON(u);
  tier_workers<2>(A, u, lane_id());
}
#endif
#if JACO_TU_HAS(7)
__global__ __launch_bounds__(64) void jaco_physics_kernel_huge_drain(JacoStepArgs A) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ JacoHugeLDS u;
  JEMU_POISON(u);
  tier_drain<2>(A, u, lane_id());
}
#endif

// Launchers, one per kernel, defined next to it (kernels.hip) and called by the host side (jaco_env.hip).  k: 0 step kernel, 1 full-code
// twin (reset-time modes), 2 / 3 medium workers / drain, 4 / 5 heavy, 6 / 7 huge, 8 contact-free step kernel.
#ifndef JACO_EMULATED
void jaco_launch_kernel(int k, unsigned grid, hipStream_t st, const JacoStepArgs& A);
#define JACO_DEFINE_LAUNCHER(n, kernel) void jaco_launch_kernel_##n(unsigned grid, hipStream_t st, const JacoStepArgs& A) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, st, A); }
#if defined(JACO_TU) && JACO_TU == 0
JACO_DEFINE_LAUNCHER(0, jaco_physics_kernel)
#elif defined(JACO_TU) && JACO_TU == 1
JACO_DEFINE_LAUNCHER(1, jaco_physics_kernel_listed)
#elif defined(JACO_TU) && JACO_TU == 2
JACO_DEFINE_LAUNCHER(2, jaco_physics_kernel_medium)
#elif defined(JACO_TU) && JACO_TU == 3
JACO_DEFINE_LAUNCHER(3, jaco_physics_kernel_medium_drain)
#elif defined(JACO_TU) && JACO_TU == 4
JACO_DEFINE_LAUNCHER(4, jaco_physics_kernel_heavy_workers)
#elif defined(JACO_TU) && JACO_TU == 5
JACO_DEFINE_LAUNCHER(5, jaco_physics_kernel_heavy_drain)
#elif defined(JACO_TU) && JACO_TU == 6
JACO_DEFINE_LAUNCHER(6, jaco_physics_kernel_huge_workers)
#elif defined(JACO_TU) && JACO_TU == 7
JACO_DEFINE_LAUNCHER(7, jaco_physics_kernel_huge_drain)
#elif defined(JACO_TU) && JACO_TU == 8
JACO_DEFINE_LAUNCHER(8, jaco_physics_kernel_arm)
#endif
#endif
