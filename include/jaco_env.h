/* C ABI of the MI355X-native batched Jaco environment (libjaco_env.so).
 *
 * The reference has no FFI of its own: its only native boundary is mujoco-py's Cython layer
 * (mujoco_py.cymj), crossed at the call sites listed per entry point below (all paths relative to
 * /root/reference).  This library replaces what sits behind those calls, batched over num_envs
 * independent environments, one 64-lane wavefront per environment.
 *
 * Conventions: every function returns 0 on success and a negative JACO_E* code on failure, never
 * throws; jaco_last_error() gives the message.  All "*_dev" arguments are DEVICE pointers supplied and
 * owned by the caller (e.g. torch tensor .data_ptr()); the library owns only its model constants,
 * per-env state and scratch.  Calls are asynchronous on the given HIP stream (hipStream_t passed as
 * void*; NULL = default stream) and perform no hidden synchronisation unless stated.  One handle per
 * GPU; a handle is not thread-safe (the reference is single-threaded: main.py:27).
 *
 * Batched array layout: row-major [num_envs][n] fp32, i.e. the 64 lanes of the wavefront that owns
 * environment e read consecutive addresses of row e.
 */
#ifndef JACO_ENV_H
#define JACO_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JACO_OK 0
#define JACO_EINVAL (-1)   /* bad argument / model */
#define JACO_EHIP (-2)     /* HIP runtime error */
#define JACO_ENODEV (-3)   /* no usable gfx950 device */

/* per-env sticky flag bits (jaco_get_flags) */
#define JACO_FLAG_CON_OVERFLOW 1u   /* more contacts than the per-env contact buffer */
#define JACO_FLAG_EFC_OVERFLOW 2u   /* more constraint rows than the per-env row buffer */
#define JACO_FLAG_CAND_OVERFLOW 4u  /* more broadphase survivors than the candidate buffer */
#define JACO_FLAG_NAN 8u            /* non-finite velocity: env should be reset */
#define JACO_FLAG_SOLVER_MAXITER 16u
#define JACO_FLAG_HEAVY_TIER 32u     /* informational: stepped by a bigger tier (128 / 256 / 512 rows) at least once (not an error) */
#define JACO_FLAG_TIER_RETURN 128u   /* informational: the heavy tier gave the env back to the light code in mid-step (overflow was transient) */
#define JACO_FLAG_BAIL_CAUSE_SHIFT 8  /* informational, bits 8..16: which capacity (bit 0 contacts, 1 rows, 2 candidates) made tier 0 / 1 / 2 (3 bits each) hand the env on */
#define JACO_FLAG_OSC_SINGULAR 64u   /* informational: |det(J M^-1 J^T)| < 1e-3, the controller used its pseudo-inverse branch */
#define JACO_FLAG_PREREACH_CAP 0x20000u /* the grasping reset's pre-reach loops (unbounded in the reference) stopped at the substep cap */

/* task ids (env_script/env_mujoco.py:18-23; only picking/placing return the 4-tuple step() unpacks) */
#define JACO_TASK_PICKING 0
#define JACO_TASK_PLACING 1
#define JACO_TASK_REACHING 2
#define JACO_TASK_GRASPING 3       /* reward env_mujoco_util.py:352-391, termination :521-536 (+ the success flag its 3-tuple lacks), reset pre-reach :123-170 */
#define JACO_TASK_PICKANDPLACE 4   /* reward 0, termination :585-600 with the `picked` flag, 1200-step episodes */
#define JACO_TASK_CARRYING 5       /* reset = in-hand hold :106-117 + pre-reach :123-170; reward 0; every episode ends in its first step (:549-550) */
#define JACO_TASK_RELEASING 6      /* own init pose :186-189, in-hand hold :106-117; reward 0; termination :551-566 (+ success flag) */
#define JACO_TASK_PUSHING 7        /* 6-wide action (env_mujoco.py:79-82); reward 0; every episode ends in its first step (:583-584) */

typedef struct JacoHandle JacoHandle;

typedef struct JacoConfig {
  const void* model_blob;      /* JACOMDL1 bytes (host memory), see mujoco_jaco_amd/modelc */
  size_t model_blob_size;
  int num_envs;
  int device;                  /* HIP device ordinal */
  int frame_skip;              /* physics substeps per env step; reference: 50 (env_mujoco.py:24) */
  int task;                    /* JACO_TASK_* */
  uint64_t seed;               /* counter-based RNG seed for reset / sub-goal noise */
} JacoConfig;

/* JacoMujocoEnvUtil.__init__ -> MujocoConfig(xml) + Mujoco.connect()  (env_mujoco_util.py:28-33,
 * mujoco_config.py:74 mjp.load_model_from_path, mujoco.py:55-56 MjSim + forward). */
int jaco_create(const JacoConfig* cfg, JacoHandle** out);
int jaco_destroy(JacoHandle* h);
/* Message of the last failure on this handle; pass NULL for a failure of jaco_create itself. */
const char* jaco_last_error(const JacoHandle* h);

/* model.nq / nv / nu, len(sensordata); observation and action widths (env_mujoco.py:51-63,79-89). */
int jaco_dims(const JacoHandle* h, int* nq, int* nv, int* nu, int* nsensor, int* nobs, int* nact);
int jaco_num_envs(const JacoHandle* h);

/* sim.get_state()/set_state() + sim.data.qpos/qvel writes (mujoco.py:213-246,332-347).
 * Any pointer may be NULL to skip that field.  qacc_warmstart is part of the state because the
 * constraint solver is warm-started from it, as in MuJoCo. Device-to-device copies on `stream`.
 * Precision: the library carries qpos / qvel as compensated pairs of floats (hi + lo, ~48 significant bits: the integrators
 * advance the pair, every other computation reads the fp32 value `hi`; option "compensated" = 0 turns that off).  get_state
 * returns `hi`, the fp32 rounding of the state; set_state writes `hi` and clears `lo` (the state becomes exactly the floats
 * handed in), so a get -> set round trip rounds the state to fp32 once. */
int jaco_set_state(JacoHandle* h, const float* qpos_dev, const float* qvel_dev, const float* qacc_ws_dev, void* stream);
int jaco_get_state(JacoHandle* h, float* qpos_dev, float* qvel_dev, float* qacc_ws_dev, void* stream);
/* sim.reset() for every env: qpos0, zero velocities (env_mujoco_util.py:93). */
int jaco_reset_state(JacoHandle* h, void* stream);

/* Mujoco.send_forces(u): sim.data.ctrl[:] = u; sim.step()  (mujoco.py:258-278), `nsub` times with
 * the same ctrl, for all envs.  ctrl_dev: [num_envs][nu] fp32.  This is the ctrl-level entry used
 * for oracle parity and the physics benchmark. */
int jaco_physics_step(JacoHandle* h, const float* ctrl_dev, int nsub, void* stream);

/* sim.data.get_sensor(...) for the 20 touch sensors (env_mujoco_util.py:470-475): values computed
 * by the last substep, sensordata order of the XML (EE_touch first). out_dev: [num_envs][nsensor]. */
int jaco_get_sensordata(JacoHandle* h, float* out_dev, void* stream);

/* Per-env sticky error bits / last-substep statistics [num_envs][4] = {ncon, nefc, solver iterations,
 * narrowphase candidates}. */
int jaco_get_flags(JacoHandle* h, uint32_t* out_dev, void* stream);
int jaco_clear_flags(JacoHandle* h, void* stream);
int jaco_get_stats(JacoHandle* h, int32_t* out_dev, void* stream);

/* ---- env level: JacoMujocoEnv.reset / step (env_script/env_mujoco.py:99-139), batched -------------------------
 * jaco_reset: _reset (env_mujoco_util.py:92-174) for the envs whose mask byte is non-zero (NULL = all): counter-based
 *   RNG draws of the per-task initial state (Appendix A of SURVEY.md), sim.forward(), then _get_observation for those
 *   envs into their rows of obs_dev [num_envs][26] (rows of the other envs are left as they are).  Task `placing` runs jaco_placing_hold(mask, 150) between the draws and the observation.
 * jaco_placing_hold: the object part of the placing reset (env_mujoco_util.py:106-117): object to the grasp frame EE_obj
 *   (4 cm back along its x axis), EE target = current EE pose, then nsub x { OSC torque from the current state's M, J, bias
 *   (each iteration follows a sim.forward()), sim.step() with gripper command 0.6, set_obj_xyz: object re-pinned, velocities
 *   of all free bodies zeroed (mujoco.py:217-227) }.  Exposed so that a caller can replay the hold from its own state.
 * jaco_step: np.clip of the action to [-1, 1] (env_mujoco.py:117) is done inside the kernel; then _take_action, frame_skip x
 *   (OSC torque + sim.step()), make_observation, _get_reward, terminal_inspection.  action_dev [num_envs][7] (6 for reaching),
 *   obs_dev [num_envs][26] f32, reward_dev [num_envs] f32, done_dev [num_envs] u8.  By default an env that returned done stays
 *   frozen (done = 1, reward 0, obs row untouched) until jaco_reset is called for it.  With jaco_set_option("auto_reset", 1)
 *   (tasks whose reset is draws + sim.forward(): picking, reaching, pickAndplace, pushing) the wave that ends an episode resets the
 *   env itself: done / reward are the terminal step's, the obs row is the NEW episode's first observation, and the terminal
 *   step's (success, wb) / observation are latched for jaco_get_last_terminal / jaco_get_terminal_obs.
 * jaco_forward: sim.forward() + _get_observation from the current state (after jaco_set_state / jaco_set_task_state).
 * jaco_set_noise: optional [num_envs][12] uniform draws replacing the internal RNG for the rule-based sub-goal noise
 *   (6 for the marker placed in _take_action, 6 for the observation; env_mujoco_util.py:279,295); NULL restores the RNG.
 * Task rows ([num_envs][jaco_task_row_floats()], layout JT_* in csrc/env_logic.h) expose gripper command, step
 *   counters, goals and the success flag (get_wb / accum_succ bookkeeping stay on the host side). */
int jaco_reset(JacoHandle* h, const uint8_t* mask_dev, float* obs_dev, void* stream);
int jaco_placing_hold(JacoHandle* h, const uint8_t* mask_dev, int nsub, void* stream);
/* The pre-reach part of the grasping reset (env_mujoco_util.py:123-170), run by jaco_reset for task grasping after the draws: EE target =
 * [object goal, orientation looking along EE -> object (float16 angles, yaw drawn)]; loop 1 { stop_obj, controller + sim.step } until the
 * EE is within 0.2 m of the object goal or its orientation within pi/6 of the sampled reaching goal's; loop 2 { controller + sim.step }
 * until within 0.15 m; then _get_observation into the masked rows of obs_dev.  The reference's loops are unbounded: max_substeps caps
 * them (jaco_reset uses 4000; an env that hits the cap gets JACO_FLAG_PREREACH_CAP). */
int jaco_grasping_prereach(JacoHandle* h, const uint8_t* mask_dev, int max_substeps, float* obs_dev, void* stream);
int jaco_step(JacoHandle* h, const float* action_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream);
int jaco_forward(JacoHandle* h, float* obs_dev, void* stream);
/* The two halves of jaco_step the reference also exposes as public methods of JacoMujocoEnv (env_mujoco.py:144-161):
 * jaco_take_action: take_action(a) alone -- new EE target, gripper command / ramp end points, the two marker poses, 6 noise
 *   draws; no physics.  A following jaco_step would call _take_action again, as the reference's step() does.
 * jaco_terminal_inspection: terminal_inspection() alone -- current_steps += 1, then the task's termination rule on the
 *   poses / touch sensors of the last forward pass; done_dev [num_envs] u8, bonus_dev [num_envs] f32 (the reference's
 *   additional_reward); get_wb / the success flag are then in the task row (JT_WB, JT_SUCC).  Finished envs freeze as in jaco_step. */
int jaco_take_action(JacoHandle* h, const float* action_dev, void* stream);
int jaco_terminal_inspection(JacoHandle* h, uint8_t* done_dev, float* bonus_dev, void* stream);
/* Lifetime: jaco_set_noise / jaco_set_subgoal only record the pointer in the handle; the buffer is read by every later jaco_step /
 * jaco_take_action / jaco_forward / jaco_reset on whatever stream that call uses.  It must stay allocated and unchanged until
 * those calls have completed on the device or the pointer has been replaced (NULL detaches it). */
int jaco_set_noise(JacoHandle* h, const float* noise_dev);
/* Observation branch rulebased_subgoal = False (env_mujoco_util.py:255-270; option "obs_mode" = 1): obs[17:23] is the reaching goal
 * drawn at reset, and _take_action moves the "subgoal_reach" marker to subgoal + previous target (:609) when the caller passes the
 * policy's sub-goal offsets [num_envs][6] here (NULL: the marker stays where it is). */
int jaco_set_subgoal(JacoHandle* h, const float* subgoal_dev);
/* kwarg init_buffer of the reference (env_mujoco_util.py:46,208-212): a buffer of recorded rows from which every reset draws its reaching
 * goal -- random_idx = np.random.randint(0, len(buffer) - 1) (rows 0 .. len - 2), position = row[1:4], orientation = row[4:7], no float16
 * cast -- instead of sampling it (:199-207).  rows_dev [nrows][row_floats] f32 on the device, copied by the library; NULL restores the
 * sampled goal.  nrows >= 2 (numpy's randint(0, 0) raises), row_floats >= 7. */
int jaco_set_init_buffer(JacoHandle* h, const float* rows_dev, int nrows, int row_floats, void* stream);
int jaco_get_task_state(JacoHandle* h, float* out_dev, void* stream);
int jaco_set_task_state(JacoHandle* h, const float* in_dev, void* stream);
int jaco_task_row_floats(void);
/* What the most recent TERMINAL step of every env returned besides (reward, done): out_dev [num_envs][2] f32 = (success flag, wb) -- the
 * `succ` and `wb` of terminal_inspection (env_mujoco.py:125,144-150), latched by the step that ended the episode.  Needed with option
 * "auto_reset", where the task row already belongs to the new episode when jaco_step returns (the reference's accum_succ bookkeeping,
 * env_mujoco.py:129-136, reads succ in the terminal step).  Rows of envs that have not finished an episode yet are (0, 0). */
int jaco_get_last_terminal(JacoHandle* h, float* out_dev, void* stream);
/* ... and the observation of that terminal step, out_dev [num_envs][26] f32, latched by EVERY terminal jaco_step with or without option
 * "auto_reset" (with it the env's obs_dev row already holds the new episode's first observation; a learner bootstrapping the value of a
 * timed-out state needs the last one of the old episode).  Rows of envs that have not finished an episode yet are zero. */
int jaco_get_terminal_obs(JacoHandle* h, float* out_dev, void* stream);
/* Marker poses [num_envs][2][12] f32: {"hand", "subgoal_reach"} x {position, rotation matrix row-major} -- the two mocap bodies
 * _take_action moves every env step (set_mocap_xyz / set_mocap_orientation, mujoco.py:248-256, env_mujoco_util.py:613-615,
 * 644-646).  Their contype-8 geoms collide with the EE axis sticks; jaco_step updates them in-kernel, jaco_reset* park
 * them at their XML pose. */
int jaco_get_markers(JacoHandle* h, float* out_dev, void* stream);
int jaco_set_markers(JacoHandle* h, const float* in_dev, void* stream);
int jaco_set_frame_skip(JacoHandle* h, int frame_skip);

/* Solver / collision options, MuJoCo <option> names: "iterations", "tolerance", "ls_iterations",
 * "disable_contact" (contact flag), "mpr_iterations", "mpr_tolerance", "mpr_output"; "compensated" (1 default, see jaco_set_state).
 * "auto_reset" (0 default): jaco_step resets an env whose step ended its episode inside the same call -- sim.reset(), the draws of _reset
 * (the code and RNG stream of jaco_reset), sim.forward() -- in the wavefront that finished it: reward_dev / done_dev carry the terminal
 * step's values, the env's obs_dev row and task row are the new episode's first.  Bit-identical to jaco_step followed by
 * jaco_reset(done mask); honoured for the tasks whose reset is draws + forward pass (picking, reaching, pickAndplace).
 * Setting a model option (everything in this first group except "disable_contact") SYNCHRONISES the device before the model
 * constants are re-uploaded: it is the one entry point besides the *_debug / *_time_ms hooks that does.
 * Execution options ("schedule", "concurrent_heavy", "heavy_workers", "handdown", "merge_prepare": bit-identical results; "hints" / "tier_return" pick which
 * capacity tier's code steps a substep, and the tiers group their row sums differently: results agree to fp32 rounding): "schedule" (1: launch expensive envs first), "concurrent_heavy" (1: medium / heavy / huge
 * tier workgroups resident next to the light grid), "heavy_workers" (maximum of the medium tier's; the resident number follows the
 * previous step's hand-overs), "tier_return" (1: a bigger tier gives an env back once its overflow is over), "hints" (where an env
 * starts its next step: 0 always the light tier, 1 the biggest tier its last step needed, 2 (default) the tier its last substep
 * needed), "handdown" (1, default: the first heavy drain passes calmed-down envs to a second medium drain instead of keeping them
 * for the rest of the step), "merge_prepare" (1, default: a step's routing kernel also prepares the tier queues of the NEXT launch --
 * the queue state is double-buffered -- so that steps in a row start with one small kernel; 0: every launch prepares its own queues
 * with a kernel of its own; bit-identical results). */
int jaco_set_option(JacoHandle* h, const char* name, double value);

/* Test hook: like jaco_physics_step but also copies the stage dump of environment `env` taken in
 * the last substep (layout: JDBG_* in csrc/physics_kernel.h) to host memory; synchronises. */
int jaco_physics_step_debug(JacoHandle* h, const float* ctrl_dev, int nsub, int env, float* dump_host, int dump_floats);
int jaco_debug_dump_floats(void);
/* Kernel launches issued for this handle since the previous call of this function (host-side counter, no device work):
 * bench.py's launches_per_step. */
long long jaco_launch_count(JacoHandle* h);
/* Diagnostic: control words of the tier queues after the last launch (per tier: envs queued, slots claimed by resident workers,
 * workers started / kept in reserve, envs queued before the launch from last step's hints); returns the number of words. */
int jaco_debug_queue_words(JacoHandle* h, int32_t* out_host, int n);

/* Average device time of the light-tier kernel (jaco_physics_kernel: the dominant kernel, what rocprofv3 --stats lists under that
 * name) over the step launches since jaco_enable_timing, measured with HIP events on the stream the kernel was launched on (bench.py
 * roofline leg); synchronises. */
int jaco_kernel_time_ms(JacoHandle* h, double* avg_ms, int* launches);
/* Same window, the whole launch set of a step (routing and ordering passes, light grid, the tiers' workers and drains): call it
 * BEFORE jaco_kernel_time_ms, which closes the window. */
int jaco_step_time_ms(JacoHandle* h, double* avg_ms);
int jaco_enable_timing(JacoHandle* h, int enable);

/* Diagnostic builds only (-DJACO_PROFILE_STAGES): per-env, per-stage shader-clock sums [num_envs][12] copied to host;
 * the shipped library returns JACO_EINVAL. */
int jaco_stage_profile(JacoHandle* h, uint64_t* out_host, int reset);

#ifdef __cplusplus
}
#endif
#endif /* JACO_ENV_H */
