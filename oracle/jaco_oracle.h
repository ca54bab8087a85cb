/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 *
 * fp64 CPU restatement ("oracle") of what the reference's physics step computes:
 *   Mujoco.send_forces -> sim.step()   (/root/reference/env_script/mujoco.py:258-278)
 * i.e. MuJoCo's mj_step under the defaults the reference's XML leaves in force
 * (SURVEY.md section 8 row a6, Appendix D.1).  The arithmetic itself lives in the closed
 * MuJoCo 2.0 binary reached through an un-pinned mujoco-py fork (README.md:15-24 of the
 * reference), which is absent from /root/reference and from this image, and the reference
 * has no tests.  PINNED (round 5) against the only MuJoCo-produced numbers the reference holds:
 * the object-height transients of its recorded trajectories (the .npz files under models_baseline/trajectories,
 * obs[:, 10]) -- ten float32 values of the object falling onto the floor, thirteen of the holder
 * pushing it out of its spawn overlap, both rest heights -- reproduced BIT FOR BIT
 * (tests/test_mujoco_statics.py, tests/golden/mujoco_rest_heights.json): plane-box and box-box
 * contact generation, solref / solimp mixing, impedance, regulariser, pyramidal cone, Newton
 * optimum, semi-implicit Euler.  The articulated-arm dynamics, the hull contacts (MPR) and the
 * controller are NOT pinned against MuJoCo ("parity unpinned" for those): they restate the
 * published computation pipeline [EXT] and are held by first principles and by the success
 * rates of the reference's shipped policies.  Each function cites the reference call site
 * whose result it stands for.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (mujoco_jaco_amd/) never does.
 */
#ifndef JACO_ORACLE_H
#define JACO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcModel OrcModel;
typedef struct OrcData OrcData;

#define ORC_MAXCON 256
#define ORC_MAXEFC 1024

/* solver selection for orc_set_option("solver", v) */
#define ORC_SOLVER_PGS 0     /* dual projected Gauss-Seidel (cross-check of the same QP) */
#define ORC_SOLVER_NEWTON 1  /* primal Newton + exact line search: MuJoCo's default, used for parity */

OrcModel* orc_load_model(const char* path);           /* JACOMDL1 blob written by modelc */
void orc_free_model(OrcModel* m);
OrcData* orc_make_data(const OrcModel* m);
void orc_free_data(OrcData* d);
void orc_reset(const OrcModel* m, OrcData* d);        /* sim.reset(): qpos0, zero vel/ctrl, XML mocap poses */

int orc_model_int(const OrcModel* m, const char* name);               /* nq nv nu nbody ngeom nmocap nsensor ... */
int orc_set_option(OrcModel* m, const char* name, double value);     /* timestep, iterations, tolerance, disable_contact, solver, ls_* */

/* state access; name in {qpos,qvel,ctrl,qacc_warmstart,mocap_pos,mocap_quat} for set;
 * any computed field for get (xpos,xquat,qM,qfrc_bias,qacc,sensordata,efc_force,contact_*, ...). */
int orc_set(const OrcModel* m, OrcData* d, const char* name, const double* src, int n);
int orc_get(const OrcModel* m, const OrcData* d, const char* name, double* dst, int n);
int orc_ncon(const OrcData* d);
int orc_nefc(const OrcData* d);
int orc_solver_iter(const OrcData* d);

void orc_forward(const OrcModel* m, OrcData* d);      /* sim.forward()  (mujoco.py:56,227,347) */
void orc_step(const OrcModel* m, OrcData* d);         /* sim.step()     (mujoco.py:278) */

/* mj_jacBodyCom / mj_fullM stand-ins used by the controller (mujoco_config.py:269,320) */
void orc_jac_body_com(const OrcModel* m, const OrcData* d, int body, double* jacp, double* jacr);

/* batched convenience for timing / parity: nenv independent envs stored env-major
 * (qpos[nenv][nq], ...), nsub substeps each with constant ctrl; threads via OpenMP if built with it. */
void orc_step_batch(const OrcModel* m, int nenv, int nsub, double* qpos, double* qvel, double* qacc_ws,
                    const double* ctrl, double* sensordata, int nthreads);
/* same + per-env stats[4]: max contacts, max rows, max solver iterations over the nsub steps, contacts of the last step (or NULL) */
void orc_step_batch_stats(const OrcModel* m, int nenv, int nsub, double* qpos, double* qvel, double* qacc_ws,
                          const double* ctrl, double* sensordata, int nthreads, int* stats);

#ifdef __cplusplus
}
#endif
#endif
