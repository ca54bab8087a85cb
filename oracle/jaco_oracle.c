/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.  See jaco_oracle.h for scope, citations and the
 * parity statement (pinned to MuJoCo's recorded object transients for the contact physics of the free bodies; arm dynamics, hull contacts and
 * the controller unpinned).  Plain C99, fp64, scalar; generic over the raw (unfused)
 * model arrays so that it shares no structure with the HIP path it checks.
 *
 * Pipeline restated (SURVEY.md App. D.1 numbering, all [EXT] = published MuJoCo algorithm):
 *   1 kinematics   2 CRBA + factor   3 collision   4-5 constraint rows, impedance, aref, R
 *   6 RNE bias, passive, actuation   7 dual PGS solve   8 touch sensors   9 Euler (implicit damping)
 */
#include "jaco_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999
enum { G_PLANE = 0, G_SPHERE = 2, G_CYLINDER = 5, G_BOX = 6, G_MESH = 7 };
enum { J_FREE = 0, J_HINGE = 3 };

struct OrcModel {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nmesh, nmocap, nsensor, npair;
  double timestep, gravity[3], tolerance, mpr_tolerance, meaninertia;
  int iterations, mpr_iterations, disable_contact, solver, ls_iterations;
  int mpr_output;  /* 0: libccd's closest point of the final portal triangle; 1 (default): portal plane (see mpr_penetration) */
  int round_state; /* control experiments: 1 = qpos/qvel/qacc_warmstart are rounded to fp32 after every step (an fp64 engine carrying fp32 state);
                      2 = fp64 state, but the forward pass sees its fp32 rounding (an engine that carries its state compensated) */
  double ls_tolerance;
  double boxbox_depth_scale; /* dist of a box-box face contact = this x the geometric overlap.  0.5 is what the reference's MuJoCo 2.0 did: pinned by the
                                thirteen-value push-out transient and the rest height 0.19997096 its recorded trajectories hold (tests/golden/mujoco_rest_heights.json,
                                tools/rest_height_sweep.py, profiles/r05_rest_height_holder.txt); 1 (the geometric overlap) rests 1.45e-5 m too high */
  int *body_parentid, *body_weldid, *body_mocapid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  double *body_pos, *body_quat, *body_ipos, *body_inertia, *body_mass, *body_invweight0;
  int *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
  double *jnt_pos, *jnt_axis, *jnt_range, *jnt_solref, *jnt_solimp, *qpos0;
  double *jnt_stiffness, *jnt_springref;   /* optional (NULL in blobs compiled before round 3): joint springs of the sibling MJCFs */
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  double *dof_damping, *dof_invweight0;
  int *geom_type, *geom_bodyid, *geom_dataid, *geom_contype, *geom_conaffinity, *geom_condim;
  double *geom_pos, *geom_quat, *geom_size, *geom_rbound, *geom_friction, *geom_solref, *geom_solimp, *geom_margin;
  int *mesh_vertadr, *mesh_vertnum;
  double* mesh_vert;
  int *site_bodyid, *site_type;
  double *site_pos, *site_quat, *site_size;
  int *actuator_jntid, *actuator_position, *actuator_ctrllimited, *actuator_forcelimited;
  double *actuator_kp, *actuator_ctrlrange, *actuator_forcerange;
  int *sensor_siteid, *pair_geom;
  double *mocap_pos0, *mocap_quat0;
  char* blob;
};

typedef struct {
  double dist, pos[3], frame[9], mu[5], solref[2], solimp[5], margin;
  int geom1, geom2, dim, efc_address;
} Contact;

struct OrcData {
  double *qpos, *qvel, *ctrl, *qacc_warmstart, *mocap_pos, *mocap_quat;
  double *xpos, *xquat, *xmat, *xipos, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  double *cdof, *cinert, *crb, *cvel, *cacc, *cfrc, *cdof_dot;
  double *qM, *qL, *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
  double *actuator_force, *sensordata;
  int ncon, nefc, solver_iter;
  double* scratch; long scratch_cap;   /* per-step work space of make_constraint / the solvers / integrate (one user at a time): no malloc per step */
  Contact* contact;
  double *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_aref, *efc_b, *efc_force, *efc_vel, *efc_MinvJT, *efc_AR;
  int *efc_type, *efc_id; /* type 0 = limit (id = joint), 1 = contact (id = contact index) */
  int efc_cap;
};

/* ------------------------------------------------------------------ small math */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void copy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void sub3(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void add3(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void addscl3(double* r, const double* a, const double* b, double s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static inline void scl3(double* r, const double* a, double s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
/* row-major 3x3: r = M v ; r = M^T v */
static inline void mulmv(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2], z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmtv(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2], y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2], z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulmm(double* r, const double* A, const double* B) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(r, t, sizeof t);
}
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; } /* zero quaternion -> identity (env_mujoco_util.py:119-121 relies on it) */
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat2mat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = w * w + x * x - y * y - z * z; M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = w * w - x * x + y * y - z * z; M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = w * w - x * x - y * y + z * z;
}
static void rotvecq(double* r, const double* q, const double* v) {
  double M[9];
  quat2mat(M, q);
  mulmv(r, M, v);
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(angle / 2);
  q[0] = cos(angle / 2); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ------------------------------------------------------------------ model loading */
typedef struct { const char* name; void* dst; int is_int; } Field;

OrcModel* orc_load_model(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* buf = (char*)malloc(sz);
  if (fread(buf, 1, sz, f) != (size_t)sz || memcmp(buf, "JACOMDL1", 8)) { fclose(f); free(buf); return NULL; }
  fclose(f);
  OrcModel* m = (OrcModel*)calloc(1, sizeof(OrcModel));
  m->blob = buf;
  int *nq = 0, *nv = 0, *nu = 0, *nbody = 0, *njnt = 0, *ngeom = 0, *nsite = 0, *nmesh = 0, *nmocap = 0, *nsensor = 0, *npair = 0, *iters = 0, *mpri = 0;
  double *ts = 0, *grav = 0, *tol = 0, *mprt = 0, *meani = 0;
  Field fields[] = {
      {"nq", &nq, 1}, {"nv", &nv, 1}, {"nu", &nu, 1}, {"nbody", &nbody, 1}, {"njnt", &njnt, 1}, {"ngeom", &ngeom, 1},
      {"nsite", &nsite, 1}, {"nmesh", &nmesh, 1}, {"nmocap", &nmocap, 1}, {"nsensor", &nsensor, 1}, {"npair", &npair, 1},
      {"opt_timestep", &ts, 0}, {"opt_gravity", &grav, 0}, {"opt_tolerance", &tol, 0}, {"opt_iterations", &iters, 1},
      {"opt_mpr_tolerance", &mprt, 0}, {"opt_mpr_iterations", &mpri, 1}, {"meaninertia", &meani, 0},
      {"body_parentid", &m->body_parentid, 1}, {"body_weldid", &m->body_weldid, 1}, {"body_mocapid", &m->body_mocapid, 1},
      {"body_jntadr", &m->body_jntadr, 1}, {"body_jntnum", &m->body_jntnum, 1}, {"body_dofadr", &m->body_dofadr, 1},
      {"body_dofnum", &m->body_dofnum, 1}, {"body_pos", &m->body_pos, 0}, {"body_quat", &m->body_quat, 0},
      {"body_ipos", &m->body_ipos, 0}, {"body_inertia", &m->body_inertia, 0}, {"body_mass", &m->body_mass, 0},
      {"body_invweight0", &m->body_invweight0, 0}, {"jnt_type", &m->jnt_type, 1}, {"jnt_bodyid", &m->jnt_bodyid, 1},
      {"jnt_qposadr", &m->jnt_qposadr, 1}, {"jnt_dofadr", &m->jnt_dofadr, 1}, {"jnt_limited", &m->jnt_limited, 1},
      {"jnt_pos", &m->jnt_pos, 0}, {"jnt_axis", &m->jnt_axis, 0}, {"jnt_range", &m->jnt_range, 0},
      {"jnt_solref", &m->jnt_solref, 0}, {"jnt_solimp", &m->jnt_solimp, 0}, {"qpos0", &m->qpos0, 0},
      {"jnt_stiffness", &m->jnt_stiffness, 0}, {"jnt_springref", &m->jnt_springref, 0},
      {"dof_bodyid", &m->dof_bodyid, 1}, {"dof_jntid", &m->dof_jntid, 1}, {"dof_parentid", &m->dof_parentid, 1},
      {"dof_damping", &m->dof_damping, 0}, {"dof_invweight0", &m->dof_invweight0, 0},
      {"geom_type", &m->geom_type, 1}, {"geom_bodyid", &m->geom_bodyid, 1}, {"geom_dataid", &m->geom_dataid, 1},
      {"geom_contype", &m->geom_contype, 1}, {"geom_conaffinity", &m->geom_conaffinity, 1}, {"geom_condim", &m->geom_condim, 1},
      {"geom_pos", &m->geom_pos, 0}, {"geom_quat", &m->geom_quat, 0}, {"geom_size", &m->geom_size, 0},
      {"geom_rbound", &m->geom_rbound, 0}, {"geom_friction", &m->geom_friction, 0}, {"geom_solref", &m->geom_solref, 0},
      {"geom_solimp", &m->geom_solimp, 0}, {"geom_margin", &m->geom_margin, 0},
      {"mesh_vertadr", &m->mesh_vertadr, 1}, {"mesh_vertnum", &m->mesh_vertnum, 1}, {"mesh_vert", &m->mesh_vert, 0},
      {"site_bodyid", &m->site_bodyid, 1}, {"site_type", &m->site_type, 1}, {"site_pos", &m->site_pos, 0},
      {"site_quat", &m->site_quat, 0}, {"site_size", &m->site_size, 0},
      {"actuator_jntid", &m->actuator_jntid, 1}, {"actuator_position", &m->actuator_position, 1},
      {"actuator_ctrllimited", &m->actuator_ctrllimited, 1}, {"actuator_forcelimited", &m->actuator_forcelimited, 1},
      {"actuator_kp", &m->actuator_kp, 0}, {"actuator_ctrlrange", &m->actuator_ctrlrange, 0},
      {"actuator_forcerange", &m->actuator_forcerange, 0}, {"sensor_siteid", &m->sensor_siteid, 1},
      {"pair_geom", &m->pair_geom, 1}, {"mocap_pos0", &m->mocap_pos0, 0}, {"mocap_quat0", &m->mocap_quat0, 0},
  };
  int narr = *(int32_t*)(buf + 8);
  long off = 16;
  for (int a = 0; a < narr; a++) {
    const char* name = buf + off;
    int code = *(int32_t*)(buf + off + 32), count = *(int32_t*)(buf + off + 36);
    off += 40;
    long nbytes = (long)count * (code == 0 ? 8 : 4);
    for (size_t k = 0; k < sizeof(fields) / sizeof(fields[0]); k++)
      if (!strcmp(fields[k].name, name) && fields[k].is_int == (code == 1)) *(void**)fields[k].dst = buf + off;
    off += nbytes + ((8 - nbytes % 8) % 8);
  }
  if (!nq || !nv || !nbody || !ts) { orc_free_model(m); return NULL; }
  m->nq = *nq; m->nv = *nv; m->nu = *nu; m->nbody = *nbody; m->njnt = *njnt; m->ngeom = *ngeom; m->nsite = *nsite;
  m->nmesh = *nmesh; m->nmocap = *nmocap; m->nsensor = *nsensor; m->npair = *npair;
  m->timestep = *ts; memcpy(m->gravity, grav, 24); m->tolerance = *tol; m->iterations = *iters;
  m->mpr_tolerance = *mprt; m->mpr_iterations = *mpri; m->meaninertia = *meani;
  m->solver = ORC_SOLVER_NEWTON; m->ls_iterations = 50; m->ls_tolerance = 0.01; m->mpr_output = 1; m->boxbox_depth_scale = 0.5;
  return m;
}
void orc_free_model(OrcModel* m) { if (m) { free(m->blob); free(m); } }

int orc_model_int(const OrcModel* m, const char* n) {
#define MI(x) if (!strcmp(n, #x)) return m->x;
  MI(nq) MI(nv) MI(nu) MI(nbody) MI(njnt) MI(ngeom) MI(nsite) MI(nmesh) MI(nmocap) MI(nsensor) MI(npair) MI(iterations)
#undef MI
  return -1;
}
int orc_set_option(OrcModel* m, const char* n, double v) {
  if (!strcmp(n, "timestep")) m->timestep = v;
  else if (!strcmp(n, "iterations")) m->iterations = (int)v;
  else if (!strcmp(n, "tolerance")) m->tolerance = v;
  else if (!strcmp(n, "disable_contact")) m->disable_contact = (int)v;
  else if (!strcmp(n, "solver")) m->solver = (int)v;
  else if (!strcmp(n, "ls_iterations")) m->ls_iterations = (int)v;
  else if (!strcmp(n, "ls_tolerance")) m->ls_tolerance = v;
  else if (!strcmp(n, "mpr_iterations")) m->mpr_iterations = (int)v;
  else if (!strcmp(n, "mpr_tolerance")) m->mpr_tolerance = v;
  else if (!strcmp(n, "round_state")) m->round_state = (int)v;
  else if (!strcmp(n, "mpr_output")) m->mpr_output = (int)v;
  else if (!strcmp(n, "boxbox_depth_scale")) m->boxbox_depth_scale = v;
  else return -1;
  return 0;
}

static double* dalloc(long n) { return (double*)calloc(n > 0 ? n : 1, sizeof(double)); }
static double* scratch(OrcData* d, long n) {
  if (n > d->scratch_cap) { free(d->scratch); d->scratch_cap = n + n / 2 + 64; d->scratch = (double*)malloc(sizeof(double) * d->scratch_cap); }
  return d->scratch;
}

OrcData* orc_make_data(const OrcModel* m) {
  OrcData* d = (OrcData*)calloc(1, sizeof(OrcData));
  int nv = m->nv, nb = m->nbody;
  d->qpos = dalloc(m->nq); d->qvel = dalloc(nv); d->ctrl = dalloc(m->nu); d->qacc_warmstart = dalloc(nv);
  d->mocap_pos = dalloc(3 * m->nmocap); d->mocap_quat = dalloc(4 * m->nmocap);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb); d->xipos = dalloc(3 * nb);
  d->xanchor = dalloc(3 * m->njnt); d->xaxis = dalloc(3 * m->njnt);
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom);
  d->site_xpos = dalloc(3 * m->nsite); d->site_xmat = dalloc(9 * m->nsite);
  d->cdof = dalloc(6 * nv); d->cdof_dot = dalloc(6 * nv); d->cinert = dalloc(10 * nb); d->crb = dalloc(10 * nb);
  d->cvel = dalloc(6 * nb); d->cacc = dalloc(6 * nb); d->cfrc = dalloc(6 * nb);
  d->qM = dalloc(nv * nv); d->qL = dalloc(nv * nv);
  d->qfrc_bias = dalloc(nv); d->qfrc_passive = dalloc(nv); d->qfrc_actuator = dalloc(nv); d->qfrc_smooth = dalloc(nv);
  d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qacc = dalloc(nv);
  d->actuator_force = dalloc(m->nu); d->sensordata = dalloc(m->nsensor);
  d->contact = (Contact*)calloc(ORC_MAXCON, sizeof(Contact));
  d->efc_cap = 64;
  int c = d->efc_cap;
  d->efc_J = dalloc((long)c * nv); d->efc_MinvJT = dalloc((long)c * nv); d->efc_AR = dalloc((long)c * c);
  d->efc_pos = dalloc(c); d->efc_margin = dalloc(c); d->efc_diagApprox = dalloc(c); d->efc_R = dalloc(c);
  d->efc_aref = dalloc(c); d->efc_b = dalloc(c); d->efc_force = dalloc(c); d->efc_vel = dalloc(c);
  d->efc_type = (int*)calloc(c, sizeof(int)); d->efc_id = (int*)calloc(c, sizeof(int));
  orc_reset(m, d);
  return d;
}
static void efc_reserve(const OrcModel* m, OrcData* d, int n) {
  if (n <= d->efc_cap) return;
  int c = d->efc_cap;
  while (c < n) c *= 2;
  int nv = m->nv;
  d->efc_J = (double*)realloc(d->efc_J, sizeof(double) * c * nv);
  d->efc_MinvJT = (double*)realloc(d->efc_MinvJT, sizeof(double) * c * nv);
  d->efc_AR = (double*)realloc(d->efc_AR, sizeof(double) * c * c);
#define RE(x) d->x = (double*)realloc(d->x, sizeof(double) * c)
  RE(efc_pos); RE(efc_margin); RE(efc_diagApprox); RE(efc_R); RE(efc_aref); RE(efc_b); RE(efc_force); RE(efc_vel);
#undef RE
  d->efc_type = (int*)realloc(d->efc_type, sizeof(int) * c);
  d->efc_id = (int*)realloc(d->efc_id, sizeof(int) * c);
  d->efc_cap = c;
}
void orc_free_data(OrcData* d) {
  if (!d) return;
  double* p[] = {d->qpos, d->qvel, d->ctrl, d->qacc_warmstart, d->mocap_pos, d->mocap_quat, d->xpos, d->xquat, d->xmat, d->xipos,
                 d->xanchor, d->xaxis, d->geom_xpos, d->geom_xmat, d->site_xpos, d->site_xmat, d->cdof, d->cdof_dot, d->cinert, d->crb,
                 d->cvel, d->cacc, d->cfrc, d->qM, d->qL, d->qfrc_bias, d->qfrc_passive, d->qfrc_actuator, d->qfrc_smooth, d->qacc_smooth,
                 d->qfrc_constraint, d->qacc, d->actuator_force, d->sensordata, d->efc_J, d->efc_MinvJT, d->efc_AR, d->efc_pos,
                 d->efc_margin, d->efc_diagApprox, d->efc_R, d->efc_aref, d->efc_b, d->efc_force, d->efc_vel};
  for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); i++) free(p[i]);
  free(d->contact); free(d->efc_type); free(d->efc_id); free(d->scratch); free(d);
}
void orc_reset(const OrcModel* m, OrcData* d) {
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memcpy(d->mocap_pos, m->mocap_pos0, sizeof(double) * 3 * m->nmocap);
  memcpy(d->mocap_quat, m->mocap_quat0, sizeof(double) * 4 * m->nmocap);
}

/* ------------------------------------------------------------------ 1. kinematics */
static void kinematics(const OrcModel* m, OrcData* d) {
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  quat2mat(d->xmat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b], mid = m->body_mocapid[b];
    double pos[3], quat[4];
    int ja = m->body_jntadr[b], jn = m->body_jntnum[b];
    if (mid >= 0) {
      copy3(pos, d->mocap_pos + 3 * mid);
      memcpy(quat, d->mocap_quat + 4 * mid, 32);
      quat_normalize(quat);
    } else if (jn == 1 && m->jnt_type[ja] == J_FREE) {
      int qa = m->jnt_qposadr[ja];
      copy3(pos, d->qpos + qa);
      memcpy(quat, d->qpos + qa + 3, 32);
      quat_normalize(quat);
      copy3(d->xanchor + 3 * ja, pos);
      d->xaxis[3 * ja] = 0; d->xaxis[3 * ja + 1] = 0; d->xaxis[3 * ja + 2] = 1;
    } else {
      double t[3];
      mulmv(t, d->xmat + 9 * p, m->body_pos + 3 * b);
      add3(pos, d->xpos + 3 * p, t);
      quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
      for (int j = ja; j < ja + jn; j++) {
        int qa = m->jnt_qposadr[j];
        double ang = d->qpos[qa] - m->qpos0[qa], off[3], qj[4], qn[4];
        rotvecq(off, quat, m->jnt_pos + 3 * j);
        add3(d->xanchor + 3 * j, pos, off);
        rotvecq(d->xaxis + 3 * j, quat, m->jnt_axis + 3 * j);
        axisangle2quat(qj, m->jnt_axis + 3 * j, ang);
        quat_mul(qn, quat, qj);
        memcpy(quat, qn, 32);
        rotvecq(off, quat, m->jnt_pos + 3 * j);
        sub3(pos, d->xanchor + 3 * j, off);
      }
      quat_normalize(quat);
    }
    copy3(d->xpos + 3 * b, pos);
    memcpy(d->xquat + 4 * b, quat, 32);
    quat2mat(d->xmat + 9 * b, quat);
    double t[3];
    mulmv(t, d->xmat + 9 * b, m->body_ipos + 3 * b);
    add3(d->xipos + 3 * b, pos, t);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3], gm[9];
    mulmv(t, d->xmat + 9 * b, m->geom_pos + 3 * g);
    add3(d->geom_xpos + 3 * g, d->xpos + 3 * b, t);
    quat2mat(gm, m->geom_quat + 4 * g);
    mulmm(d->geom_xmat + 9 * g, d->xmat + 9 * b, gm);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3], gm[9];
    mulmv(t, d->xmat + 9 * b, m->site_pos + 3 * s);
    add3(d->site_xpos + 3 * s, d->xpos + 3 * b, t);
    quat2mat(gm, m->site_quat + 4 * s);
    mulmm(d->site_xmat + 9 * s, d->xmat + 9 * b, gm);
  }
}

/* spatial quantities about the world origin.  motion = [w(3), v(3)], force = [n(3), f(3)],
 * inertia = [m, h(3) = m*c, I_O(6) = xx yy zz xy xz yz]. */
static void inert_mul(double* F, const double* I, const double* mv) {
  const double *w = mv, *v = mv + 3, *h = I + 1;
  double t[3];
  cross3(t, w, h);
  F[3] = I[0] * v[0] + t[0]; F[4] = I[0] * v[1] + t[1]; F[5] = I[0] * v[2] + t[2];
  cross3(t, h, v);
  F[0] = I[4] * w[0] + I[7] * w[1] + I[8] * w[2] + t[0];
  F[1] = I[7] * w[0] + I[5] * w[1] + I[9] * w[2] + t[1];
  F[2] = I[8] * w[0] + I[9] * w[1] + I[6] * w[2] + t[2];
}
static inline double dot6(const double* a, const double* b) { return dot3(a, b) + dot3(a + 3, b + 3); }
static void cross_motion(double* r, const double* a, const double* b) { /* a x_m b */
  double t1[3], t2[3];
  cross3(r, a, b);
  cross3(t1, a, b + 3);
  cross3(t2, a + 3, b);
  add3(r + 3, t1, t2);
}
static void cross_force(double* r, const double* a, const double* f) { /* a x* f */
  double t1[3], t2[3];
  cross3(t1, a, f);
  cross3(t2, a + 3, f + 3);
  add3(r, t1, t2);
  cross3(r + 3, a, f + 3);
}

/* 2. composite inertias, motion subspaces, mass matrix (CRBA), Cholesky factor.
 * Stands for the qM that mj_fullM exposes to the controller (mujoco_config.py:320). */
static void com_pos_crb(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  for (int b = 0; b < m->nbody; b++) {
    double* I = d->cinert + 10 * b;
    double mass = m->body_mass[b];
    const double *c = d->xipos + 3 * b, *R = d->xmat + 9 * b;
    double Iw[9], T[9], RT[9];
    mulmm(T, R, m->body_inertia + 9 * b);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RT[3 * i + j] = R[3 * j + i];
    mulmm(Iw, T, RT);
    double cc = dot3(c, c);
    I[0] = mass; I[1] = mass * c[0]; I[2] = mass * c[1]; I[3] = mass * c[2];
    I[4] = Iw[0] + mass * (cc - c[0] * c[0]); I[5] = Iw[4] + mass * (cc - c[1] * c[1]); I[6] = Iw[8] + mass * (cc - c[2] * c[2]);
    I[7] = Iw[1] - mass * c[0] * c[1]; I[8] = Iw[2] - mass * c[0] * c[2]; I[9] = Iw[5] - mass * c[1] * c[2];
  }
  for (int j = 0; j < m->njnt; j++) {
    int da = m->jnt_dofadr[j], b = m->jnt_bodyid[j];
    if (m->jnt_type[j] == J_FREE) {
      for (int k = 0; k < 3; k++) {
        double* S = d->cdof + 6 * (da + k);
        memset(S, 0, 48);
        S[3 + k] = 1;
        double* Sr = d->cdof + 6 * (da + 3 + k);
        const double* R = d->xmat + 9 * b;
        Sr[0] = R[k]; Sr[1] = R[3 + k]; Sr[2] = R[6 + k];
        cross3(Sr + 3, d->xpos + 3 * b, Sr);
      }
    } else {
      double* S = d->cdof + 6 * da;
      copy3(S, d->xaxis + 3 * j);
      cross3(S + 3, d->xanchor + 3 * j, S);
    }
  }
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * m->nbody);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  memset(d->qM, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    double F[6];
    inert_mul(F, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = dot6(d->cdof + 6 * j, F);
      d->qM[i * nv + j] = v;
      d->qM[j * nv + i] = v;
    }
  }
}
static int cholesky(double* L, const double* A, int n) {
  memcpy(L, A, sizeof(double) * n * n);
  for (int j = 0; j < n; j++) {
    double s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (s < MINVAL) return -1;
    s = sqrt(s);
    L[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      double t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / s;
    }
    for (int i = 0; i < j; i++) L[i * n + j] = 0;
  }
  return 0;
}
static void chol_solve(const double* L, double* x, int n) {
  for (int i = 0; i < n; i++) {
    double t = x[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}

/* 6a. velocities + RNE bias (gravity, Coriolis, centrifugal) = qfrc_bias, the `g` the
 * controller reads back (mujoco_config.py:216). */
static void rne_bias(const OrcModel* m, OrcData* d) {
  memset(d->cvel, 0, 48);
  memset(d->cacc, 0, 48);
  d->cacc[3] = -m->gravity[0]; d->cacc[4] = -m->gravity[1]; d->cacc[5] = -m->gravity[2];
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    double *cv = d->cvel + 6 * b, *ca = d->cacc + 6 * b;
    memcpy(cv, d->cvel + 6 * p, 48);
    memcpy(ca, d->cacc + 6 * p, 48);
    for (int j = m->body_jntadr[b]; j < m->body_jntadr[b] + m->body_jntnum[b]; j++) {
      int da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == J_FREE) {
        for (int k = 0; k < 3; k++) {
          memset(d->cdof_dot + 6 * (da + k), 0, 48);
          for (int c = 0; c < 6; c++) cv[c] += d->cdof[6 * (da + k) + c] * d->qvel[da + k];
        }
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (da + k), cv, d->cdof + 6 * (da + k));
        for (int k = 3; k < 6; k++)
          for (int c = 0; c < 6; c++) {
            cv[c] += d->cdof[6 * (da + k) + c] * d->qvel[da + k];
            ca[c] += d->cdof_dot[6 * (da + k) + c] * d->qvel[da + k];
          }
      } else {
        cross_motion(d->cdof_dot + 6 * da, cv, d->cdof + 6 * da);
        for (int c = 0; c < 6; c++) {
          cv[c] += d->cdof[6 * da + c] * d->qvel[da];
          ca[c] += d->cdof_dot[6 * da + c] * d->qvel[da];
        }
      }
    }
    double Ia[6], Iv[6], t[6];
    inert_mul(Ia, d->cinert + 10 * b, ca);
    inert_mul(Iv, d->cinert + 10 * b, cv);
    cross_force(t, cv, Iv);
    for (int c = 0; c < 6; c++) d->cfrc[6 * b + c] = Ia[c] + t[c];
  }
  memset(d->cfrc, 0, 48);
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 6; c++) d->cfrc[6 * p + c] += d->cfrc[6 * b + c];
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof + 6 * i, d->cfrc + 6 * m->dof_bodyid[i]);
}

/* Jacobian of a world point moving with `body`: jp, jr are 3 x nv row-major. */
static void jac_point(const OrcModel* m, const OrcData* d, int body, const double* p, double* jp, double* jr) {
  int nv = m->nv;
  if (jp) memset(jp, 0, sizeof(double) * 3 * nv);
  if (jr) memset(jr, 0, sizeof(double) * 3 * nv);
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  for (int i = m->body_dofadr[body] + m->body_dofnum[body] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double* S = d->cdof + 6 * i;
    if (jr) { jr[i] = S[0]; jr[nv + i] = S[1]; jr[2 * nv + i] = S[2]; }
    if (jp) {
      double t[3];
      cross3(t, S, p);
      jp[i] = S[3] + t[0]; jp[nv + i] = S[4] + t[1]; jp[2 * nv + i] = S[5] + t[2];
    }
  }
}
void orc_jac_body_com(const OrcModel* m, const OrcData* d, int body, double* jacp, double* jacr) {
  jac_point(m, d, body, d->xipos + 3 * body, jacp, jacr);
}

/* 6b. passive + actuation (motor clamp; position servo kp*(ctrl - q), ctrl- and force-clamped) */
static void passive_actuation(const OrcModel* m, OrcData* d) {
  for (int i = 0; i < m->nv; i++) { d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i]; d->qfrc_actuator[i] = 0; }
  if (m->jnt_stiffness) /* joint springs [EXT mj_passive]: -stiffness (qpos - qpos_spring), hinge joints (jaco2_torque.xml:109-133) */
    for (int j = 0; j < m->njnt; j++)
      if (m->jnt_type[j] == J_HINGE && m->jnt_stiffness[j] != 0)
        d->qfrc_passive[m->jnt_dofadr[j]] -= m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->jnt_springref[j]);
  for (int a = 0; a < m->nu; a++) {
    int j = m->actuator_jntid[a];
    double c = d->ctrl[a];
    if (m->actuator_ctrllimited[a]) c = fmax(m->actuator_ctrlrange[2 * a], fmin(m->actuator_ctrlrange[2 * a + 1], c));
    double f = m->actuator_position[a] ? m->actuator_kp[a] * (c - d->qpos[m->jnt_qposadr[j]]) : c;
    if (m->actuator_forcelimited[a]) f = fmax(m->actuator_forcerange[2 * a], fmin(m->actuator_forcerange[2 * a + 1], f));
    d->actuator_force[a] = f;
    d->qfrc_actuator[m->jnt_dofadr[j]] += f;
  }
}

/* ------------------------------------------------------------------ 3. collision */
static void make_frame(double* fr) { /* fr[0:3] = normal given; builds two tangents */
  normalize3(fr);
  double* y = fr + 3;
  y[0] = y[1] = y[2] = 0;
  if (fr[1] < 0.5 && fr[1] > -0.5) y[1] = 1; else y[2] = 1;
  double dd = dot3(fr, y);
  addscl3(y, y, fr, -dd);
  normalize3(y);
  cross3(fr + 6, fr, y);
}
static Contact* add_contact(OrcData* d, int g1, int g2, double dist, const double* pos, const double* normal) {
  if (d->ncon >= ORC_MAXCON) return NULL;
  Contact* c = d->contact + d->ncon++;
  c->dist = dist; copy3(c->pos, pos); copy3(c->frame, normal);
  make_frame(c->frame);
  c->geom1 = g1; c->geom2 = g2;
  return c;
}
static void support_geom(const OrcModel* m, const OrcData* d, int g, const double* dir, double* out) {
  const double *R = d->geom_xmat + 9 * g, *p = d->geom_xpos + 3 * g, *sz = m->geom_size + 3 * g;
  double l[3], s[3];
  mulmtv(l, R, dir);
  switch (m->geom_type[g]) {
    case G_SPHERE: { double n = norm3(l); scl3(s, l, n > MINVAL ? sz[0] / n : 0); break; }
    case G_BOX: for (int i = 0; i < 3; i++) s[i] = l[i] > 0 ? sz[i] : -sz[i]; break;
    case G_CYLINDER: { /* radius sz[0], half height sz[1], axis z [EXT: mjc_Convex treats cylinders through their support function] */
      double rr = sqrt(l[0] * l[0] + l[1] * l[1]), k = rr > MINVAL ? sz[0] / rr : 0;
      s[0] = l[0] * k; s[1] = l[1] * k; s[2] = l[2] > 0 ? sz[1] : -sz[1];
      break;
    }
    case G_MESH: {
      int md = m->geom_dataid[g], n = m->mesh_vertnum[md];
      const double* v = m->mesh_vert + 3 * m->mesh_vertadr[md];
      int best = 0; double bd = -1e300;
      for (int i = 0; i < n; i++) { double t = dot3(v + 3 * i, l); if (t > bd) { bd = t; best = i; } }
      copy3(s, v + 3 * best);
      break;
    }
    default: s[0] = s[1] = s[2] = 0;
  }
  mulmv(out, R, s);
  add3(out, out, p);
}

static void collide_plane_sphere(const OrcModel* m, OrcData* d, int g1, int g2) {
  const double *R = d->geom_xmat + 9 * g1;
  double n[3] = {R[2], R[5], R[8]}, t[3];
  sub3(t, d->geom_xpos + 3 * g2, d->geom_xpos + 3 * g1);
  double dist = dot3(t, n) - m->geom_size[3 * g2];
  if (dist > 0) return;
  double pos[3];
  addscl3(pos, d->geom_xpos + 3 * g2, n, -m->geom_size[3 * g2] - dist / 2);
  add_contact(d, g1, g2, dist, pos, n);
}
static void collide_plane_box(const OrcModel* m, OrcData* d, int g1, int g2) {
  const double *R = d->geom_xmat + 9 * g1, *B = d->geom_xmat + 9 * g2, *sz = m->geom_size + 3 * g2;
  double n[3] = {R[2], R[5], R[8]};
  int cnt = 0;
  for (int i = 0; i < 8 && cnt < 4; i++) {
    double l[3] = {(i & 1 ? sz[0] : -sz[0]), (i & 2 ? sz[1] : -sz[1]), (i & 4 ? sz[2] : -sz[2])}, c[3], t[3];
    mulmv(c, B, l);
    add3(c, c, d->geom_xpos + 3 * g2);
    sub3(t, c, d->geom_xpos + 3 * g1);
    double dist = dot3(t, n);
    if (dist > 0) continue;
    double pos[3];
    addscl3(pos, c, n, -dist / 2);
    add_contact(d, g1, g2, dist, pos, n);
    cnt++;
  }
}
static void collide_plane_convex(const OrcModel* m, OrcData* d, int g1, int g2) {
  const double* R = d->geom_xmat + 9 * g1;
  double n[3] = {R[2], R[5], R[8]}, nn[3] = {-n[0], -n[1], -n[2]}, s[3], t[3];
  support_geom(m, d, g2, nn, s);
  sub3(t, s, d->geom_xpos + 3 * g1);
  double dist = dot3(t, n);
  if (dist > 0) return;
  double pos[3];
  addscl3(pos, s, n, -dist / 2);
  add_contact(d, g1, g2, dist, pos, n);
}

/* box-box: 15-axis SAT; face case -> polygon intersection of reference rectangle with the
 * projected incident face (vertices-inside + edge crossings, <= 8 points); edge case -> 1 point. */
static void collide_box_box(const OrcModel* m, OrcData* d, int g1, int g2) {
  const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2, *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
  double A[3][3], B[3][3], pp[3];
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = R1[3 * k + i]; B[i][k] = R2[3 * k + i]; }
  sub3(pp, p2, p1);
  double C[3][3], Q[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
  double best = 1e300; int code = -1; double bsign = 1;
  double pa[3] = {dot3(pp, A[0]), dot3(pp, A[1]), dot3(pp, A[2])}, pb[3] = {dot3(pp, B[0]), dot3(pp, B[1]), dot3(pp, B[2])};
  for (int i = 0; i < 3; i++) {
    double pen = s1[i] + s2[0] * Q[i][0] + s2[1] * Q[i][1] + s2[2] * Q[i][2] - fabs(pa[i]);
    if (pen < 0) return;
    if (pen < best) { best = pen; code = i; bsign = pa[i] < 0 ? -1 : 1; }
  }
  for (int j = 0; j < 3; j++) {
    double pen = s2[j] + s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j] - fabs(pb[j]);
    if (pen < 0) return;
    if (pen < best) { best = pen; code = 3 + j; bsign = pb[j] < 0 ? -1 : 1; }
  }
  double ebest = 1e300; int ecode = -1; double esign = 1, eaxis[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double ax[3];
      cross3(ax, A[i], B[j]);
      double l = norm3(ax);
      if (l < 1e-6) continue;
      scl3(ax, ax, 1 / l);
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += s1[k] * fabs(dot3(ax, A[k])); rb += s2[k] * fabs(dot3(ax, B[k])); }
      double dp = dot3(pp, ax), pen = ra + rb - fabs(dp);
      if (pen < 0) return;
      if (pen < ebest) { ebest = pen; ecode = 6 + 3 * i + j; esign = dp < 0 ? -1 : 1; copy3(eaxis, ax); }
    }
  if (ecode >= 0 && ebest * 1.05 < best) { /* edge-edge */
    int i = (ecode - 6) / 3, j = (ecode - 6) % 3;
    double n[3];
    scl3(n, eaxis, esign);
    double ea[3], eb[3];
    copy3(ea, p1);
    for (int k = 0; k < 3; k++) if (k != i) addscl3(ea, ea, A[k], dot3(n, A[k]) > 0 ? s1[k] : -s1[k]);
    copy3(eb, p2);
    for (int k = 0; k < 3; k++) if (k != j) addscl3(eb, eb, B[k], dot3(n, B[k]) > 0 ? -s2[k] : s2[k]);
    /* closest points of lines ea + s A[i], eb + t B[j] */
    double r[3];
    sub3(r, eb, ea);
    double uv = C[i][j], du = dot3(r, A[i]), dv = dot3(r, B[j]), den = 1 - uv * uv;
    double s = den > 1e-12 ? (du - uv * dv) / den : 0, t = den > 1e-12 ? (uv * du - dv) / den : 0;
    double ca[3], cb[3], pos[3];
    addscl3(ca, ea, A[i], s);
    addscl3(cb, eb, B[j], t);
    add3(pos, ca, cb);
    scl3(pos, pos, 0.5);
    add_contact(d, g1, g2, -ebest, pos, n);
    return;
  }
  /* face contact: reference box r (axis ia), incident box o.  Contact point = midpoint between the incident point and the reference face;
     dist = HALF the overlap there (m->boxbox_depth_scale): with the full overlap the object rests on the holder at 0.19998548 and leaves its
     1 cm spawn overlap in ~20 ms; the reference's MuJoCo rests at 0.19997096 and takes 600 ms, and with the half depth every one of its
     thirteen recorded float32 values is reproduced (tests/test_mujoco_statics.py).  The edge-edge branch above has no recorded datum: full depth. */
  int refis1 = code < 3, ia = refis1 ? code : code - 3;
  const double (*RA)[3] = refis1 ? A : B, (*IA)[3] = refis1 ? B : A;
  const double *rs = refis1 ? s1 : s2, *is = refis1 ? s2 : s1, *rp = refis1 ? p1 : p2, *ip = refis1 ? p2 : p1;
  double nref[3]; /* outward normal of the reference face, pointing toward the incident box */
  scl3(nref, RA[ia], refis1 ? bsign : -bsign);
  int iu = (ia + 1) % 3, iv = (ia + 2) % 3;
  /* incident face: most anti-parallel to nref */
  int ib = 0; double mind = 1e300, isg = 1;
  for (int k = 0; k < 3; k++) {
    double dd = dot3(IA[k], nref);
    if (-fabs(dd) < mind) { mind = -fabs(dd); ib = k; isg = dd > 0 ? -1 : 1; }
  }
  int ju = (ib + 1) % 3, jv = (ib + 2) % 3;
  double fc[3]; /* incident face centre */
  addscl3(fc, ip, IA[ib], isg * is[ib]);
  /* express incident face in reference 2D coords (u,v) + height w above the reference face plane */
  double rel[3], c2[3], eu[3], ev[3];
  sub3(rel, fc, rp);
  c2[0] = dot3(rel, RA[iu]); c2[1] = dot3(rel, RA[iv]); c2[2] = dot3(rel, nref) - rs[ia];
  eu[0] = dot3(IA[ju], RA[iu]) * is[ju]; eu[1] = dot3(IA[ju], RA[iv]) * is[ju]; eu[2] = dot3(IA[ju], nref) * is[ju];
  ev[0] = dot3(IA[jv], RA[iu]) * is[jv]; ev[1] = dot3(IA[jv], RA[iv]) * is[jv]; ev[2] = dot3(IA[jv], nref) * is[jv];
  static const double sg[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
  double q[4][3];
  for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) q[k][c] = c2[c] + sg[k][0] * eu[c] + sg[k][1] * ev[c];
  double a = rs[iu], b = rs[iv];
  double pts[24][3]; int np = 0;
  /* (a) incident vertices inside the reference rectangle */
  for (int k = 0; k < 4; k++)
    if (fabs(q[k][0]) <= a && fabs(q[k][1]) <= b) { memcpy(pts[np++], q[k], 24); }
  /* (b) reference corners inside the incident parallelogram */
  double det = eu[0] * ev[1] - eu[1] * ev[0];
  if (fabs(det) > 1e-14)
    for (int k = 0; k < 4; k++) {
      double x = sg[k][0] * a - c2[0], y = sg[k][1] * b - c2[1];
      double al = (x * ev[1] - y * ev[0]) / det, be = (eu[0] * y - eu[1] * x) / det;
      if (fabs(al) < 1 && fabs(be) < 1) {
        pts[np][0] = sg[k][0] * a; pts[np][1] = sg[k][1] * b; pts[np][2] = c2[2] + al * eu[2] + be * ev[2];
        np++;
      }
    }
  /* (c) proper crossings of incident edges with reference edges */
  for (int k = 0; k < 4; k++) {
    const double *q0 = q[k], *q1 = q[(k + 1) & 3];
    for (int e = 0; e < 4; e++) {
      int ax = e & 1; /* 0: edge u = +-a, 1: edge v = +-b */
      double lim = (e & 2 ? 1 : -1) * (ax ? b : a), other = ax ? a : b;
      double d0 = q0[ax] - lim, d1 = q1[ax] - lim;
      if ((d0 < 0) == (d1 < 0) || d0 == d1) continue;
      double t = d0 / (d0 - d1);
      if (!(t > 0 && t < 1)) continue;
      double o = q0[1 - ax] + t * (q1[1 - ax] - q0[1 - ax]);
      if (!(fabs(o) < other)) continue;
      pts[np][ax] = lim; pts[np][1 - ax] = o; pts[np][2] = q0[2] + t * (q1[2] - q0[2]);
      np++;
    }
  }
  double n12[3];
  scl3(n12, nref, refis1 ? 1 : -1);
  for (int k = 0; k < np; k++) {
    double w = pts[k][2];
    if (w > 0) continue;
    double pos[3];
    addscl3(pos, rp, RA[iu], pts[k][0]);
    addscl3(pos, pos, RA[iv], pts[k][1]);
    addscl3(pos, pos, nref, rs[ia] + w / 2);
    add_contact(d, g1, g2, w * m->boxbox_depth_scale, pos, n12);
  }
}

/* ---- MPR (Minkowski Portal Refinement) penetration query; Minkowski difference = geom1 - geom2 */
typedef struct { double v[3], v1[3], v2[3]; } Sup;
static _Thread_local long g_mpr_queries_now;   /* (thread-local: orc_step_batch runs envs on all cores) */
static void mpr_support(const OrcModel* m, const OrcData* d, int g1, int g2, const double* dir, Sup* s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  g_mpr_queries_now++;   /* (diagnostic) */
  support_geom(m, d, g1, dir, s->v1);
  support_geom(m, d, g2, nd, s->v2);
  sub3(s->v, s->v1, s->v2);
}
static void portal_dir(const Sup* p, double* dir) {
  double a[3], b[3];
  sub3(a, p[2].v, p[1].v);
  sub3(b, p[3].v, p[1].v);
  cross3(dir, a, b);
  normalize3(dir);
}
static int reach_tol(const Sup* p, const Sup* v4, const double* dir, double tol) {
  double dv4 = dot3(v4->v, dir), m1 = dv4 - dot3(p[1].v, dir), m2 = dv4 - dot3(p[2].v, dir), m3 = dv4 - dot3(p[3].v, dir);
  double mn = fmin(m1, fmin(m2, m3));
  return mn <= tol;
}
static void expand_portal(Sup* p, const Sup* v4) {
  double c[3];
  cross3(c, v4->v, p[0].v);
  if (dot3(p[1].v, c) > 0) {
    if (dot3(p[2].v, c) > 0) p[1] = *v4; else p[3] = *v4;
  } else {
    if (dot3(p[3].v, c) > 0) p[2] = *v4; else p[1] = *v4;
  }
}
static double point_tri_closest(const double* a, const double* b, const double* c, double* cp) { /* closest point to origin */
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(ab, b, a); sub3(ac, c, a);
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { copy3(cp, a); return norm3(cp); }
  double bp[3] = {-b[0], -b[1], -b[2]}, d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { copy3(cp, b); return norm3(cp); }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { addscl3(cp, a, ab, d1 / (d1 - d3)); return norm3(cp); }
  double cpv[3] = {-c[0], -c[1], -c[2]}, d5 = dot3(ab, cpv), d6 = dot3(ac, cpv);
  if (d6 >= 0 && d5 <= d6) { copy3(cp, c); return norm3(cp); }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { addscl3(cp, a, ac, d2 / (d2 - d6)); return norm3(cp); }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double bc[3];
    sub3(bc, c, b);
    addscl3(cp, b, bc, (d4 - d3) / ((d4 - d3) + (d5 - d6)));
    return norm3(cp);
  }
  double den = 1 / (va + vb + vc), v = vb * den, w = vc * den;
  addscl3(cp, a, ab, v);
  addscl3(cp, cp, ac, w);
  return norm3(cp);
}
static void mpr_find_pos(const Sup* p, double* pos) {
  double b[4], t[3], dir[3];
  cross3(t, p[1].v, p[2].v); b[0] = dot3(t, p[3].v);
  cross3(t, p[3].v, p[2].v); b[1] = dot3(t, p[0].v);
  cross3(t, p[0].v, p[1].v); b[2] = dot3(t, p[3].v);
  cross3(t, p[2].v, p[1].v); b[3] = dot3(t, p[0].v);
  double sum = b[0] + b[1] + b[2] + b[3];
  if (sum <= 0) {
    b[0] = 0;
    portal_dir(p, dir);
    cross3(t, p[2].v, p[3].v); b[1] = dot3(t, dir);
    cross3(t, p[3].v, p[1].v); b[2] = dot3(t, dir);
    cross3(t, p[1].v, p[2].v); b[3] = dot3(t, dir);
    sum = b[1] + b[2] + b[3];
  }
  double inv = 1 / sum, a1[3] = {0, 0, 0}, a2[3] = {0, 0, 0};
  for (int k = 0; k < 4; k++) { addscl3(a1, a1, p[k].v1, b[k]); addscl3(a2, a2, p[k].v2, b[k]); }
  add3(pos, a1, a2);
  scl3(pos, pos, 0.5 * inv);
}
/* diagnostic counters (tools/mpr_query_stats.py): [0] MPR calls, [1] hits, [2] support queries of the hit calls, [3] of the miss calls */
static _Thread_local long g_orc_counter[4];   /* (per thread: read them from the thread that stepped, tools/mpr_query_stats.py is single-threaded) */
long orc_debug_counter(int i, int reset) { long v = g_orc_counter[i & 3]; if (reset) g_orc_counter[i & 3] = 0; return v; }
static int mpr_penetration_impl(const OrcModel* m, const OrcData* d, int g1, int g2, double* depth, double* dir, double* pos);
/* returns 0 on penetration (depth, dir = from geom1 toward geom2, pos), -1 otherwise */
static int mpr_penetration(const OrcModel* m, const OrcData* d, int g1, int g2, double* depth, double* dir, double* pos) {
  g_mpr_queries_now = 0;
  const int rc = mpr_penetration_impl(m, d, g1, g2, depth, dir, pos);
  g_orc_counter[0]++;
  if (rc == 0) { g_orc_counter[1]++; g_orc_counter[2] += g_mpr_queries_now; } else g_orc_counter[3] += g_mpr_queries_now;
  return rc;
}
static int mpr_penetration_impl(const OrcModel* m, const OrcData* d, int g1, int g2, double* depth, double* dir, double* pos) {
  Sup p[4], v4;
  double tol = m->mpr_tolerance, dr[3], va[3], vb[3];
  sub3(p[0].v, d->geom_xpos + 3 * g1, d->geom_xpos + 3 * g2);
  copy3(p[0].v1, d->geom_xpos + 3 * g1);
  copy3(p[0].v2, d->geom_xpos + 3 * g2);
  if (norm3(p[0].v) < 1e-12) p[0].v[0] = 1e-5;
  scl3(dr, p[0].v, -1);
  normalize3(dr);
  mpr_support(m, d, g1, g2, dr, &p[1]);
  if (dot3(p[1].v, dr) <= 0) return -1;
  cross3(dr, p[0].v, p[1].v);
  if (norm3(dr) < 1e-12) { /* origin on the v0-v1 ray: penetration along it */
    *depth = norm3(p[1].v);
    copy3(dir, p[1].v);
    normalize3(dir);
    add3(pos, p[1].v1, p[1].v2);
    scl3(pos, pos, 0.5);
    return 0;
  }
  normalize3(dr);
  mpr_support(m, d, g1, g2, dr, &p[2]);
  if (dot3(p[2].v, dr) <= 0) return -1;
  sub3(va, p[1].v, p[0].v);
  sub3(vb, p[2].v, p[0].v);
  cross3(dr, va, vb);
  normalize3(dr);
  if (dot3(dr, p[0].v) > 0) { Sup t = p[1]; p[1] = p[2]; p[2] = t; scl3(dr, dr, -1); }
  for (int it = 0;; it++) {
    if (it > 100) return -1;
    mpr_support(m, d, g1, g2, dr, &p[3]);
    if (dot3(p[3].v, dr) <= 0) return -1;
    int cont = 0;
    cross3(va, p[1].v, p[3].v);
    if (dot3(va, p[0].v) < -1e-14) { p[2] = p[3]; cont = 1; }
    if (!cont) {
      cross3(va, p[3].v, p[2].v);
      if (dot3(va, p[0].v) < -1e-14) { p[1] = p[3]; cont = 1; }
    }
    if (!cont) break;
    sub3(va, p[1].v, p[0].v);
    sub3(vb, p[2].v, p[0].v);
    cross3(dr, va, vb);
    normalize3(dr);
  }
  /* refine until the portal encloses the origin */
  for (int it = 0;; it++) {
    portal_dir(p, dr);
    if (dot3(dr, p[1].v) >= 0) break;
    mpr_support(m, d, g1, g2, dr, &v4);
    if (dot3(v4.v, dr) < 0 || reach_tol(p, &v4, dr, tol) || it > m->mpr_iterations) return -1;
    expand_portal(p, &v4);
  }
  /* penetration info */
  for (int it = 0;; it++) {
    portal_dir(p, dr);
    mpr_support(m, d, g1, g2, dr, &v4);
    if (reach_tol(p, &v4, dr, tol) || it > m->mpr_iterations) {
      if (m->mpr_output == 1) {
        /* Portal-plane output: the refined portal lies (within mpr_tolerance) in the face of the Minkowski difference that
         * the ray from the interior point through the origin leaves by; its normal and the support value h(dr) = v4 . dr
         * do not depend on WHICH triangle of that face the refinement ended on.  libccd's closest point of the final
         * triangle (mpr_output 0) equals this whenever the origin projects into the triangle, and otherwise depends on the
         * refinement path even in exact arithmetic (and, for shallow contacts, on the last bits of a ~1e-6 vector). */
        copy3(dir, dr);
        *depth = dot3(v4.v, dr);
      } else {
        double cp[3];
        *depth = point_tri_closest(p[1].v, p[2].v, p[3].v, cp);
        if (*depth < 1e-14) copy3(dir, dr); else { copy3(dir, cp); normalize3(dir); }
      }
      mpr_find_pos(p, pos);
      return 0;
    }
    expand_portal(p, &v4);
  }
}
static void collide_convex(const OrcModel* m, OrcData* d, int g1, int g2) {
  double depth, dir[3], pos[3];
  if (mpr_penetration(m, d, g1, g2, &depth, dir, pos)) return;
  add_contact(d, g1, g2, -depth, pos, dir);
}

static void collision(const OrcModel* m, OrcData* d) {
  d->ncon = 0;
  if (m->disable_contact) return;
  for (int k = 0; k < m->npair; k++) {
    int g1 = m->pair_geom[2 * k], g2 = m->pair_geom[2 * k + 1];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    if (t1 > t2) { int t = g1; g1 = g2; g2 = t; t = t1; t1 = t2; t2 = t; }
    double df[3];
    sub3(df, d->geom_xpos + 3 * g2, d->geom_xpos + 3 * g1);
    if (t1 == G_PLANE) {
      const double* R = d->geom_xmat + 9 * g1;
      double n[3] = {R[2], R[5], R[8]};
      if (dot3(df, n) > m->geom_rbound[g2]) continue;
    } else {
      double r = m->geom_rbound[g1] + m->geom_rbound[g2];
      if (dot3(df, df) > r * r) continue;
    }
    int before = d->ncon;
    if (t1 == G_PLANE && t2 == G_SPHERE) collide_plane_sphere(m, d, g1, g2);
    else if (t1 == G_PLANE && t2 == G_BOX) collide_plane_box(m, d, g1, g2);
    else if (t1 == G_PLANE && t2 == G_MESH) collide_plane_convex(m, d, g1, g2);
    else if (t1 == G_PLANE) continue;
    else if (t1 == G_BOX && t2 == G_BOX) collide_box_box(m, d, g1, g2);
    else collide_convex(m, d, g1, g2);
    /* contact parameters: max condim, max friction, mean solref/solimp (equal solmix), refsafe */
    for (int c = before; c < d->ncon; c++) {
      Contact* con = d->contact + c;
      con->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
      double f[3];
      for (int i = 0; i < 3; i++) f[i] = fmax(m->geom_friction[3 * g1 + i], m->geom_friction[3 * g2 + i]);
      con->mu[0] = con->mu[1] = f[0]; con->mu[2] = f[1]; con->mu[3] = con->mu[4] = f[2];
      for (int i = 0; i < 2; i++) con->solref[i] = 0.5 * (m->geom_solref[2 * g1 + i] + m->geom_solref[2 * g2 + i]);
      for (int i = 0; i < 5; i++) con->solimp[i] = 0.5 * (m->geom_solimp[5 * g1 + i] + m->geom_solimp[5 * g2 + i]);
      con->solref[0] = fmax(con->solref[0], 2 * m->timestep);
      con->margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    }
  }
}

/* ------------------------------------------------------------------ 4-5. constraint rows */
static double impedance(const double* solimp, double pos) {
  double dmin = fmin(MAXIMP, fmax(MINIMP, solimp[0])), dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
  double width = fmax(MINVAL, solimp[2]), mid = fmin(MAXIMP, fmax(MINIMP, solimp[3])), power = fmax(1, solimp[4]);
  double x = fabs(pos) / width, y;
  if (x >= 1) return dmax;
  if (power == 1) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
static void finish_row(const OrcModel* m, OrcData* d, int r, const double* solref, const double* solimp, double diagApprox) {
  int nv = m->nv;
  double vel = 0;
  for (int i = 0; i < nv; i++) vel += d->efc_J[r * nv + i] * d->qvel[i];
  double pos = d->efc_pos[r] - d->efc_margin[r];
  double imp = impedance(solimp, pos);
  double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
  double K = 1 / fmax(MINVAL, dmax * dmax * solref[0] * solref[0] * solref[1] * solref[1]);
  double B = 2 / fmax(MINVAL, dmax * solref[0]);
  d->efc_vel[r] = vel;
  d->efc_aref[r] = -B * vel - K * imp * pos;
  d->efc_diagApprox[r] = diagApprox;
  d->efc_R[r] = fmax(MINVAL, (1 - imp) * diagApprox / imp);
}
static void make_constraint(const OrcModel* m, OrcData* d) {
  int nv = m->nv, r = 0;
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] != J_HINGE) continue;
    double q = d->qpos[m->jnt_qposadr[j]];
    for (int side = 0; side < 2; side++) {
      double dist = side == 0 ? q - m->jnt_range[2 * j] : m->jnt_range[2 * j + 1] - q;
      if (dist >= 0) continue;
      efc_reserve(m, d, r + 1);
      memset(d->efc_J + r * nv, 0, sizeof(double) * nv);
      d->efc_J[r * nv + m->jnt_dofadr[j]] = side == 0 ? 1 : -1;
      d->efc_pos[r] = dist; d->efc_margin[r] = 0; d->efc_type[r] = 0; d->efc_id[r] = j;
      double sr[2] = {fmax(m->jnt_solref[2 * j], 2 * m->timestep), m->jnt_solref[2 * j + 1]};
      finish_row(m, d, r, sr, m->jnt_solimp + 5 * j, m->dof_invweight0[m->jnt_dofadr[j]]);
      r++;
    }
  }
  /* contacts, pyramidal friction cone */
  double *jp1 = scratch(d, 18 * nv), *jr1 = jp1 + 3 * nv, *jp2 = jr1 + 3 * nv, *jr2 = jp2 + 3 * nv;
  double* Jc = jp1 + 12 * nv;
  for (int c = 0; c < d->ncon; c++) {
    Contact* con = d->contact + c;
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2], dim = con->dim;
    jac_point(m, d, b1, con->pos, jp1, jr1);
    jac_point(m, d, b2, con->pos, jp2, jr2);
    for (int k = 0; k < 3; k++)
      for (int i = 0; i < nv; i++) {
        double tp = 0, tr = 0;
        for (int a = 0; a < 3; a++) {
          tp += con->frame[3 * k + a] * (jp2[a * nv + i] - jp1[a * nv + i]);
          tr += con->frame[3 * k + a] * (jr2[a * nv + i] - jr1[a * nv + i]);
        }
        Jc[k * nv + i] = tp;
        Jc[(3 + k) * nv + i] = tr;
      }
    double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    double rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
    int nrow = dim == 1 ? 1 : 2 * (dim - 1);
    efc_reserve(m, d, r + nrow);
    con->efc_address = r;
    for (int e = 0; e < nrow; e++, r++) {
      int k = dim == 1 ? 0 : 1 + e / 2;
      double sgn = (e & 1) ? -1 : 1, mu = dim == 1 ? 0 : con->mu[k - 1];
      for (int i = 0; i < nv; i++) d->efc_J[r * nv + i] = Jc[i] + (dim == 1 ? 0 : sgn * mu * Jc[k * nv + i]);
      d->efc_pos[r] = con->dist; d->efc_margin[r] = con->margin; d->efc_type[r] = 1; d->efc_id[r] = c;
      double da = dim == 1 ? tran : tran + mu * mu * (k < 3 ? tran : rot);
      finish_row(m, d, r, con->solref, con->solimp, da);
    }
    if (dim > 1) { /* pyramidal: all edges share R = 2 mu^2 R_first (impratio = 1) */
      double Rpy = 2 * con->mu[0] * con->mu[0] * d->efc_R[con->efc_address];
      for (int e = 0; e < nrow; e++) d->efc_R[con->efc_address + e] = fmax(MINVAL, Rpy);
    }
  }
  d->nefc = r;
}

/* ------------------------------------------------------------------ 7. dual PGS */
static void solve_pgs(const OrcModel* m, OrcData* d) {
  int nv = m->nv, ne = d->nefc;
  memset(d->qfrc_constraint, 0, sizeof(double) * nv);
  memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  d->solver_iter = 0;
  if (!ne) return;
  for (int r = 0; r < ne; r++) {
    memcpy(d->efc_MinvJT + r * nv, d->efc_J + r * nv, sizeof(double) * nv);
    chol_solve(d->qL, d->efc_MinvJT + r * nv, nv);
    double b = -d->efc_aref[r];
    for (int i = 0; i < nv; i++) b += d->efc_J[r * nv + i] * d->qacc_smooth[i];
    d->efc_b[r] = b;
  }
  double* AR = d->efc_AR;
  for (int r = 0; r < ne; r++)
    for (int c = 0; c <= r; c++) {
      double s = 0;
      for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * d->efc_MinvJT[c * nv + i];
      AR[r * ne + c] = AR[c * ne + r] = s;
    }
  for (int r = 0; r < ne; r++) AR[r * ne + r] += d->efc_R[r];
  /* warmstart from qacc_warmstart; fall back to zero if that costs more */
  double cost = 0;
  for (int r = 0; r < ne; r++) {
    double jar = -d->efc_aref[r];
    for (int i = 0; i < nv; i++) jar += d->efc_J[r * nv + i] * d->qacc_warmstart[i];
    d->efc_force[r] = jar < 0 ? -jar / d->efc_R[r] : 0;
  }
  for (int r = 0; r < ne; r++) {
    double s = 0;
    for (int c = 0; c < ne; c++) s += AR[r * ne + c] * d->efc_force[c];
    cost += d->efc_force[r] * (0.5 * s + d->efc_b[r]);
  }
  if (cost > 0) memset(d->efc_force, 0, sizeof(double) * ne);
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  int it = 0;
  for (; it < m->iterations; it++) {
    double improvement = 0;
    for (int r = 0; r < ne; r++) {
      double res = d->efc_b[r];
      for (int c = 0; c < ne; c++) res += AR[r * ne + c] * d->efc_force[c];
      double old = d->efc_force[r], nw = fmax(0, old - res / AR[r * ne + r]), dl = nw - old;
      d->efc_force[r] = nw;
      improvement -= 0.5 * dl * dl * AR[r * ne + r] + dl * res;
    }
    if (improvement * scale < m->tolerance) { it++; break; }
  }
  d->solver_iter = it;
  for (int r = 0; r < ne; r++)
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * d->efc_force[r];
  double* t = scratch(d, nv);
  memcpy(t, d->qfrc_constraint, sizeof(double) * nv);
  chol_solve(d->qL, t, nv);
  for (int i = 0; i < nv; i++) d->qacc[i] = d->qacc_smooth[i] + t[i];
}


/* ------------------------------------------------------------------ 7'. primal Newton (MuJoCo's default solver)
 * minimise  1/2 (a-a_s)' M (a-a_s) + sum_i 1/2 D_i min(0, J_i a - aref_i)^2   over a = qacc,
 * Newton direction from H = M + J' diag(D*active) J, exact line search on the piecewise-quadratic
 * 1-D restriction (safeguarded Newton on phi'), termination on scaled improvement / gradient. */
static double primal_cost(const OrcModel* m, const OrcData* d, const double* a, double* x, double* Ma) {
  int nv = m->nv, ne = d->nefc;
  double cost = 0;
  for (int i = 0; i < nv; i++) {
    double s = 0;
    for (int j = 0; j < nv; j++) s += d->qM[i * nv + j] * (a[j] - d->qacc_smooth[j]);
    Ma[i] = s;
    cost += 0.5 * s * (a[i] - d->qacc_smooth[i]);
  }
  for (int r = 0; r < ne; r++) {
    double jar = -d->efc_aref[r];
    for (int i = 0; i < nv; i++) jar += d->efc_J[r * nv + i] * a[i];
    x[r] = jar;
    if (jar < 0) cost += 0.5 * jar * jar / d->efc_R[r];
  }
  return cost;
}
static void solve_newton(const OrcModel* m, OrcData* d) {
  int nv = m->nv, ne = d->nefc;
  memset(d->qfrc_constraint, 0, sizeof(double) * nv);
  memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  d->solver_iter = 0;
  if (!ne) return;
  double* w = scratch(d, 6 * nv + 2 * nv * nv + 2 * ne);
  double *a = w, *Ma = a + nv, *grad = Ma + nv, *p = grad + nv, *Mp = p + nv, *t = Mp + nv, *H = t + nv, *L = H + nv * nv;
  double *x = L + nv * nv, *jp = x + ne;
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  /* warmstart: the cheaper of qacc_warmstart and qacc_smooth */
  memcpy(a, d->qacc_warmstart, sizeof(double) * nv);
  double cost = primal_cost(m, d, a, x, Ma);
  double c0 = primal_cost(m, d, d->qacc_smooth, x, Ma);
  if (c0 < cost) memcpy(a, d->qacc_smooth, sizeof(double) * nv);
  cost = primal_cost(m, d, a, x, Ma);
  int it = 0;
  for (; it < m->iterations; it++) {
    memcpy(grad, Ma, sizeof(double) * nv);
    memcpy(H, d->qM, sizeof(double) * nv * nv);
    for (int r = 0; r < ne; r++) {
      if (x[r] >= 0) continue;
      double D = 1 / d->efc_R[r], f = -D * x[r];
      const double* J = d->efc_J + r * nv;
      for (int i = 0; i < nv; i++) {
        grad[i] -= J[i] * f;
        if (J[i] == 0) continue;
        for (int j = 0; j < nv; j++) H[i * nv + j] += D * J[i] * J[j];
      }
    }
    double gn = 0;
    for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
    if (sqrt(gn) * scale < m->tolerance) break;
    cholesky(L, H, nv);
    for (int i = 0; i < nv; i++) p[i] = -grad[i];
    chol_solve(L, p, nv);
    /* line search: phi(al) = cost(a + al p) */
    double pMp = 0, pMa = 0;
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int j = 0; j < nv; j++) s += d->qM[i * nv + j] * p[j];
      Mp[i] = s;
      pMp += p[i] * s;
      pMa += p[i] * Ma[i];
    }
    for (int r = 0; r < ne; r++) {
      double s = 0;
      for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * p[i];
      jp[r] = s;
    }
    double al = 0, lo = 0, hi = 1e300, d1 = 0, d2 = 0;
    for (int ls = 0; ls < m->ls_iterations; ls++) {
      d1 = pMa + al * pMp; d2 = pMp;
      for (int r = 0; r < ne; r++) {
        double xr = x[r] + al * jp[r];
        if (xr < 0) { double D = 1 / d->efc_R[r]; d1 += D * xr * jp[r]; d2 += D * jp[r] * jp[r]; }
      }
      if (ls > 0 && fabs(d1) * scale < m->ls_tolerance * m->tolerance) break;
      if (d1 < 0) lo = al; else hi = al;
      double nx = al - d1 / d2;
      if (!(nx > lo && nx < hi)) nx = hi < 1e299 ? 0.5 * (lo + hi) : 2 * al + 1;
      if (nx == al) break;
      al = nx;
    }
    for (int i = 0; i < nv; i++) a[i] += al * p[i];
    double newcost = primal_cost(m, d, a, x, Ma);
    double improvement = scale * (cost - newcost);
    cost = newcost;
    if (improvement < m->tolerance) { it++; break; }
  }
  d->solver_iter = it;
  memcpy(d->qacc, a, sizeof(double) * nv);
  for (int r = 0; r < ne; r++) {
    d->efc_force[r] = x[r] < 0 ? -x[r] / d->efc_R[r] : 0;
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * d->efc_force[r];
  }
}

/* ------------------------------------------------------------------ 8. touch sensors
 * (sensordata read by _get_touch, env_mujoco_util.py:470-475).  A contact counts when one of its geoms
 * is on the site's own body and the ray from the contact point along the normal, pointing away from
 * that body, meets the site volume (always true for a point inside it) [EXT: MuJoCo sensor/touch]. */
/* SITE_EPS: pad contacts routinely sit exactly on the lateral boundary of their site (the pad and its site have
 * the same footprint and a pad face lying inside the object's face contributes its own corners), where the
 * inclusive test is decided by rounding noise; 10 um of slack makes that case deterministic across precisions. */
#define SITE_EPS 1e-5
static int ray_hits_site(int type, const double* size, const double* p, const double* d) {
  double sz[3] = {size[0] + SITE_EPS, size[1] + SITE_EPS, size[2] + SITE_EPS};
  if (type == G_BOX) {
    double t0 = 0, t1 = 1e300;
    for (int i = 0; i < 3; i++) {
      if (fabs(d[i]) < MINVAL) { if (fabs(p[i]) > sz[i]) return 0; continue; }
      double a = (-sz[i] - p[i]) / d[i], b = (sz[i] - p[i]) / d[i];
      if (a > b) { double t = a; a = b; b = t; }
      if (a > t0) t0 = a;
      if (b < t1) t1 = b;
    }
    return t1 >= t0;
  }
  if (type == G_CYLINDER) { /* radius sz[0], half height sz[1], axis z */
    double t0 = 0, t1 = 1e300;
    if (fabs(d[2]) < MINVAL) { if (fabs(p[2]) > sz[1]) return 0; }
    else {
      double a = (-sz[1] - p[2]) / d[2], b = (sz[1] - p[2]) / d[2];
      if (a > b) { double t = a; a = b; b = t; }
      if (a > t0) t0 = a;
      if (b < t1) t1 = b;
    }
    double A = d[0] * d[0] + d[1] * d[1], B = p[0] * d[0] + p[1] * d[1], C = p[0] * p[0] + p[1] * p[1] - sz[0] * sz[0];
    if (A < MINVAL) { if (C > 0) return 0; }
    else {
      double disc = B * B - A * C;
      if (disc < 0) return 0;
      double sq = sqrt(disc), a = (-B - sq) / A, b = (-B + sq) / A;
      if (a > t0) t0 = a;
      if (b < t1) t1 = b;
    }
    return t1 >= t0;
  }
  /* sphere */
  double A = dot3(d, d), B = dot3(p, d), C = dot3(p, p) - sz[0] * sz[0];
  if (A < MINVAL) return C <= 0;
  double disc = B * B - A * C;
  if (disc < 0) return 0;
  return (-B + sqrt(disc)) / A >= 0;
}
static void touch_sensors(const OrcModel* m, OrcData* d) {
  for (int s = 0; s < m->nsensor; s++) {
    int site = m->sensor_siteid[s], sb = m->site_bodyid[site];
    double sum = 0;
    for (int c = 0; c < d->ncon; c++) {
      const Contact* con = d->contact + c;
      int on1 = m->geom_bodyid[con->geom1] == sb, on2 = m->geom_bodyid[con->geom2] == sb;
      if (!on1 && !on2) continue;
      int nrow = con->dim == 1 ? 1 : 2 * (con->dim - 1);
      double fn = 0;
      for (int e = 0; e < nrow; e++) fn += d->efc_force[con->efc_address + e];
      if (fn <= MINVAL) continue;
      double rel[3], l[3], ray[3], lr[3];
      sub3(rel, con->pos, d->site_xpos + 3 * site);
      mulmtv(l, d->site_xmat + 9 * site, rel);
      scl3(ray, con->frame, on2 ? -1.0 : 1.0);
      mulmtv(lr, d->site_xmat + 9 * site, ray);
      if (ray_hits_site(m->site_type[site], m->site_size + 3 * site, l, lr)) sum += fn;
    }
    d->sensordata[s] = sum;
  }
}

/* ------------------------------------------------------------------ forward / step */
void orc_forward(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  kinematics(m, d);
  com_pos_crb(m, d);
  cholesky(d->qL, d->qM, nv);
  collision(m, d);
  rne_bias(m, d);
  passive_actuation(m, d);
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  memcpy(d->qacc_smooth, d->qfrc_smooth, sizeof(double) * nv);
  chol_solve(d->qL, d->qacc_smooth, nv);
  make_constraint(m, d);
  if (m->solver == ORC_SOLVER_PGS) solve_pgs(m, d); else solve_newton(m, d);
  touch_sensors(m, d);
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

/* 9. semi-implicit Euler with implicit joint damping; quaternion integration for free joints */
static void integrate(const OrcModel* m, OrcData* d) {
  int nv = m->nv;
  double h = m->timestep;
  double* qacc = scratch(d, nv + 2 * nv * nv);
  int damped = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  if (damped) {
    double *Mh = qacc + nv, *L = Mh + nv * nv;
    memcpy(Mh, d->qM, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) Mh[i * nv + i] += h * m->dof_damping[i];
    cholesky(L, Mh, nv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    chol_solve(L, qacc, nv);
  } else {
    memcpy(qacc, d->qacc, sizeof(double) * nv);
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == J_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
      double w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]}, ang = h * norm3(w), dq[4], qn[4];
      if (ang > 0) {
        double ax[3];
        copy3(ax, w);
        normalize3(ax);
        axisangle2quat(dq, ax, ang);
        quat_mul(qn, d->qpos + qa + 3, dq);
        memcpy(d->qpos + qa + 3, qn, 32);
      }
      quat_normalize(d->qpos + qa + 3);
    } else {
      d->qpos[qa] += h * d->qvel[da];
    }
  }
}
void orc_step(const OrcModel* m, OrcData* d) {
  if (m->round_state >= 2) {
    /* control experiment: the STATE stays fp64, but everything the forward pass derives (kinematics, contacts, forces, qacc)
     * sees its fp32 rounding -- the ceiling for an fp32 engine that carries its state compensated (hi + lo floats): rounding
     * then perturbs each evaluation but never accumulates in the state */
    double q[64], v[64];
    if (m->nq > 64 || m->nv > 64) { fprintf(stderr, "orc_step: round_state >= 2 supports at most 64 coordinates (model has nq %d nv %d)\n", m->nq, m->nv); abort(); }
    /* variant 3 exempts the height of the LAST free body (the destination pedestal of jaco2_curtain_torque: qpos 18); a model without a
     * free joint has nothing to exempt */
    int exempt = -1;
    for (int b = 0; b < m->nbody; b++)
      for (int j = m->body_jntadr[b]; j >= 0 && j < m->body_jntadr[b] + m->body_jntnum[b]; j++)
        if (m->jnt_type[j] == J_FREE) exempt = m->jnt_qposadr[j] + 2;
    memcpy(q, d->qpos, sizeof(double) * m->nq);
    memcpy(v, d->qvel, sizeof(double) * m->nv);
    for (int i = 0; i < m->nq; i++) if (!(m->round_state == 3 && i == exempt)) d->qpos[i] = (double)(float)d->qpos[i];   /* (3: the pedestal's height keeps its exact value -- its bottom face starts exactly on the floor plane, see tools/drift_control.py variant D) */
    for (int i = 0; i < m->nv; i++) d->qvel[i] = (double)(float)d->qvel[i];
    orc_forward(m, d);
    memcpy(d->qpos, q, sizeof(double) * m->nq);
    memcpy(d->qvel, v, sizeof(double) * m->nv);
    integrate(m, d);
    return;
  }
  orc_forward(m, d);
  integrate(m, d);
  if (m->round_state == 1) { /* control experiment (tools/drift_control.py): what an fp32-state engine loses per step, nothing else */
    for (int i = 0; i < m->nq; i++) d->qpos[i] = (double)(float)d->qpos[i];
    for (int i = 0; i < m->nv; i++) { d->qvel[i] = (double)(float)d->qvel[i]; d->qacc_warmstart[i] = (double)(float)d->qacc_warmstart[i]; }
  }
}

/* ------------------------------------------------------------------ accessors */
int orc_ncon(const OrcData* d) { return d->ncon; }
int orc_nefc(const OrcData* d) { return d->nefc; }
int orc_solver_iter(const OrcData* d) { return d->solver_iter; }

int orc_set(const OrcModel* m, OrcData* d, const char* name, const double* src, int n) {
  double* dst = NULL; int len = 0;
  if (!strcmp(name, "qpos")) { dst = d->qpos; len = m->nq; }
  else if (!strcmp(name, "qvel")) { dst = d->qvel; len = m->nv; }
  else if (!strcmp(name, "ctrl")) { dst = d->ctrl; len = m->nu; }
  else if (!strcmp(name, "qacc_warmstart")) { dst = d->qacc_warmstart; len = m->nv; }
  else if (!strcmp(name, "mocap_pos")) { dst = d->mocap_pos; len = 3 * m->nmocap; }
  else if (!strcmp(name, "mocap_quat")) { dst = d->mocap_quat; len = 4 * m->nmocap; }
  if (!dst || n != len) return -1;
  memcpy(dst, src, sizeof(double) * n);
  return 0;
}
int orc_get(const OrcModel* m, const OrcData* d, const char* name, double* out, int n) {
  const double* src = NULL; int len = 0, nv = m->nv;
#define G(nm, p, l) else if (!strcmp(name, nm)) { src = p; len = l; }
  if (0) {}
  G("qpos", d->qpos, m->nq) G("qvel", d->qvel, nv) G("ctrl", d->ctrl, m->nu) G("qacc_warmstart", d->qacc_warmstart, nv)
  G("mocap_pos", d->mocap_pos, 3 * m->nmocap) G("mocap_quat", d->mocap_quat, 4 * m->nmocap)
  G("xpos", d->xpos, 3 * m->nbody) G("xquat", d->xquat, 4 * m->nbody) G("xmat", d->xmat, 9 * m->nbody)
  G("xipos", d->xipos, 3 * m->nbody) G("geom_xpos", d->geom_xpos, 3 * m->ngeom) G("geom_xmat", d->geom_xmat, 9 * m->ngeom)
  G("site_xpos", d->site_xpos, 3 * m->nsite) G("site_xmat", d->site_xmat, 9 * m->nsite)
  G("qM", d->qM, nv * nv) G("qfrc_bias", d->qfrc_bias, nv) G("qfrc_passive", d->qfrc_passive, nv)
  G("qfrc_actuator", d->qfrc_actuator, nv) G("qfrc_smooth", d->qfrc_smooth, nv) G("qacc_smooth", d->qacc_smooth, nv)
  G("qfrc_constraint", d->qfrc_constraint, nv) G("qacc", d->qacc, nv) G("actuator_force", d->actuator_force, m->nu)
  G("sensordata", d->sensordata, m->nsensor) G("efc_J", d->efc_J, d->nefc * nv) G("efc_force", d->efc_force, d->nefc)
  G("efc_aref", d->efc_aref, d->nefc) G("efc_R", d->efc_R, d->nefc) G("efc_pos", d->efc_pos, d->nefc)
  G("efc_b", d->efc_b, d->nefc) G("efc_diagApprox", d->efc_diagApprox, d->nefc)
#undef G
  if (!strcmp(name, "contact")) { /* per contact: dist, pos3, normal3, geom1, geom2, dim, efc_address = 11 */
    if (n != 11 * d->ncon) return -1;
    for (int c = 0; c < d->ncon; c++) {
      const Contact* k = d->contact + c;
      double* o = out + 11 * c;
      o[0] = k->dist; copy3(o + 1, k->pos); copy3(o + 4, k->frame); o[7] = k->geom1; o[8] = k->geom2; o[9] = k->dim; o[10] = k->efc_address;
    }
    return 0;
  }
  if (!src || n != len) return -1;
  memcpy(out, src, sizeof(double) * n);
  return 0;
}

void orc_step_batch(const OrcModel* m, int nenv, int nsub, double* qpos, double* qvel, double* qacc_ws, const double* ctrl,
                    double* sensordata, int nthreads) {
  orc_step_batch_stats(m, nenv, nsub, qpos, qvel, qacc_ws, ctrl, sensordata, nthreads, NULL);
}
/* same, and per env stats[4] = max contacts, max constraint rows, max solver iterations over the nsub steps, contacts of the last step */
void orc_step_batch_stats(const OrcModel* m, int nenv, int nsub, double* qpos, double* qvel, double* qacc_ws, const double* ctrl,
                          double* sensordata, int nthreads, int* stats) {
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
  {
    OrcData* d = orc_make_data(m);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int e = 0; e < nenv; e++) {
      memcpy(d->qpos, qpos + (long)e * m->nq, sizeof(double) * m->nq);
      memcpy(d->qvel, qvel + (long)e * m->nv, sizeof(double) * m->nv);
      memcpy(d->qacc_warmstart, qacc_ws + (long)e * m->nv, sizeof(double) * m->nv);
      memcpy(d->ctrl, ctrl + (long)e * m->nu, sizeof(double) * m->nu);
      int mc = 0, me = 0, mi = 0;
      for (int s = 0; s < nsub; s++) {
        orc_step(m, d);
        if (d->ncon > mc) mc = d->ncon;
        if (d->nefc > me) me = d->nefc;
        if (d->solver_iter > mi) mi = d->solver_iter;
      }
      if (stats) { stats[4 * e] = mc; stats[4 * e + 1] = me; stats[4 * e + 2] = mi; stats[4 * e + 3] = d->ncon; }
      memcpy(qpos + (long)e * m->nq, d->qpos, sizeof(double) * m->nq);
      memcpy(qvel + (long)e * m->nv, d->qvel, sizeof(double) * m->nv);
      memcpy(qacc_ws + (long)e * m->nv, d->qacc_warmstart, sizeof(double) * m->nv);
      if (sensordata) memcpy(sensordata + (long)e * m->nsensor, d->sensordata, sizeof(double) * m->nsensor);
    }
    orc_free_data(d);
  }
  (void)nthreads;
}
