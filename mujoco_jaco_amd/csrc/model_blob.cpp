// Host-side: JACOMDL1 blob (fused view, f_* arrays) -> JacoModelDev + hull vertex table.
// No HIP dependency so the same loader serves the product library and the CPU-side kernel checks.
#include "model_blob.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>

namespace {
struct Arr { int code; int count; const char* data; };

struct Blob {
  std::map<std::string, Arr> a;
  std::string err;
  bool parse(const void* buf, size_t size) {
    const char* p = (const char*)buf;
    if (size < 16 || memcmp(p, "JACOMDL1", 8)) { err = "bad magic"; return false; }
    int n = *(const int32_t*)(p + 8);
    size_t off = 16;
    for (int i = 0; i < n; i++) {
      if (off + 40 > size) { err = "truncated header"; return false; }
      std::string name(p + off, strnlen(p + off, 32));
      int code = *(const int32_t*)(p + off + 32), count = *(const int32_t*)(p + off + 36);
      off += 40;
      size_t nb = (size_t)count * (code == 0 ? 8 : 4);
      if (off + nb > size) { err = "truncated payload: " + name; return false; }
      a[name] = Arr{code, count, p + off};
      off += nb + ((8 - nb % 8) % 8);
    }
    return true;
  }
  const double* f64(const char* n, int count = -1) {
    auto it = a.find(n);
    if (it == a.end() || it->second.code != 0 || (count >= 0 && it->second.count != count)) {
      if (err.empty()) err = std::string("missing/mis-sized f64 array ") + n;
      return nullptr;
    }
    return (const double*)it->second.data;
  }
  const int32_t* i32(const char* n, int count = -1) {
    auto it = a.find(n);
    if (it == a.end() || it->second.code != 1 || (count >= 0 && it->second.count != count)) {
      if (err.empty()) err = std::string("missing/mis-sized i32 array ") + n;
      return nullptr;
    }
    return (const int32_t*)it->second.data;
  }
  int count(const char* n) { auto it = a.find(n); return it == a.end() ? -1 : it->second.count; }
  bool has(const char* n) { return a.count(n) != 0; }
};

void quat2mat(const double* q, float* M) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = (float)(w * w + x * x - y * y - z * z); M[1] = (float)(2 * (x * y - w * z)); M[2] = (float)(2 * (x * z + w * y));
  M[3] = (float)(2 * (x * y + w * z)); M[4] = (float)(w * w - x * x + y * y - z * z); M[5] = (float)(2 * (y * z - w * x));
  M[6] = (float)(2 * (x * z - w * y)); M[7] = (float)(2 * (y * z + w * x)); M[8] = (float)(w * w - x * x - y * y + z * z);
}
void cp3(float* d, const double* s) { d[0] = (float)s[0]; d[1] = (float)s[1]; d[2] = (float)s[2]; }
}  // namespace

int jaco_model_from_blob(const void* buf, size_t size, JacoModelDev* m, std::vector<float>* hull, std::string* error) {
  Blob B;
  memset(m, 0, sizeof(*m));
#define FAIL(msg) do { *error = (msg); return -1; } while (0)
  if (!B.parse(buf, size)) FAIL(B.err);
  const int32_t *pnb = B.i32("f_nbody", 1), *pnv = B.i32("nv", 1), *pnq = B.i32("nq", 1), *pnu = B.i32("nu", 1);
  const int32_t *png = B.i32("f_ngeom", 1), *pnp = B.i32("f_npair", 1), *pns = B.i32("f_nsensor", 1), *pnm = B.i32("nmocap", 1);
  if (!pnb || !pnv || !pnq || !pnu || !png || !pnp || !pns || !pnm) FAIL(B.err);
  int nb = *pnb, nv = *pnv, nq = *pnq, nu = *pnu, ng = *png, np = *pnp, ns = *pns;
  if (nb > JNB || nv > JNV || nq > JNQ || nu > JNU || ng > JMAXGEOM || np > JMAXPAIR || ns > JNSENS || *pnm > JNMOCAP)
    FAIL("model exceeds compiled capacities (JNB/JNV/JMAXGEOM/JMAXPAIR/...)");
  m->nbody = nb; m->nv = nv; m->nq = nq; m->nu = nu; m->ngeom = ng; m->npair = np; m->nsensor = ns; m->nmocap = *pnm;
  const double *ts = B.f64("opt_timestep", 1), *gr = B.f64("opt_gravity", 3), *tol = B.f64("opt_tolerance", 1);
  const double *mi = B.f64("meaninertia", 1), *mt = B.f64("opt_mpr_tolerance", 1);
  const int32_t *it = B.i32("opt_iterations", 1), *mpi = B.i32("opt_mpr_iterations", 1);
  if (!ts || !gr || !tol || !mi || !mt || !it || !mpi) FAIL(B.err);
  m->timestep = (float)*ts; m->timestep_lo = (float)(*ts - (double)m->timestep); m->compensated = 1; cp3(m->gravity, gr); m->tolerance = (float)*tol; m->meaninertia = (float)*mi;
  m->mpr_tolerance = (float)*mt; m->iterations = *it; m->mpr_iterations = *mpi; m->ls_iterations = 50; m->ls_tolerance = 0.01f; m->mpr_output = 1;

  // ---- bodies
  const int32_t *par = B.i32("f_parent", nb), *jt = B.i32("f_jtype", nb), *qa = B.i32("f_qposadr", nb), *da = B.i32("f_dofadr", nb);
  const int32_t* lim = B.i32("f_limited", nb);
  const double *pos = B.f64("f_pos", 3 * nb), *quat = B.f64("f_quat", 4 * nb), *axis = B.f64("f_axis", 3 * nb), *jpos = B.f64("f_jpos", 3 * nb);
  const double *q0 = B.f64("f_qpos0", nb), *mass = B.f64("f_mass", nb), *com = B.f64("f_com", 3 * nb), *inr = B.f64("f_inertia", 6 * nb);
  const double *rng = B.f64("f_range", 2 * nb), *jsr = B.f64("jnt_solref", 2 * nb), *jsi = B.f64("jnt_solimp", 5 * nb);
  if (!jsr || !jsi) FAIL(B.err);
  if (!par || !jt || !qa || !da || !lim || !pos || !quat || !axis || !jpos || !q0 || !mass || !com || !inr || !rng) FAIL(B.err);
  for (int b = 0; b < nb; b++) {
    if (par[b] >= b) FAIL("body parents must precede children");
    if (jpos[3 * b] != 0 || jpos[3 * b + 1] != 0 || jpos[3 * b + 2] != 0) FAIL("joint anchors off the body origin are not supported");
    if (jt[b] != JJ_FREE && jt[b] != JJ_HINGE) FAIL("unsupported joint type");
    if (jt[b] == JJ_FREE && par[b] != -1) FAIL("free joint must hang off the world");
    m->b_parent[b] = par[b]; m->b_jtype[b] = jt[b]; m->b_qadr[b] = qa[b]; m->b_dadr[b] = da[b]; m->b_limited[b] = lim[b];
    cp3(m->b_pos[b], pos + 3 * b); quat2mat(quat + 4 * b, m->b_mat[b]); cp3(m->b_axis[b], axis + 3 * b);
    m->b_qpos0[b] = (float)q0[b]; m->b_qpos0_lo[b] = (float)(q0[b] - (double)m->b_qpos0[b]); m->b_mass[b] = (float)mass[b]; cp3(m->b_com[b], com + 3 * b);
    for (int k = 0; k < 6; k++) m->b_inertia[b][k] = (float)inr[6 * b + k];
    m->b_range[b][0] = (float)rng[2 * b]; m->b_range[b][1] = (float)rng[2 * b + 1];
    m->b_solref[b][0] = (float)fmax(jsr[2 * b], 2 * *ts); m->b_solref[b][1] = (float)jsr[2 * b + 1];  // refsafe
    for (int k = 0; k < 5; k++) m->b_solimp[b][k] = (float)jsi[5 * b + k];
  }
  // ---- dofs
  const int32_t* dpar = B.i32("dof_parentid", nv);
  const double *ddamp = B.f64("dof_damping", nv), *diw = B.f64("dof_invweight0", nv);
  if (!dpar || !ddamp || !diw) FAIL(B.err);
  for (int b = 0; b < nb; b++) {
    int n = jt[b] == JJ_FREE ? 6 : 1;
    for (int k = 0; k < n; k++) m->d_body[da[b] + k] = b;
  }
  for (int d = 0; d < nv; d++) {
    m->d_parent[d] = dpar[d]; m->d_damping[d] = (float)ddamp[d]; m->d_invweight[d] = (float)diw[d];
    if (ddamp[d] > 0) m->has_damping = (d < 6 || m->has_damping == 2) ? 2 : 1;   // 1: only the finger joints (dofs >= 6), see ldl_block0_dual
    if (ddamp[d] > 0 && d >= JB0) FAIL("joint damping outside the arm/finger dof block is not supported by the kernels");
    m->d_qadr[d] = -1;
  }
  {   // joint springs (optional arrays: blobs compiled before round 3 have none)
    const double *fst = B.has("f_stiffness") ? B.f64("f_stiffness", nb) : nullptr, *fsr = B.has("f_springref") ? B.f64("f_springref", nb) : nullptr;
    for (int b = 0; b < nb; b++) {
      if (jt[b] != JJ_HINGE) continue;
      m->d_qadr[da[b]] = qa[b];
      if (fst && fsr && fst[b] != 0) { m->d_stiffness[da[b]] = (float)fst[b]; m->d_springref[da[b]] = (float)fsr[b]; m->has_springs = 1; }
    }
  }
  for (int b = 0; b < nb; b++) {   // the kernels' block-diagonal solves assume this dof layout (one block per kinematic tree)
    int blk = da[b] < JB0 ? 0 : (da[b] < JB1 ? 1 : 2), root = b;
    while (par[root] >= 0) root = par[root];
    int rblk = da[root] < JB0 ? 0 : (da[root] < JB1 ? 1 : 2);
    int n = jt[b] == JJ_FREE ? 6 : 1;
    if (blk != rblk || (da[b] + n > JB0 && da[b] < JB0) || (da[b] + n > JB1 && da[b] < JB1)) FAIL("dof layout does not match this build's dof blocks [0,JB0) [JB0,JB1) [JB1,JNV) (jaco/model_dev.h)");
  }
  for (int b = 0; b < nb; b++) {
    unsigned mask = par[b] >= 0 ? m->b_chainmask[par[b]] : 0u;
    int n = jt[b] == JJ_FREE ? 6 : 1;
    for (int k = 0; k < n; k++) mask |= 1u << (da[b] + k);
    m->b_chainmask[b] = mask;
    if (__builtin_popcount(mask) > JMAXCHAIN) FAIL("a body is moved by more dofs than JMAXCHAIN (jaco/model_dev.h)");
  }
  // ---- subtree tables for the composite-inertia / mass-matrix stages
  int ninner = 0;
  for (int b = 0; b < nb; b++) {
    unsigned desc = 0;   // bit x: body x lies in the subtree rooted at b (b included)
    for (int x = 0; x < nb; x++)
      for (int y = x; y >= 0; y = par[y]) if (y == b) { desc |= 1u << x; break; }
    m->b_descmask[b] = desc;
    if (__builtin_popcount(desc) > JMAXDESC) FAIL("a subtree holds more bodies than JMAXDESC (jaco/model_dev.h)");
    if (desc != (1u << b)) {
      if (ninner >= JMAXINNER) FAIL("too many bodies with children for the kernels' subtree stage");
      m->inner_body[ninner++] = b;
    }
  }
  m->ninner = ninner;
  int nmp = 0;   // structurally non-zero lower-triangle entries of the mass matrix: (dof d, ancestor-or-self dof j)
  for (int d = 0; d < nv; d++)
    for (int j = d; j >= 0; j = dpar[j]) {
      if (nmp >= JMAXMPAIR) FAIL("too many mass-matrix entries for the kernels' pair table");
      m->mpair[nmp++] = d | (j << 8);
    }
  m->nmpair = nmp;
  // ---- actuators
  const int32_t *ajnt = B.i32("actuator_jntid", nu), *apos = B.i32("actuator_position", nu), *acl = B.i32("actuator_ctrllimited", nu);
  const int32_t *afl = B.i32("actuator_forcelimited", nu), *jq = B.i32("jnt_qposadr"), *jd = B.i32("jnt_dofadr");
  const double *akp = B.f64("actuator_kp", nu), *acr = B.f64("actuator_ctrlrange", 2 * nu), *afr = B.f64("actuator_forcerange", 2 * nu);
  if (!ajnt || !apos || !acl || !afl || !jq || !jd || !akp || !acr || !afr) FAIL(B.err);
  unsigned seen = 0;
  for (int a = 0; a < nu; a++) {
    m->a_dof[a] = jd[ajnt[a]]; m->a_qadr[a] = jq[ajnt[a]]; m->a_position[a] = apos[a];
    if (seen & (1u << m->a_dof[a])) FAIL("two actuators on one dof are not supported");
    seen |= 1u << m->a_dof[a];
    m->a_ctrllimited[a] = acl[a]; m->a_forcelimited[a] = afl[a]; m->a_kp[a] = (float)akp[a];
    m->a_ctrlrange[a][0] = (float)acr[2 * a]; m->a_ctrlrange[a][1] = (float)acr[2 * a + 1];
    m->a_forcerange[a][0] = (float)afr[2 * a]; m->a_forcerange[a][1] = (float)afr[2 * a + 1];
  }
  // ---- geoms
  const int32_t *gb = B.i32("f_geom_body", ng), *gty = B.i32("f_geom_type", ng), *gva = B.i32("f_geom_vertadr", ng), *gvn = B.i32("f_geom_vertnum", ng);
  const int32_t *gob = B.i32("f_geom_origbody", ng), *gmo = B.i32("f_geom_mocap", ng);
  const double *gp = B.f64("f_geom_pos", 3 * ng), *gq = B.f64("f_geom_quat", 4 * ng), *gs = B.f64("f_geom_size", 3 * ng);
  const double *grb = B.f64("f_geom_rbound", ng), *giw = B.f64("f_geom_invweight", 2 * ng);
  const double *mp0 = B.f64("mocap_pos0"), *mq0 = B.f64("mocap_quat0");
  if (!gb || !gty || !gva || !gvn || !gob || !gmo || !gp || !gq || !gs || !grb || !giw || !mp0 || !mq0) FAIL(B.err);
  const int32_t* mk = B.has("f_marker_mocap") ? B.i32("f_marker_mocap", 2) : nullptr;
  for (int k = 0; k < 2; k++) {
    for (int i = 0; i < 12; i++) m->marker_rest[k][i] = 0.f;
    m->marker_rest[k][3] = m->marker_rest[k][7] = m->marker_rest[k][11] = 1.f;
    if (mk && mk[k] >= 0) { cp3(m->marker_rest[k], mp0 + 3 * mk[k]); quat2mat(mq0 + 4 * mk[k], m->marker_rest[k] + 3); }
  }
  for (int g = 0; g < ng; g++) {
    m->g_marker[g] = -1;
    m->g_body[g] = gb[g]; m->g_type[g] = gty[g]; m->g_vertadr[g] = gva[g]; m->g_vertnum[g] = gvn[g];
    m->g_origbody[g] = gob[g]; m->g_mocap[g] = gmo[g];
    cp3(m->g_pos[g], gp + 3 * g); quat2mat(gq + 4 * g, m->g_mat[g]); cp3(m->g_size[g], gs + 3 * g);
    m->g_rbound[g] = (float)grb[g]; m->g_invweight[g][0] = (float)giw[2 * g]; m->g_invweight[g][1] = (float)giw[2 * g + 1];
    if (gmo[g] >= 0) {  // marker geoms ride on mocap bodies: compose with the XML mocap pose (static until the task layer moves them)
      float R[9], lp[3] = {m->g_pos[g][0], m->g_pos[g][1], m->g_pos[g][2]}, lm[9];
      memcpy(lm, m->g_mat[g], sizeof lm);
      for (int k = 0; k < 2; k++)
        if (mk && mk[k] == gmo[g]) { m->g_marker[g] = k; memcpy(m->g_lpos[g], lp, sizeof lp); memcpy(m->g_lmat[g], lm, sizeof lm); }
      quat2mat(mq0 + 4 * gmo[g], R);
      for (int i = 0; i < 3; i++) {
        m->g_pos[g][i] = (float)mp0[3 * gmo[g] + i] + R[3 * i] * lp[0] + R[3 * i + 1] * lp[1] + R[3 * i + 2] * lp[2];
        for (int j = 0; j < 3; j++) m->g_mat[g][3 * i + j] = R[3 * i] * lm[j] + R[3 * i + 1] * lm[3 + j] + R[3 * i + 2] * lm[6 + j];
      }
    }
  }
  // ---- pairs
  const int32_t *pg = B.i32("f_pair_geom", 2 * np), *pd = B.i32("f_pair_condim", np);
  const double *pmu = B.f64("f_pair_mu", 5 * np), *pref = B.f64("f_pair_solref", 2 * np), *pimp = B.f64("f_pair_solimp", 5 * np), *pmar = B.f64("f_pair_margin", np);
  if (!pg || !pd || !pmu || !pref || !pimp || !pmar) FAIL(B.err);
  for (int k = 0; k < np; k++) {
    JacoPairParam& P = m->pair[k];
    int g1 = pg[2 * k], g2 = pg[2 * k + 1];
    if (gty[g1] > gty[g2]) { int t = g1; g1 = g2; g2 = t; }
    P.g1 = g1; P.g2 = g2; P.condim = pd[k]; P.margin = (float)pmar[k];
    P.tran = m->g_invweight[g1][0] + m->g_invweight[g2][0]; P.rot = m->g_invweight[g1][1] + m->g_invweight[g2][1];
    for (int i = 0; i < 5; i++) { P.mu[i] = (float)pmu[5 * k + i]; P.solimp[i] = (float)pimp[5 * k + i]; }
    P.solref[0] = (float)pref[2 * k]; P.solref[1] = (float)pref[2 * k + 1];
    m->pair_code[k] = g1 | (g2 << 8) | (gty[g1] << 16) | (gty[g2] << 20);
    {
      JacoPairObb& Q = m->pair_obb[k];
      Q.code = m->pair_code[k]; Q.pad = 0;
      for (int i = 0; i < 3; i++) {
        // box half sizes for the OBB cull: a sphere's (r, r, r), a cylinder's (r, r, h)
        Q.sa[i] = gty[g1] == JG_SPHERE ? m->g_size[g1][0] : (gty[g1] == JG_CYLINDER ? m->g_size[g1][i < 2 ? 0 : 1] : m->g_size[g1][i]);
        Q.sb[i] = gty[g2] == JG_SPHERE ? m->g_size[g2][0] : (gty[g2] == JG_CYLINDER ? m->g_size[g2][i < 2 ? 0 : 1] : m->g_size[g2][i]);
      }
    }
    const int b1 = m->g_body[g1], b2 = m->g_body[g2];
    P.m1 = b1 >= 0 ? m->b_chainmask[b1] : 0u; P.m2 = b2 >= 0 ? m->b_chainmask[b2] : 0u;
    P.ob = m->g_origbody[g1] | ((b1 + 1) << 8) | (m->g_origbody[g2] << 16) | ((b2 + 1) << 24);
  }
  m->plane_chunks = 0;   // (no assumption on where the plane pairs sit: the last one decides)
  for (int k = 0; k < np; k++) if (((m->pair_code[k] >> 16) & 15) == JG_PLANE) m->plane_chunks = k / 64 + 1;
  // ---- touch sites
  const int32_t *sb = B.i32("f_site_body", ns), *sty = B.i32("f_site_type", ns), *sob = B.i32("f_site_origbody", ns);
  const double *sp = B.f64("f_site_pos", 3 * ns), *sq = B.f64("f_site_quat", 4 * ns), *ssz = B.f64("f_site_size", 3 * ns);
  if (!sb || !sty || !sob || !sp || !sq || !ssz) FAIL(B.err);
  for (int s = 0; s < ns; s++) {
    m->s_body[s] = sb[s]; m->s_type[s] = sty[s]; m->s_origbody[s] = sob[s];
    if (sob[s] < 0 || sob[s] >= 128) FAIL("touch site on a body id outside [0, 128)");
    m->sens_bodymask[sob[s] >> 5] |= 1u << (sob[s] & 31);
    cp3(m->s_pos[s], sp + 3 * s); quat2mat(sq + 4 * s, m->s_mat[s]); cp3(m->s_size[s], ssz + 3 * s);
  }
  // ---- named frames
  auto frame = [&](const char* nm, int* body, float* p, float* R) {
    const double* f = B.has(nm) ? B.f64(nm, 8) : nullptr;
    if (!f) { *body = -1; return; }
    *body = (int)f[0];
    if (p) cp3(p, f + 1);
    if (R) quat2mat(f + 4, R);
  };
  int dummy;
  frame("f_frame_EE", &m->ee_body, m->ee_pos, m->ee_mat);
  frame("f_frame_EE_obj", &m->eeobj_body, m->eeobj_pos, m->eeobj_mat);
  frame("f_frame_link1", &dummy, m->base_pos, nullptr);
  if (dummy >= 0) { /* link1 origin sits on joint0's axis: world position is constant = its qpos0 pose */
    const double* f = B.f64("f_frame_link1", 8);
    (void)f;
    m->base_pos[0] = m->b_pos[0][0]; m->base_pos[1] = m->b_pos[0][1]; m->base_pos[2] = m->b_pos[0][2];
  }
  frame("f_frame_object_body", &m->obj_body, nullptr, nullptr);
  frame("f_frame_object_dest", &m->dest_body, nullptr, nullptr);
  // ---- hull vertices (float4, w = 0)
  const double* mv = B.f64("mesh_vert");
  if (!mv) FAIL(B.err);
  int nvert = B.count("mesh_vert") / 3;
  m->nhullvert = nvert;
  hull->assign((size_t)4 * nvert, 0.f);
  for (int i = 0; i < nvert; i++) for (int k = 0; k < 3; k++) (*hull)[4 * i + k] = (float)mv[3 * i + k];
  // ---- direction-indexed support tables (modelc/supportmap.py), appended to the vertex buffer: per cube-map cell 64 slots
  // of (x, y, z, vertex index as raw bits; -1 = empty).  g_celladr is in float4 units from the start of the buffer.
  for (int g = 0; g < ng; g++) { m->g_cellR[g] = 0; m->g_celladr[g] = 0; }
  if (B.has("f_hullmap_cnt") && B.has("f_geom_cellR")) {
    const int32_t *cr = B.i32("f_geom_cellR", ng), *c0 = B.i32("f_geom_cell0", ng), *cnt = B.i32("f_hullmap_cnt"), *ids = B.i32("f_hullmap_ids");
    if (!cr || !c0 || !cnt || !ids) FAIL(B.err);
    const int ncell = B.count("f_hullmap_cnt"), nid = B.count("f_hullmap_ids");
    std::vector<int> start(ncell + 1, 0);
    for (int c = 0; c < ncell; c++) { if (cnt[c] < 0 || cnt[c] > 64) FAIL("support table: bad cell count"); start[c + 1] = start[c] + cnt[c]; }
    if (start[ncell] > nid) FAIL("support table: index list too short");
    const size_t base = hull->size() / 4;   // first table entry, in float4 units
    bool any = false;
    for (int g = 0; g < ng; g++) any = any || cr[g] > 0;
    if (any) {
      hull->resize(4 * (base + (size_t)ncell * 64), 0.f);
      for (int g = 0; g < ng; g++) {
        if (cr[g] <= 0) continue;
        const int cells = 6 * cr[g] * cr[g];
        if (c0[g] < 0 || c0[g] + cells > ncell) FAIL("support table: cell range outside the table");
        m->g_cellR[g] = cr[g]; m->g_celladr[g] = (int)(base + (size_t)c0[g] * 64);
        for (int c = c0[g]; c < c0[g] + cells; c++)
          for (int j = 0; j < 64; j++) {
            float* e = hull->data() + 4 * (base + (size_t)c * 64 + j);
            int32_t id = j < cnt[c] ? ids[start[c] + j] : -1;
            if (id >= m->g_vertnum[g]) FAIL("support table: vertex index outside the hull");
            if (id >= 0) for (int k = 0; k < 3; k++) e[k] = (*hull)[4 * (size_t)(m->g_vertadr[g] + id) + k];
            memcpy(e + 3, &id, 4);
          }
      }
    }
  }
  if (!B.err.empty()) FAIL(B.err);
#undef FAIL
  return 0;
}
