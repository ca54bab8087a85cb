// Collision + contact rows + touch sensors for the one-wavefront-per-env physics kernel.
// Included by physics_kernel.h.  Restates, per environment, MuJoCo's collision pipeline as the
// reference runs it inside sim.step() (mujoco.py:278):
//   static pair whitelist -> bounding-sphere test -> (cull only) OBB test -> narrowphase:
//   plane-box / plane-sphere / plane-hull analytic, box-box SAT + face polygon intersection,
//   everything else (hull meshes, sphere) through Minkowski Portal Refinement on support functions.
// Hull support queries scan the hull vertices with all 64 lanes and reduce with a DPP argmax.
#pragma once

#define JACO_HAVE_COLLISION 1

struct GeomPose { v3 p; m3 R; };
template <class L>
JDEV GeomPose geom_pose(const L& s, int g) { GeomPose r; r.p = ld3(s.gpos[g]); r.R = ldm(s.gmat[g]); return r; }

// conservative oriented-box separation test (cull only: may say "not separated" for separated boxes, never the reverse)
JDEV bool obb_separated(const GeomPose& a, v3 sa, const GeomPose& b, v3 sb) {
  float C[3][3], Q[3][3];
  v3 d = b.p - a.p;
  float t[3] = {dot(d, col(a.R, 0)), dot(d, col(a.R, 1)), dot(d, col(a.R, 2))};
  float ea[3] = {sa.x, sa.y, sa.z}, eb[3] = {sb.x, sb.y, sb.z};
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { C[i][j] = dot(col(a.R, i), col(b.R, j)); Q[i][j] = fabsf(C[i][j]) + 1e-5f; }
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) sep |= fabsf(t[i]) > ea[i] + eb[0] * Q[i][0] + eb[1] * Q[i][1] + eb[2] * Q[i][2];
#pragma unroll
  for (int j = 0; j < 3; j++)
    sep |= fabsf(t[0] * C[0][j] + t[1] * C[1][j] + t[2] * C[2][j]) > eb[j] + ea[0] * Q[0][j] + ea[1] * Q[1][j] + ea[2] * Q[2][j];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      float ra = ea[i1] * Q[i2][j] + ea[i2] * Q[i1][j], rb = eb[j1] * Q[i][j2] + eb[j2] * Q[i][j1];
      sep |= fabsf(t[i2] * C[i1][j] - t[i1] * C[i2][j]) > ra + rb;
    }
  return sep;
}

JDEV void make_frame(v3 n, float* fr) {
  n = normalized(n);
  v3 y = (n.y < 0.5f && n.y > -0.5f) ? mk3(0.f, 1.f, 0.f) : mk3(0.f, 0.f, 1.f);
  y = normalized(y - n * dot(n, y));
  v3 z = cross(n, y);
  st3(fr, n); st3(fr + 3, y); st3(fr + 6, z);
}

// Append contacts held by lanes with `have` set (lane order), all lanes must call.
template <class L>
JDEV void push_contacts(L& s, bool have, float dist, v3 pos, v3 normal, int pair, int& ncon, unsigned& flags, int limit) {
  constexpr int MAXCON = L::Caps::MAXCON;
  unsigned long long mask = wave_ballot(have);
  int idx = wave_prefix_count(mask), n = popc64(mask);
  if (n > limit) n = limit;
  bool keep = have && idx < limit;
  if (keep && ncon + idx < MAXCON) {
    int c = ncon + idx;
    s.c_dist[c] = dist;
    st3(s.c_pos[c], pos);
    make_frame(normal, s.c_frame[c]);
    s.c_pair[c] = pair;
  }
  if (ncon + n > MAXCON) { flags |= JFLAG_CON_OVERFLOW; n = MAXCON - ncon; }
  ncon += n;
}

// ---------------------------------------------------------------- support functions (all lanes, wave-uniform direction)
template <class L>
JDEV v3 support_geom(const JacoStepArgs& A, const JacoModelDev* m, const L& s, int g, int type, v3 dir, int lane) {
  GeomPose P = geom_pose(s, g);
  v3 l = mulT(P.R, dir), sp;
  if (type == JG_BOX) {
    sp = mk3(l.x > 0.f ? m->g_size[g][0] : -m->g_size[g][0], l.y > 0.f ? m->g_size[g][1] : -m->g_size[g][1], l.z > 0.f ? m->g_size[g][2] : -m->g_size[g][2]);
  } else if (type == JG_SPHERE) {
    float n = norm(l);
    sp = l * (n > JMINVAL ? m->g_size[g][0] / n : 0.f);
  } else {  // hull mesh: 64-lane scan + argmax (lowest vertex index wins ties, like a serial first-max scan)
    int adr = m->g_vertadr[g], nvert = m->g_vertnum[g];
    float best = -3.0e38f;
    int bi = 0x7fffffff;
    for (int i = lane; i < nvert; i += 64) {
      const float* v = A.hull + 4 * (size_t)(adr + i);
      float t = v[0] * l.x + v[1] * l.y + v[2] * l.z;
      if (t > best) { best = t; bi = i; }
    }
    float bv;
    int win = wave_argmax(best, bi, &bv);
    win = win < nvert ? win : 0;   // (a non-finite direction beats nothing: stay inside the table; the env is quarantined by the epilogue)
    const float* v = A.hull + 4 * (size_t)(adr + win);
    sp = mk3(v[0], v[1], v[2]);
  }
  return P.p + mul(P.R, sp);
}

// ---------------------------------------------------------------- plane narrowphase
template <class L>
JDEV void collide_plane_box(v3 size, L& s, int g1, int g2, int pair, int lane, int& ncon, unsigned& flags) {
  GeomPose P = geom_pose(s, g1), B = geom_pose(s, g2);
  v3 n = col(P.R, 2);
  int i = lane & 7;
  v3 l = mk3((i & 1) ? size.x : -size.x, (i & 2) ? size.y : -size.y, (i & 4) ? size.z : -size.z);
  v3 c = B.p + mul(B.R, l);
  float dist = dot(c - P.p, n);
  push_contacts(s, lane < 8 && !(dist > 0.f), dist, c - n * (0.5f * dist), n, pair, ncon, flags, 4);
}
template <class L>
JDEV void collide_plane_sphere(const JacoModelDev* m, L& s, int g1, int g2, int pair, int lane, int& ncon, unsigned& flags) {
  GeomPose P = geom_pose(s, g1);
  v3 n = col(P.R, 2), c = ld3(s.gpos[g2]);
  float r = m->g_size[g2][0], dist = dot(c - P.p, n) - r;
  push_contacts(s, lane == 0 && !(dist > 0.f), dist, c - n * (r + 0.5f * dist), n, pair, ncon, flags, 1);
}
template <class L>
JDEV void collide_plane_convex(const JacoStepArgs& A, const JacoModelDev* m, L& s, int g1, int g2, int t2, int pair, int lane, int& ncon, unsigned& flags) {
  GeomPose P = geom_pose(s, g1);
  v3 n = col(P.R, 2);
  v3 sp = support_geom(A, m, s, g2, t2, -n, lane);
  float dist = dot(sp - P.p, n);
  push_contacts(s, lane == 0 && !(dist > 0.f), dist, sp - n * (0.5f * dist), n, pair, ncon, flags, 1);
}

// ---------------------------------------------------------------- box-box
// Up to four pairs per pass, one 16-lane row each (the EE's axis sticks on the hand marker's sticks alone are nine pairs, and they come
// in a row in pair order): row lane a < 15 tests one separating axis (a < 6: face axes, else edge axis A[i] x B[j]); the clipping
// candidates of a face contact are produced in two rounds, lanes 0..3 incident vertices + lanes 4..7 reference corners, then all 16
// lanes one edge crossing each.  Contacts come out row by row, within a row in candidate order (vertices, corners, crossings).
// Row r handles candidate cbase + r of the list; lane c of the A-arrays holds candidate c's pair index, code and contact bookkeeping.
template <class L>
JDEV void collide_box_box4(const JacoModelDev* m, L& s, int cbase, int nrows, int pkA, int codeA, unsigned m1A, unsigned m2A, int obA, int dimA,
                           v3 saA, v3 sbA, int lane, int& ncon, unsigned& flags) {
  constexpr int MAXCON = L::Caps::MAXCON;
  const int g = lane >> 4, a = lane & 15, rb = lane & 48;
  const bool slot = g < nrows;
  const int src = cbase + (slot ? g : 0);
  const int code = wave_shfl_i(codeA, src), pair = wave_shfl_i(pkA, src);
  const unsigned pm1 = (unsigned)wave_shfl_i((int)m1A, src), pm2 = (unsigned)wave_shfl_i((int)m2A, src);
  const int pob = wave_shfl_i(obA, src), pdim = wave_shfl_i(dimA, src);
  const int g1 = code & 255, g2 = (code >> 8) & 255;
  GeomPose P1 = geom_pose(s, g1), P2 = geom_pose(s, g2);
  // box half sizes: from the pair record lane `src` fetched with the candidate (one L2 round trip for the whole candidate chunk
  // instead of a dependent g_size load per pass)
  float s1[3] = {wave_shfl(saA.x, src), wave_shfl(saA.y, src), wave_shfl(saA.z, src)}, s2[3] = {wave_shfl(sbA.x, src), wave_shfl(sbA.y, src), wave_shfl(sbA.z, src)};
  (void)m;
  v3 Aa[3] = {col(P1.R, 0), col(P1.R, 1), col(P1.R, 2)}, Ba[3] = {col(P2.R, 0), col(P2.R, 1), col(P2.R, 2)};
  v3 pp = P2.p - P1.p;
  const int ai = a < 15 ? a : 0;
  int ei = ai >= 6 ? (ai - 6) / 3 : (ai < 3 ? ai : 0), ej = ai >= 6 ? (ai - 6) % 3 : (ai >= 3 ? ai - 3 : 0);
  // (selects instead of runtime-indexed arrays: those would live in scratch memory)
  v3 Ai = ei == 0 ? Aa[0] : (ei == 1 ? Aa[1] : Aa[2]), Bj = ej == 0 ? Ba[0] : (ej == 1 ? Ba[1] : Ba[2]);
  v3 ax = ai < 3 ? Ai : (ai < 6 ? Bj : cross(Ai, Bj));
  float len = norm(ax);
  bool degenerate = ai >= 6 && len < 1e-6f;
  ax = ax * (1.f / fmaxf(len, 1e-30f));
  float ra = s1[0] * fabsf(dot(ax, Aa[0])) + s1[1] * fabsf(dot(ax, Aa[1])) + s1[2] * fabsf(dot(ax, Aa[2]));
  float rb_ = s2[0] * fabsf(dot(ax, Ba[0])) + s2[1] * fabsf(dot(ax, Ba[1])) + s2[2] * fabsf(dot(ax, Ba[2]));
  float dp = dot(pp, ax), pen = ra + rb_ - fabsf(dp);
  bool mine = slot && a < 15 && !degenerate;
  const unsigned long long sepm = wave_ballot(mine && pen < 0.f);
  const bool alive = slot && ((sepm >> rb) & 0xffffull) == 0ull;   // no separating axis in my row
  if (wave_ballot(alive) == 0ull) return;
  float fbest, ebest;
  int fcode = row_argmax(a < 6 ? -pen : -3.0e38f, a, &fbest);
  int ecode = row_argmax((a >= 6 && mine) ? -pen : -3.0e38f, a, &ebest);
  fbest = -fbest; ebest = -ebest;
  const bool edge = ebest < 1.0e38f && ebest * 1.05f < fbest;
  // up to two contacts per lane: (h1, d1, x1) from the first candidate round (or the row's edge contact, on its lane 0), (h2, d2, x2) from the second
  bool h1 = false, h2 = false;
  float d1 = 0.f, d2 = 0.f;
  v3 x1 = mk3(0, 0, 0), x2 = mk3(0, 0, 0), nrm = mk3(0, 0, 1.f);
  if (wave_ballot(alive && edge)) {
    const int ec = edge ? ecode : 6;
    const int i = (ec - 6) / 3, j = (ec - 6) % 3;
    float sgn = wave_shfl(dp, rb + ec) < 0.f ? -1.f : 1.f;
    v3 n = mk3(wave_shfl(ax.x, rb + ec), wave_shfl(ax.y, rb + ec), wave_shfl(ax.z, rb + ec)) * sgn;
    v3 ea = P1.p, eb = P2.p;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const v3 ea2 = ea + Aa[k] * (dot(n, Aa[k]) > 0.f ? s1[k] : -s1[k]), eb2 = eb + Ba[k] * (dot(n, Ba[k]) > 0.f ? -s2[k] : s2[k]);
      if (k != i) ea = ea2;
      if (k != j) eb = eb2;
    }
    v3 ua = i == 0 ? Aa[0] : (i == 1 ? Aa[1] : Aa[2]), ub = j == 0 ? Ba[0] : (j == 1 ? Ba[1] : Ba[2]);
    v3 r = eb - ea;
    float uv = dot(ua, ub), du = dot(r, ua), dv = dot(r, ub), den = 1.f - uv * uv;
    float sa = den > 1e-12f ? (du - uv * dv) / den : 0.f, tb = den > 1e-12f ? (uv * du - dv) / den : 0.f;
    if (alive && edge) {
      h1 = a == 0; d1 = -ebest; nrm = n;
      x1 = ((ea + ua * sa) + (eb + ub * tb)) * 0.5f;
    }
  }
  if (wave_ballot(alive && !edge)) {
    // face contact: reference box (axis ia), incident box
    bool refis1 = fcode < 3;
    int ia = refis1 ? fcode : fcode - 3;
    float bsign = wave_shfl(dp, rb + fcode) < 0.f ? -1.f : 1.f;
    v3 RA[3], IA[3];
    float rs[3], is[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { RA[k] = refis1 ? Aa[k] : Ba[k]; IA[k] = refis1 ? Ba[k] : Aa[k]; rs[k] = refis1 ? s1[k] : s2[k]; is[k] = refis1 ? s2[k] : s1[k]; }
    v3 rp = refis1 ? P1.p : P2.p, ip = refis1 ? P2.p : P1.p;
    // rotate so that the reference axis is index 0 (ia), then iu, iv
    v3 Rn = ia == 0 ? RA[0] : (ia == 1 ? RA[1] : RA[2]);
    v3 Ru = ia == 0 ? RA[1] : (ia == 1 ? RA[2] : RA[0]);
    v3 Rv = ia == 0 ? RA[2] : (ia == 1 ? RA[0] : RA[1]);
    float rn = ia == 0 ? rs[0] : (ia == 1 ? rs[1] : rs[2]);
    float a_ = ia == 0 ? rs[1] : (ia == 1 ? rs[2] : rs[0]);
    float b_ = ia == 0 ? rs[2] : (ia == 1 ? rs[0] : rs[1]);
    v3 nref = Rn * (refis1 ? bsign : -bsign);
    // incident face: most anti-parallel to nref (first index wins ties)
    float d0 = dot(IA[0], nref), dd1 = dot(IA[1], nref), dd2 = dot(IA[2], nref);
    int ib = 0;
    float mind = -fabsf(d0);
    if (-fabsf(dd1) < mind) { mind = -fabsf(dd1); ib = 1; }
    if (-fabsf(dd2) < mind) { mind = -fabsf(dd2); ib = 2; }
    float dd = ib == 0 ? d0 : (ib == 1 ? dd1 : dd2), isg = dd > 0.f ? -1.f : 1.f;
    v3 In = ib == 0 ? IA[0] : (ib == 1 ? IA[1] : IA[2]);
    v3 Iu = ib == 0 ? IA[1] : (ib == 1 ? IA[2] : IA[0]);
    v3 Iv = ib == 0 ? IA[2] : (ib == 1 ? IA[0] : IA[1]);
    float in_ = ib == 0 ? is[0] : (ib == 1 ? is[1] : is[2]);
    float iu_ = ib == 0 ? is[1] : (ib == 1 ? is[2] : is[0]);
    float iv_ = ib == 0 ? is[2] : (ib == 1 ? is[0] : is[1]);
    v3 fc = ip + In * (isg * in_);
    v3 rel = fc - rp;
    v3 c2 = mk3(dot(rel, Ru), dot(rel, Rv), dot(rel, nref) - rn);
    v3 eu = mk3(dot(Iu, Ru), dot(Iu, Rv), dot(Iu, nref)) * iu_;
    v3 ev = mk3(dot(Iv, Ru), dot(Iv, Rv), dot(Iv, nref)) * iv_;
    // corner k of the (+-1, +-1) square, counter-clockwise from (-1, -1)
    bool hv1 = false, hv2 = false;
    v3 pt1 = mk3(0, 0, 0), pt2 = mk3(0, 0, 0);
    {   // round 1: lanes 0..3 incident vertex k inside the reference face; lanes 4..7 reference corner k inside the incident face
      const int k = a & 3;
      const float kx = (k == 1 || k == 2) ? 1.f : -1.f, ky = k >= 2 ? 1.f : -1.f;
      const v3 q0 = c2 + eu * kx + ev * ky;
      if (a < 4) {
        hv1 = fabsf(q0.x) <= a_ && fabsf(q0.y) <= b_;
        pt1 = q0;
      } else if (a < 8) {
        float det = eu.x * ev.y - eu.y * ev.x;
        if (fabsf(det) > 1e-14f) {
          float x = kx * a_ - c2.x, y = ky * b_ - c2.y;
          float al = (x * ev.y - y * ev.x) / det, be = (eu.x * y - eu.y * x) / det;
          hv1 = fabsf(al) < 1.f && fabsf(be) < 1.f;
          pt1 = mk3(kx * a_, ky * b_, c2.z + al * eu.z + be * ev.z);
        }
      }
    }
    {   // round 2: incident edge k (corner k to corner k + 1) against reference side e
      const int k = a >> 2, k1 = (k + 1) & 3;
      const float kx = (k == 1 || k == 2) ? 1.f : -1.f, ky = k >= 2 ? 1.f : -1.f;
      const float k1x = (k1 == 1 || k1 == 2) ? 1.f : -1.f, k1y = k1 >= 2 ? 1.f : -1.f;
      const v3 q0 = c2 + eu * kx + ev * ky, q1 = c2 + eu * k1x + ev * k1y;
      const int e = a & 3, axn = e & 1;
      float lim = ((e & 2) ? 1.f : -1.f) * (axn ? b_ : a_), other = axn ? a_ : b_;
      float q0a = axn ? q0.y : q0.x, q1a = axn ? q1.y : q1.x, q0o = axn ? q0.x : q0.y, q1o = axn ? q1.x : q1.y;
      float e0 = q0a - lim, e1 = q1a - lim;
      if (!((e0 < 0.f) == (e1 < 0.f) || e0 == e1)) {
        float t = e0 / (e0 - e1);
        float o = q0o + t * (q1o - q0o);
        if (t > 0.f && t < 1.f && fabsf(o) < other) {
          hv2 = true;
          pt2 = axn ? mk3(o, lim, q0.z + t * (q1.z - q0.z)) : mk3(lim, o, q0.z + t * (q1.z - q0.z));
        }
      }
    }
    if (alive && !edge) {
      h1 = hv1 && !(pt1.z > 0.f); h2 = hv2 && !(pt2.z > 0.f);
      // dist = HALF the overlap at the contact's midpoint: what the reference's MuJoCo 2.0 reports for box-box face contacts, pinned by the
      // object-on-holder transient and rest height its recorded trajectories hold (tests/golden/mujoco_rest_heights.json)
      d1 = 0.5f * pt1.z; d2 = 0.5f * pt2.z;
      x1 = rp + Ru * pt1.x + Rv * pt1.y + nref * (rn + 0.5f * pt1.z);
      x2 = rp + Ru * pt2.x + Rv * pt2.y + nref * (rn + 0.5f * pt2.z);
      nrm = nref * (refis1 ? 1.f : -1.f);
    }
  }
  // append: row by row; inside a row the first round's contacts (lane order), then the second round's
  const unsigned long long M1 = wave_ballot(h1), M2 = wave_ballot(h2);
  int off = 0, tot = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int cr = popc64((M1 >> (16 * r)) & 0xffffull) + popc64((M2 >> (16 * r)) & 0xffffull);
    off += r < g ? cr : 0;
    tot += cr;
  }
  const unsigned long long r1 = (M1 >> rb) & 0xffffull, r2 = (M2 >> rb) & 0xffffull, below = (1ull << a) - 1ull;
  const int i1 = ncon + off + popc64(r1 & below), i2 = ncon + off + popc64(r1) + popc64(r2 & below);
  if (h1 && i1 < MAXCON) {
    s.c_dist[i1] = d1; st3(s.c_pos[i1], x1); make_frame(nrm, s.c_frame[i1]); s.c_pair[i1] = pair;
    s.c_m1[i1] = pm1; s.c_m2[i1] = pm2; s.c_ob[i1] = pob; s.c_dim[i1] = pdim;
  }
  if (h2 && i2 < MAXCON) {
    s.c_dist[i2] = d2; st3(s.c_pos[i2], x2); make_frame(nrm, s.c_frame[i2]); s.c_pair[i2] = pair;
    s.c_m1[i2] = pm1; s.c_m2[i2] = pm2; s.c_ob[i2] = pob; s.c_dim[i2] = pdim;
  }
  if (ncon + tot > MAXCON) { flags |= JFLAG_CON_OVERFLOW; tot = MAXCON - ncon; }
  ncon += tot;
}

// ---------------------------------------------------------------- MPR (all lanes run the same serial control flow)
struct Sup { v3 v, v1; };   // a point of the Minkowski difference G1 - G2 and its witness on G1 (the one on G2 is v1 - v: not kept -- the routine is the kernel's register hot spot)
// Everything a support query needs about one geom, fetched once per candidate pair.
struct MprGeom { GeomPose P; v3 size; int type, adr, nvert, cellR, celladr; };
#ifdef JACO_ROW_REUSE   // A/B build: a hull's support-table row is kept while consecutive queries fall into the same cube-map cell
struct HullRow { int cell; v4 e; };
#define JROWARG , HullRow& H1, HullRow& H2
#define JROWPASS , H1, H2
#else
#define JROWARG
#define JROWPASS
#endif
template <class L>
JDEV MprGeom mpr_geom(const JacoModelDev* m, const L& s, int g, int type) {
  MprGeom G;
  G.P = geom_pose(s, g);
  G.size = ld3(m->g_size[g]);
  G.type = type;
  G.adr = type == JG_MESH ? m->g_vertadr[g] : 0;
  G.nvert = type == JG_MESH ? m->g_vertnum[g] : 0;
  G.cellR = type == JG_MESH ? m->g_cellR[g] : 0;
  G.celladr = type == JG_MESH ? m->g_celladr[g] : 0;
  return G;
}
// Cube-map cell of a (wave-uniform) direction; same rule as modelc/supportmap.py:cube_cell.
JDEV int cube_cell(v3 d, int R) {
  float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  const bool fx = ax >= ay && ax >= az, fy = !fx && ay >= az;
  int face = fx ? (d.x > 0.f ? 0 : 1) : (fy ? (d.y > 0.f ? 2 : 3) : (d.z > 0.f ? 4 : 5));
  float mx = fx ? ax : (fy ? ay : az);
  mx = mx < 1e-30f ? 1.f : mx;
  float u = (fx ? d.y : d.x) / mx, v = ((fx || fy) ? d.z : d.y) / mx;
  int iu = (int)floorf((u + 1.f) * 0.5f * (float)R), iv = (int)floorf((v + 1.f) * 0.5f * (float)R);
  iu = iu < 0 ? 0 : (iu > R - 1 ? R - 1 : iu);
  iv = iv < 0 ? 0 : (iv > R - 1 ? R - 1 : iv);
  return (face * R + iu) * R + iv;
}
JDEV v3 support_prim(const MprGeom& G, v3 l) {   // box / cylinder / sphere, local frame
  if (G.type == JG_BOX) return mk3(l.x > 0.f ? G.size.x : -G.size.x, l.y > 0.f ? G.size.y : -G.size.y, l.z > 0.f ? G.size.z : -G.size.z);
  if (G.type == JG_CYLINDER) {   // radius size.x, half height size.y, axis z: rim point towards the direction's radial part
    const float rr = sqrtf(l.x * l.x + l.y * l.y), k = rr > JMINVAL ? G.size.x / rr : 0.f;
    return mk3(l.x * k, l.y * k, l.z > 0.f ? G.size.y : -G.size.y);
  }
  float n = norm(l);
  return l * (n > JMINVAL ? G.size.x / n : 0.f);
}
// Support point of the Minkowski difference G1 - G2 along `dir` (wave-uniform).  Hull meshes: the vertices of both hulls
// are scanned in one loop (64 lanes x float4 loads, both geoms' loads in flight together); every lane keeps the coordinates
// of its own best vertex, so after the DPP argmax the winner is broadcast from its lane instead of being re-fetched from
// memory.  Lowest vertex index wins ties (a serial first-max scan), as in support_geom.
JDEV Sup mpr_support(const JacoStepArgs& A, const MprGeom& G1, const MprGeom& G2, v3 dir, int lane JROWARG) {
#ifdef JACO_EMULATED
  if (lane == 0) emu_counter[7]++;
#endif
  const v3 l1 = mulT(G1.P.R, dir), l2 = mulT(G2.P.R, -dir);
  // hulls with a support table (modelc/supportmap.py): the cell of the direction lists every vertex that can be the
  // maximiser, one slot per lane -> one 16-byte load instead of a scan; the others are scanned (both hulls in one loop)
  const bool tab1 = G1.cellR > 0, tab2 = G2.cellR > 0;
  const int n1 = G1.nvert, n2 = G2.nvert, s1 = tab1 ? 0 : n1, s2 = tab2 ? 0 : n2, nmax = s1 > s2 ? s1 : s2;
  float best1 = -3.0e38f, best2 = -3.0e38f;
  int bi1 = 0x7fffffff, bi2 = 0x7fffffff;
  v3 c1 = mk3(0.f, 0.f, 0.f), c2 = mk3(0.f, 0.f, 0.f);
  if (tab1) {
#ifdef JACO_ROW_REUSE
    const int cell1 = wave_uniform_i(cube_cell(l1, G1.cellR));
    if (cell1 != H1.cell) { H1.e = ld4(A.hull + 4 * ((size_t)G1.celladr + (size_t)cell1 * 64 + lane)); H1.cell = cell1; }
    const v4 e = H1.e;
#else
    const v4 e = ld4(A.hull + 4 * ((size_t)G1.celladr + (size_t)cube_cell(l1, G1.cellR) * 64 + lane));
#endif
    const int id = __builtin_bit_cast(int, e.w);
    if (id >= 0) { best1 = e.x * l1.x + e.y * l1.y + e.z * l1.z; bi1 = id; c1 = mk3(e.x, e.y, e.z); }
  }
  if (tab2) {
#ifdef JACO_ROW_REUSE
    const int cell2 = wave_uniform_i(cube_cell(l2, G2.cellR));
    if (cell2 != H2.cell) { H2.e = ld4(A.hull + 4 * ((size_t)G2.celladr + (size_t)cell2 * 64 + lane)); H2.cell = cell2; }
    const v4 e = H2.e;
#else
    const v4 e = ld4(A.hull + 4 * ((size_t)G2.celladr + (size_t)cube_cell(l2, G2.cellR) * 64 + lane));
#endif
    const int id = __builtin_bit_cast(int, e.w);
    if (id >= 0) { best2 = e.x * l2.x + e.y * l2.y + e.z * l2.z; bi2 = id; c2 = mk3(e.x, e.y, e.z); }
  }
  for (int i = lane; i < nmax; i += 64) {
    if (i < s1) {
      const v4 v = ld4(A.hull + 4 * (size_t)(G1.adr + i));
      float t = v.x * l1.x + v.y * l1.y + v.z * l1.z;
      if (t > best1) { best1 = t; bi1 = i; c1 = mk3(v.x, v.y, v.z); }
    }
    if (i < s2) {
      const v4 v = ld4(A.hull + 4 * (size_t)(G2.adr + i));
      float t = v.x * l2.x + v.y * l2.y + v.z * l2.z;
      if (t > best2) { best2 = t; bi2 = i; c2 = mk3(v.x, v.y, v.z); }
    }
  }
  v3 sp1, sp2;
  if (n1 > 0) {
    float bv;
    int wl;
    if (tab1) { bv = wave_max(best1); wl = ffs64(wave_ballot(best1 == bv && bi1 != 0x7fffffff)); }   // slots are in index order: lowest lane = lowest index
    else { const int win = wave_argmax(best1, bi1, &bv); wl = ffs64(wave_ballot(bi1 == win)); }      // the lane that holds the winning vertex
    wl = wl < 0 ? 0 : wl;   // (no lane matches for a non-finite direction)
    sp1 = mk3(wave_bcast(c1.x, wl), wave_bcast(c1.y, wl), wave_bcast(c1.z, wl));
  } else sp1 = support_prim(G1, l1);
  if (n2 > 0) {
    float bv;
    int wl;
    if (tab2) { bv = wave_max(best2); wl = ffs64(wave_ballot(best2 == bv && bi2 != 0x7fffffff)); }
    else { const int win = wave_argmax(best2, bi2, &bv); wl = ffs64(wave_ballot(bi2 == win)); }
    wl = wl < 0 ? 0 : wl;
    sp2 = mk3(wave_bcast(c2.x, wl), wave_bcast(c2.y, wl), wave_bcast(c2.z, wl));
  } else sp2 = support_prim(G2, l2);
  Sup r;
  r.v1 = G1.P.p + mul(G1.P.R, sp1);
  r.v = r.v1 - (G2.P.p + mul(G2.P.R, sp2));
  return r;
}
// The portal (p0 interior point, p1..p3 triangle) is kept in four named variables: an indexed array would live in scratch.
JDEV v3 portal_dir(const Sup& p1, const Sup& p2, const Sup& p3) { return normalized(cross(p2.v - p1.v, p3.v - p1.v)); }
JDEV bool reach_tol(const Sup& p1, const Sup& p2, const Sup& p3, const Sup& v4, v3 dir, float tol) {
  float dv4 = dot(v4.v, dir);
  float mn = fminf(dv4 - dot(p1.v, dir), fminf(dv4 - dot(p2.v, dir), dv4 - dot(p3.v, dir)));
  return mn <= tol;
}
JDEV void expand_portal(const Sup& p0, Sup& p1, Sup& p2, Sup& p3, const Sup& v4) {
  v3 c = cross(v4.v, p0.v);
  if (dot(p1.v, c) > 0.f) {
    if (dot(p2.v, c) > 0.f) p1 = v4; else p3 = v4;
  } else {
    if (dot(p3.v, c) > 0.f) p2 = v4; else p1 = v4;
  }
}
JDEV float point_tri_closest(v3 a, v3 b, v3 c, v3* cp) {  // closest point of triangle abc to the origin
  v3 ab = b - a, ac = c - a, ap = -a;
  float d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0.f && d2 <= 0.f) { *cp = a; return norm(a); }
  v3 bp = -b;
  float d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0.f && d4 <= d3) { *cp = b; return norm(b); }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { *cp = a + ab * (d1 / (d1 - d3)); return norm(*cp); }
  v3 cpv = -c;
  float d5 = dot(ab, cpv), d6 = dot(ac, cpv);
  if (d6 >= 0.f && d5 <= d6) { *cp = c; return norm(c); }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { *cp = a + ac * (d2 / (d2 - d6)); return norm(*cp); }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { *cp = b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6))); return norm(*cp); }
  float den = 1.f / (va + vb + vc);
  *cp = a + ab * (vb * den) + ac * (vc * den);
  return norm(*cp);
}
JDEV v3 mpr_find_pos(const Sup& p0, const Sup& p1, const Sup& p2, const Sup& p3) {
  float b0 = dot(cross(p1.v, p2.v), p3.v), b1 = dot(cross(p3.v, p2.v), p0.v);
  float b2 = dot(cross(p0.v, p1.v), p3.v), b3 = dot(cross(p2.v, p1.v), p0.v);
  float sum = b0 + b1 + b2 + b3;
  if (sum <= 0.f) {
    v3 dir = portal_dir(p1, p2, p3);
    b0 = 0.f;
    b1 = dot(cross(p2.v, p3.v), dir); b2 = dot(cross(p3.v, p1.v), dir); b3 = dot(cross(p1.v, p2.v), dir);
    sum = b1 + b2 + b3;
  }
  // mid-point of the two witnesses: (a1 + a2) / 2 with a2 = a1 - sum_k b_k v_k
  v3 a1 = p0.v1 * b0 + p1.v1 * b1 + p2.v1 * b2 + p3.v1 * b3;
  v3 av = p0.v * b0 + p1.v * b1 + p2.v * b2 + p3.v * b3;
  return (a1 - av * 0.5f) * (1.f / sum);
}
// returns true on penetration
// On a miss that ended on a support test (the support of G1 - G2 along `dr` does not reach past the origin) *sep = dr: a separating
// direction, which the caller caches for the pair (stage_collision, "separating directions").
#ifdef JACO_MPR_FSM
// A/B build (tools/build_variant.sh mprfsm -DJACO_MPR_FSM): the same routine as a state machine around ONE support-query site (the straight-line
// version below inlines the query six times).  Phases: 0 / 1 the first two portal points, 2 portal discovery, 3 refinement until the portal
// faces the origin, 4 refinement to tolerance.  Same arithmetic in the same order per phase: results are bit-identical.
JDEV bool mpr_penetration(const JacoStepArgs& A, const JacoModelDev* m, const MprGeom& G1, const MprGeom& G2, int lane,
                          float* depth, v3* dirout, v3* pos, v3* sep, bool* sepvalid JROWARG) {
  Sup p0, p1, p2, p3;
  const float tol = m->mpr_tolerance;
  const int maxit = m->mpr_iterations;
  *sepvalid = false;
  p0.v1 = G1.P.p; p0.v = p0.v1 - G2.P.p;
  if (norm(p0.v) < 1e-9f) p0.v.x = 1e-5f;
  p1 = p0; p2 = p0; p3 = p0;
  v3 dr = normalized(-p0.v);
  int phase = 0, it = 0;
  for (;;) {
    phase = wave_uniform_i(phase);
    if (phase == 2 && it > 100) return false;
    const Sup q = mpr_support(A, G1, G2, dr, lane JROWPASS);
    const float qd = dot(q.v, dr);
    if (phase == 0) {
      p1 = q;
      if (qd <= 0.f) { *sep = dr; *sepvalid = true; return false; }
      dr = cross(p0.v, p1.v);
      if (norm(dr) < 1e-9f) {
        *depth = norm(p1.v); *dirout = normalized(p1.v); *pos = p1.v1 - p1.v * 0.5f;
        return true;
      }
      dr = normalized(dr);
      phase = 1;
    } else if (phase == 1) {
      p2 = q;
      if (qd <= 0.f) { *sep = dr; *sepvalid = true; return false; }
      dr = normalized(cross(p1.v - p0.v, p2.v - p0.v));
      if (dot(dr, p0.v) > 0.f) { Sup t = p1; p1 = p2; p2 = t; dr = -dr; }
      phase = 2; it = 0;
    } else if (phase == 2) {
      p3 = q;
      if (qd <= 0.f) { *sep = dr; *sepvalid = true; return false; }
      bool cont = false;
      if (dot(cross(p1.v, p3.v), p0.v) < -1e-11f) { p2 = p3; cont = true; }
      if (!cont && dot(cross(p3.v, p2.v), p0.v) < -1e-11f) { p1 = p3; cont = true; }
      if (cont) { dr = normalized(cross(p1.v - p0.v, p2.v - p0.v)); it++; }
      else {
        dr = portal_dir(p1, p2, p3);
        phase = dot(dr, p1.v) >= 0.f ? 4 : 3; it = 0;
      }
    } else if (phase == 3) {
      if (qd < 0.f) { *sep = dr; *sepvalid = true; return false; }
      if (reach_tol(p1, p2, p3, q, dr, tol) || it > maxit) return false;
      expand_portal(p0, p1, p2, p3, q);
      dr = portal_dir(p1, p2, p3);
      it++;
      if (dot(dr, p1.v) >= 0.f) { phase = 4; it = 0; }
    } else {
      if (reach_tol(p1, p2, p3, q, dr, tol) || it > maxit) {
        if (m->mpr_output == 1) { *dirout = dr; *depth = qd; }
        else {
          v3 cp;
          *depth = point_tri_closest(p1.v, p2.v, p3.v, &cp);
          *dirout = *depth < 1e-10f ? dr : normalized(cp);
        }
        *pos = mpr_find_pos(p0, p1, p2, p3);
        return true;
      }
      expand_portal(p0, p1, p2, p3, q);
      dr = portal_dir(p1, p2, p3);
      it++;
    }
  }
}
#else
JDEV bool mpr_penetration(const JacoStepArgs& A, const JacoModelDev* m, const MprGeom& G1, const MprGeom& G2, int lane,
                          float* depth, v3* dirout, v3* pos, v3* sep, bool* sepvalid JROWARG) {
  Sup p0, p1, p2, p3, v4;
  float tol = m->mpr_tolerance;
  *sepvalid = false;
  p0.v1 = G1.P.p; p0.v = p0.v1 - G2.P.p;
  if (norm(p0.v) < 1e-9f) p0.v.x = 1e-5f;
  v3 dr = normalized(-p0.v);
  p1 = mpr_support(A, G1, G2, dr, lane JROWPASS);
  if (dot(p1.v, dr) <= 0.f) { *sep = dr; *sepvalid = true; return false; }
  dr = cross(p0.v, p1.v);
  if (norm(dr) < 1e-9f) {
    *depth = norm(p1.v); *dirout = normalized(p1.v); *pos = p1.v1 - p1.v * 0.5f;
    return true;
  }
  dr = normalized(dr);
  p2 = mpr_support(A, G1, G2, dr, lane JROWPASS);
  if (dot(p2.v, dr) <= 0.f) { *sep = dr; *sepvalid = true; return false; }
  dr = normalized(cross(p1.v - p0.v, p2.v - p0.v));
  if (dot(dr, p0.v) > 0.f) { Sup t = p1; p1 = p2; p2 = t; dr = -dr; }
  for (int it = 0;; it++) {
    if (it > 100) return false;
    p3 = mpr_support(A, G1, G2, dr, lane JROWPASS);
    if (dot(p3.v, dr) <= 0.f) { *sep = dr; *sepvalid = true; return false; }
    bool cont = false;
    if (dot(cross(p1.v, p3.v), p0.v) < -1e-11f) { p2 = p3; cont = true; }
    if (!cont && dot(cross(p3.v, p2.v), p0.v) < -1e-11f) { p1 = p3; cont = true; }
    if (!cont) break;
    dr = normalized(cross(p1.v - p0.v, p2.v - p0.v));
  }
  for (int it = 0;; it++) {
    dr = portal_dir(p1, p2, p3);
    if (dot(dr, p1.v) >= 0.f) break;
    v4 = mpr_support(A, G1, G2, dr, lane JROWPASS);
    if (dot(v4.v, dr) < 0.f) { *sep = dr; *sepvalid = true; return false; }
    if (reach_tol(p1, p2, p3, v4, dr, tol) || it > m->mpr_iterations) return false;
    expand_portal(p0, p1, p2, p3, v4);
  }
  for (int it = 0;; it++) {
    dr = portal_dir(p1, p2, p3);
    v4 = mpr_support(A, G1, G2, dr, lane JROWPASS);
    if (reach_tol(p1, p2, p3, v4, dr, tol) || it > m->mpr_iterations) {
      if (m->mpr_output == 1) {
        // Portal-plane output: the refined portal lies (within mpr_tolerance) in the face of the Minkowski difference that the
        // ray from the interior point through the origin leaves by; its normal and the support value h(dr) = v4 . dr do not
        // depend on which triangle of that face the refinement ended on.  libccd's closest point of the final triangle
        // (mpr_output 0) is the same whenever the origin projects into the triangle; otherwise it depends on the refinement
        // path, and for shallow contacts its direction is the normalisation of a ~1e-6 vector: noise in fp32.
        *dirout = dr;
        *depth = dot(v4.v, dr);
      } else {
        v3 cp;
        *depth = point_tri_closest(p1.v, p2.v, p3.v, &cp);
        *dirout = *depth < 1e-10f ? dr : normalized(cp);
      }
      *pos = mpr_find_pos(p0, p1, p2, p3);
      return true;
    }
    expand_portal(p0, p1, p2, p3, v4);
  }
}
#endif

// ---------------------------------------------------------------- MPR, two pairs side by side -- built, measured, NOT shipped (-DJACO_MPR_PAIRS=1)
// Measured on MI355X (round 5, profiles/r05_ab_variants.txt item 8): bit-identical results, and no gain -- policy-driven 1.378 / 1.382 M without, 1.379 / 1.370 M with
// the option on the same build -- while the mere presence of the code (scratch 288 -> 348 B per lane in the light kernel: the narrowphase is the register
// hot spot) costs the headline 2.4 % (1.90 -> 1.855 M).  Kept as a build option with its emulator test (tests/test_kernel_emu.py, on-demand build).
#ifndef JACO_MPR_PAIRS
#define JACO_MPR_PAIRS 0
#endif
#if JACO_MPR_PAIRS
// Everything MPR computes between two support queries is the same handful of 3-vector operations in every lane; the 64 lanes only matter inside a
// query, where they hold the 64 candidate vertices of a hull's support-table cell.  Under the shipped policy a grasping env runs ~6 hull pairs
// through MPR per substep, ~10 queries and portal steps each -- half of its substep.  Here TWO candidates go through the routine at once, one per
// half wave: lanes 0..31 carry pair A's geoms, portal and direction, lanes 32..63 pair B's (no register more than before: the "uniform" values simply
// differ between the halves); a lane holds TWO table slots (l and l + 32 of its pair's cell), the maximum is taken inside the half (DPP row
// maximum + one exchange with the neighbouring row).  The control flow is mpr_penetration's state machine with the phase as a per-lane value:
// every pass of the loop makes one query for both halves, then each half takes its own step.  Per pair the arithmetic, its order and the
// tie-breaking (lowest vertex index) are those of the one-pair routine: results are bit-identical (tests: test_mpr_pairs_*).
// Eligible: candidates whose hull meshes all have a support table (every finger / hand / link hull; the 1 511-vertex base mesh is scanned).
// support point of one geom (per-half values G, l) in its local frame; hulls: table cell of the half's direction, two slots per lane
JDEV v3 support_half(const JacoStepArgs& A, const MprGeom& G, v3 l, int lane, bool anyhull) {
  v3 sp = support_prim(G, l);
  if (anyhull) {   // (wave-uniform: some half's geom is a hull; the other half loads a cell of the table's first row for nothing)
    const bool hull = G.cellR > 0;
    const int gl = lane & 31, gb = lane & 32;
    const float* row = A.hull + 4 * ((size_t)(hull ? G.celladr : 0) + (size_t)(hull ? cube_cell(l, G.cellR) : 0) * 64);
    const v4 ea = ld4(row + 4 * gl), eb = ld4(row + 4 * (gl + 32));
    const int ida = __builtin_bit_cast(int, ea.w), idb = __builtin_bit_cast(int, eb.w);
    const float ta = (hull && ida >= 0) ? ea.x * l.x + ea.y * l.y + ea.z * l.z : -3.0e38f;
    const float tb = (hull && idb >= 0) ? eb.x * l.x + eb.y * l.y + eb.z * l.z : -3.0e38f;
    const bool useb = tb > ta;   // (slots are in vertex-index order: on a tie the lower slot stays)
    const float best = useb ? tb : ta;
    const bool have = hull && (useb ? idb : ida) >= 0;
    const v3 c = useb ? mk3(eb.x, eb.y, eb.z) : mk3(ea.x, ea.y, ea.z);
    const float bv = half_max(best);
    const bool cand = have && best == bv;
    // lowest slot among the maxima: the lower 32 slots (held as `a`) before the upper 32
    const unsigned ma = (unsigned)(wave_ballot(cand && !useb) >> gb), mb = (unsigned)(wave_ballot(cand && useb) >> gb);
    const unsigned mm = ma ? ma : mb;
    const int wl = gb + (mm ? ffs64((unsigned long long)mm) : 0);   // (no lane matches for a non-finite direction: the half's first lane)
    const v3 w = mk3(wave_shfl(c.x, wl), wave_shfl(c.y, wl), wave_shfl(c.z, wl));
    if (hull) sp = w;
  }
  return sp;
}
JDEV Sup mpr_support_half(const JacoStepArgs& A, const MprGeom& G1, const MprGeom& G2, v3 dir, int lane, bool anyhull1, bool anyhull2) {
  const v3 l1 = mulT(G1.P.R, dir), l2 = mulT(G2.P.R, -dir);
  const v3 sp1 = support_half(A, G1, l1, lane, anyhull1), sp2 = support_half(A, G2, l2, lane, anyhull2);
  Sup r;
  r.v1 = G1.P.p + mul(G1.P.R, sp1);
  r.v = r.v1 - (G2.P.p + mul(G2.P.R, sp2));
  return r;
}
// Per half: the cached-direction test (sepd.w != 0: one query along the cached direction; apart -> nothing to do) and, if that does not settle it,
// mpr_penetration's state machine.  `active`: the half has a pair at all.  Outputs are per-half values: hit / depth / dir / pos as mpr_penetration's,
// ran = MPR itself was run (the caller then stores sep / sepvalid in the pair's cache entry, as the one-pair path does).
JDEV void mpr_pair2(const JacoStepArgs& A, const JacoModelDev* m, const MprGeom& G1, const MprGeom& G2, v4 sepd, bool active, int lane,
                    bool* hit_out, float* depth, v3* dirout, v3* pos, v3* sep, bool* sepvalid, bool* ran_out) {
  const bool anyhull1 = wave_ballot(G1.cellR > 0) != 0ull, anyhull2 = wave_ballot(G2.cellR > 0) != 0ull;
  const float tol = m->mpr_tolerance;
  const int maxit = m->mpr_iterations;
  Sup p0, p1, p2, p3;
  p0.v1 = G1.P.p; p0.v = p0.v1 - G2.P.p;
  if (norm(p0.v) < 1e-9f) p0.v.x = 1e-5f;
  p1 = p0; p2 = p0; p3 = p0;
  enum { DONE = 9 };
  int phase = !active ? DONE : (sepd.w != 0.f ? -1 : 0), it = 0;
  v3 dr = phase == -1 ? mk3(sepd.x, sepd.y, sepd.z) : normalized(-p0.v);
  bool hit = false, ran = active && phase == 0, sv = false;
  v3 sp = mk3(0.f, 0.f, 0.f);
  *depth = 0.f; *dirout = mk3(0.f, 0.f, 1.f); *pos = mk3(0.f, 0.f, 0.f);
#ifdef JACO_EMULATED
  int nq = 0;   // (CPU diagnostics: support queries of this half's pair, tools/mpr_query_stats.py)
#endif
  while (wave_ballot(phase != DONE) != 0ull) {
    if (phase == 2 && it > 100) phase = DONE;
#ifdef JACO_EMULATED
    if (phase != DONE) nq++;
#endif
    const Sup q = mpr_support_half(A, G1, G2, dr, lane, anyhull1, anyhull2);   // (every lane, every pass: the one place with cross-lane traffic)
    const float qd = dot(q.v, dr);
    if (phase == -1) {
      const bool apart = qd < -1e-5f && fabsf(dot(dr, dr) - 1.f) < 1e-3f;
      if (apart) phase = DONE;
      else { phase = 0; ran = true; dr = normalized(-p0.v); }
    } else if (phase == 0) {
      p1 = q;
      if (qd <= 0.f) { sp = dr; sv = true; phase = DONE; }
      else {
        dr = cross(p0.v, p1.v);
        if (norm(dr) < 1e-9f) {
          *depth = norm(p1.v); *dirout = normalized(p1.v); *pos = p1.v1 - p1.v * 0.5f;
          hit = true; phase = DONE;
        } else { dr = normalized(dr); phase = 1; }
      }
    } else if (phase == 1) {
      p2 = q;
      if (qd <= 0.f) { sp = dr; sv = true; phase = DONE; }
      else {
        dr = normalized(cross(p1.v - p0.v, p2.v - p0.v));
        if (dot(dr, p0.v) > 0.f) { Sup t = p1; p1 = p2; p2 = t; dr = -dr; }
        phase = 2; it = 0;
      }
    } else if (phase == 2) {
      p3 = q;
      if (qd <= 0.f) { sp = dr; sv = true; phase = DONE; }
      else {
        bool cont = false;
        if (dot(cross(p1.v, p3.v), p0.v) < -1e-11f) { p2 = p3; cont = true; }
        if (!cont && dot(cross(p3.v, p2.v), p0.v) < -1e-11f) { p1 = p3; cont = true; }
        if (cont) { dr = normalized(cross(p1.v - p0.v, p2.v - p0.v)); it++; }
        else {
          dr = portal_dir(p1, p2, p3);
          phase = dot(dr, p1.v) >= 0.f ? 4 : 3; it = 0;
        }
      }
    } else if (phase == 3) {
      if (qd < 0.f) { sp = dr; sv = true; phase = DONE; }
      else if (reach_tol(p1, p2, p3, q, dr, tol) || it > maxit) phase = DONE;
      else {
        expand_portal(p0, p1, p2, p3, q);
        dr = portal_dir(p1, p2, p3);
        it++;
        if (dot(dr, p1.v) >= 0.f) { phase = 4; it = 0; }
      }
    } else if (phase == 4) {
      if (reach_tol(p1, p2, p3, q, dr, tol) || it > maxit) {
        if (m->mpr_output == 1) { *dirout = dr; *depth = qd; }
        else {
          v3 cp;
          *depth = point_tri_closest(p1.v, p2.v, p3.v, &cp);
          *dirout = *depth < 1e-10f ? dr : normalized(cp);
        }
        *pos = mpr_find_pos(p0, p1, p2, p3);
        hit = true; phase = DONE;
      } else {
        expand_portal(p0, p1, p2, p3, q);
        dr = portal_dir(p1, p2, p3);
        it++;
      }
    }
  }
  *hit_out = hit; *sep = sp; *sepvalid = sv; *ran_out = ran;
#ifdef JACO_EMULATED
  if ((lane & 31) == 0 && active) { emu_counter[7] += nq; emu_counter[hit ? 5 : 6] += nq; }
#endif
}

#endif   // JACO_MPR_PAIRS

// ---------------------------------------------------------------- stage C
// Pair list of the bounding-sphere phase (temporal coherence across the substeps of a launch).  Geoms move well under a
// millimetre per 1 ms substep, yet the phase used to test all ~720 whitelisted pairs every substep.  A full pass now also lists,
// in pair order, the pairs whose sphere gap is below JACO_PAIRLIST_SLACK (1.5 cm; lane l keeps entries l, l + 64, ... in registers) and
// remembers where every geom was; a later substep runs the SAME exact test on the listed pairs only, as long as no geom centre has
// moved further than 0.45 x slack since (an unlisted pair then still has a gap of a tenth of the slack: it would fail the exact test
// by millimetres).  The survivors -- and through them every contact -- are identical to those of the all-pairs pass; the marker
// geoms that _take_action moves and a reset trigger the rebuild through the same displacement test.
#ifndef JACO_PAIRLIST_SLACK
#define JACO_PAIRLIST_SLACK 0.015f
#endif
template <class C>
struct PairList {
  static constexpr int NV = C::MAXCAND > 256 ? 8 : 4;   // registers per lane, 64 entries each
  unsigned e[NV];        // pair index | g1 << JPL_KBITS | g2 << (JPL_KBITS + JPL_GBITS) | (g1 is a plane) << (JPL_KBITS + 2 JPL_GBITS)
  float px[JGEOM_PASSES], py[JGEOM_PASSES], pz[JGEOM_PASSES];   // geom 64 p + lane: its position when the list was built
  int n;                 // entries (wave-uniform); -1: no list (first substep of a launch, or more near pairs than the registers hold)
  unsigned planes;       // bit j: entries 64 j .. 64 j + 63 hold a plane pair
};
#define JPL_KBITS (JMAXPAIR <= 1024 ? 10 : 12)
#define JPL_GBITS (JMAXGEOM <= 64 ? 6 : 7)
static_assert(JMAXPAIR <= (1 << JPL_KBITS) && JMAXGEOM <= (1 << JPL_GBITS) && JPL_KBITS + 2 * JPL_GBITS + 1 <= 32, "pair-list entry packing");
// the exact bounding test of one pair: fixed fma chains, so that the all-pairs pass and the list pass round identically
JDEV bool sphere_test(const v4& pa, const v4& pb, const v3& nrm, bool plane, float slack) {
  const float dx = pb.x - pa.x, dy = pb.y - pa.y, dz = pb.z - pa.z;
  const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx)), dn = fmaf(dz, nrm.z, fmaf(dy, nrm.y, dx * nrm.x));
  const float r = (pa.w + pb.w) + slack, rp = pb.w + slack;
  return !((plane ? dn : dd) > (plane ? rp : r * r));
}
// all-pairs pass over the chunks 4 G .. 4 G + 3 (lane = pair): survivors are appended to s.cand in pair order, near pairs to `tmp`
template <bool PLANES, class L>
JDEV void sphere_group(L& s, const JacoModelDev* m, int G, int npair, int lane, int& n1, int& nl, unsigned* tmp, int tmpcap) {
  v4 pa[4], pb[4];
  v3 nrm[4];
  int codes[4];
#pragma unroll
  for (int c = 0; c < 4; c++) codes[c] = m->pair_code[(4 * G + c) * 64 + lane];   // (zero-padded to JMAXPAIR: straight-line loads)
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int g1 = codes[c] & 255, g2 = (codes[c] >> 8) & 255;
    pa[c] = ld4(s.gpos[g1]); pb[c] = ld4(s.gpos[g2]);
    nrm[c] = mk3(0.f, 0.f, 0.f);
    if (PLANES) {   // branch-free: every lane also evaluates the plane form (normal = third column of g1's frame)
      nrm[c] = mk3(s.gmat[g1][2], s.gmat[g1][5], s.gmat[g1][8]);
      keep_loaded(nrm[c].x, nrm[c].y, nrm[c].z);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int k = (4 * G + c) * 64 + lane;
    const bool valid = k < npair;
    const int g1 = codes[c] & 255, g2 = (codes[c] >> 8) & 255;
    const bool plane = PLANES && ((codes[c] >> 16) & 15) == JG_PLANE;
    const bool pass = sphere_test(pa[c], pb[c], nrm[c], plane, 0.f) & valid;
    const bool near = sphere_test(pa[c], pb[c], nrm[c], plane, JACO_PAIRLIST_SLACK) & valid;
    const unsigned long long mask = wave_ballot(pass), nmask = wave_ballot(near);
    const int idx = n1 + wave_prefix_count(mask), nidx = nl + wave_prefix_count(nmask);
    if (pass && idx < L::Caps::MAXCAND) s.cand[idx] = k;
    if (near && nidx < tmpcap) tmp[nidx] = (unsigned)k | ((unsigned)g1 << JPL_KBITS) | ((unsigned)g2 << (JPL_KBITS + JPL_GBITS)) | (plane ? 1u << (JPL_KBITS + 2 * JPL_GBITS) : 0u);
    n1 += popc64(mask);
    nl += popc64(nmask);
  }
}
// list pass over entries 64 j .. 64 j + 63
template <bool PLANES, class L>
JDEV void list_chunk(L& s, unsigned ent, bool valid, int& n1) {
  const int k = (int)(ent & ((1u << JPL_KBITS) - 1u)), g1 = (int)((ent >> JPL_KBITS) & ((1u << JPL_GBITS) - 1u)), g2 = (int)((ent >> (JPL_KBITS + JPL_GBITS)) & ((1u << JPL_GBITS) - 1u));
  const v4 pa = ld4(s.gpos[g1]), pb = ld4(s.gpos[g2]);
  v3 nrm = mk3(0.f, 0.f, 0.f);
  if (PLANES) nrm = mk3(s.gmat[g1][2], s.gmat[g1][5], s.gmat[g1][8]);
  const bool pass = sphere_test(pa, pb, nrm, PLANES && ((ent >> (JPL_KBITS + 2 * JPL_GBITS)) & 1u) != 0u, 0.f) & valid;
  const unsigned long long mask = wave_ballot(pass);
  const int idx = n1 + wave_prefix_count(mask);
  if (pass && idx < L::Caps::MAXCAND) s.cand[idx] = k;
  n1 += popc64(mask);
}

template <class L>
JDEV void stage_collision(const JacoStepArgs& A, const JacoModelDev* m, L& s, int env, int lane, unsigned& flags, JProfCtx& pc, PairList<typename L::Caps>& pl) {
  (void)pc;
  typedef PairList<typename L::Caps> PL;
  // phase 1: bounding spheres.  How far has any geom centre moved since the list was built?
  int n1 = 0;
  const int npair = m->npair, plane_chunks = m->plane_chunks;
  static_assert(JMAXPAIR % 256 == 0, "groups of four 64-pair chunks");
  v3 gp[JGEOM_PASSES];
  bool rebuild = pl.n < 0 || A.no_pairlist;
  float d2 = 0.f;
#pragma unroll
  for (int p = 0; p < JGEOM_PASSES; p++) {
    const bool isg = 64 * p + lane < m->ngeom;
    gp[p] = isg ? ld3(s.gpos[isg ? 64 * p + lane : 0]) : mk3(0.f, 0.f, 0.f);
    if (!rebuild) {
      const float dx = gp[p].x - pl.px[p], dy = gp[p].y - pl.py[p], dz = gp[p].z - pl.pz[p];
      const float dd = isg ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : 0.f;
      d2 = dd > d2 || !(dd == dd) ? dd : d2;   // (keeps a NaN)
    }
  }
  if (!rebuild) {
    const float lim = 0.45f * JACO_PAIRLIST_SLACK;
    rebuild = wave_ballot(!(d2 < lim * lim)) != 0ull;   // (a non-finite position rebuilds, too)
  }
#ifdef JACO_EMULATED
  if (lane == 0) { emu_counter[rebuild ? 0 : 1]++; if (!rebuild) emu_counter[2] += pl.n; }   // (CPU tests: how often each pass ran, entries tested by the list passes)
#endif
  if (rebuild) {
    // all pairs, lane = pair.  Four chunks per straight-line group: all their LDS gathers are in flight before the first ballot.
    // Plane pairs sit in the leading chunks only (pairs are ordered by geom id and planes are static world geoms), so the groups
    // behind them skip the plane form.  The near pairs are collected in the early stages' scratch area (dead by now) ...
    unsigned* tmp = reinterpret_cast<unsigned*>(s.early_scratch);
    static_assert(PL::NV * 64 <= JSCRATCH, "pair-list staging area");
    int nl = 0;
#if JMAXPAIR == 768
    if (0 < plane_chunks) sphere_group<true>(s, m, 0, npair, lane, n1, nl, tmp, PL::NV * 64); else sphere_group<false>(s, m, 0, npair, lane, n1, nl, tmp, PL::NV * 64);
    if (256 < npair) { if (4 < plane_chunks) sphere_group<true>(s, m, 1, npair, lane, n1, nl, tmp, PL::NV * 64); else sphere_group<false>(s, m, 1, npair, lane, n1, nl, tmp, PL::NV * 64); }
    if (512 < npair) { if (8 < plane_chunks) sphere_group<true>(s, m, 2, npair, lane, n1, nl, tmp, PL::NV * 64); else sphere_group<false>(s, m, 2, npair, lane, n1, nl, tmp, PL::NV * 64); }
#else   // (bigger pair tables: a loop over the groups)
    for (int G = 0; 256 * G < npair; G++) {
      if (4 * G < plane_chunks) sphere_group<true>(s, m, G, npair, lane, n1, nl, tmp, PL::NV * 64); else sphere_group<false>(s, m, G, npair, lane, n1, nl, tmp, PL::NV * 64);
    }
#endif
    wave_sync();
    // ... and dealt out to the lanes' registers
    pl.n = nl <= PL::NV * 64 ? nl : -1;
    pl.planes = 0u;
#pragma unroll
    for (int j = 0; j < PL::NV; j++) {
      const bool have = j * 64 + lane < nl && pl.n >= 0;
      pl.e[j] = have ? tmp[have ? j * 64 + lane : 0] : 0u;
      pl.planes |= wave_ballot(have && ((pl.e[j] >> (JPL_KBITS + 2 * JPL_GBITS)) & 1u) != 0u) ? 1u << j : 0u;
    }
#pragma unroll
    for (int p = 0; p < JGEOM_PASSES; p++) { pl.px[p] = gp[p].x; pl.py[p] = gp[p].y; pl.pz[p] = gp[p].z; }
  } else {
#pragma unroll
    for (int j = 0; j < PL::NV; j++) {
      if (j * 64 < pl.n) {
        const bool valid = j * 64 + lane < pl.n;
        if ((pl.planes >> j) & 1u) list_chunk<true>(s, pl.e[j], valid, n1); else list_chunk<false>(s, pl.e[j], valid, n1);
      }
    }
  }
  JSTAMP(9);
  if (lane == 0) s.nsphere = n1;
  if (n1 > L::Caps::MAXCAND) { flags |= JFLAG_CAND_OVERFLOW; n1 = L::Caps::MAXCAND; }
  wave_sync();
  // phase 2: oriented-box cull of the survivors, lane = survivor; compacted in place (order preserved)
  int ncand = 0;
  for (int base = 0; base < n1; base += 64) {
    int ci = base + lane;
    int k = ci < n1 ? s.cand[ci] : 0;
    bool keep = ci < n1;
    if (keep) {
      const v4 q0 = ld4(reinterpret_cast<const float*>(&m->pair_obb[k])), q1 = ld4(reinterpret_cast<const float*>(&m->pair_obb[k]) + 4);
      const int code = __builtin_bit_cast(int, q0.x);
      int g1 = code & 255, g2 = (code >> 8) & 255, t1 = (code >> 16) & 15;
      if (t1 != JG_PLANE) keep = !obb_separated(geom_pose(s, g1), mk3(q0.y, q0.z, q0.w), geom_pose(s, g2), mk3(q1.x, q1.y, q1.z));
    }
    wave_sync();
    unsigned long long mask = wave_ballot(keep);
    if (keep) s.cand[ncand + wave_prefix_count(mask)] = k;
    ncand += popc64(mask);
    wave_sync();
  }
  JSTAMP(10);
  // phase 3: narrowphase, whole wave per candidate
  int ncon = 0;
  // 64 candidates at a time: their pair index, code and contact bookkeeping are fetched by lane c in one go and broadcast inside
  // the loop (instead of an LDS read + two dependent scalar loads in front of every narrowphase call)
  for (int cb = 0; cb < ncand; cb += 64) {
    const int nhere = ncand - cb < 64 ? ncand - cb : 64;
    const int pkA = s.cand[lane < nhere ? cb + lane : cb];
    // the pair's 32-byte record: code + the two geoms' box half sizes (what box-box and plane-box need next)
    const v4 rec0 = ld4(reinterpret_cast<const float*>(&m->pair_obb[pkA])), rec1 = ld4(reinterpret_cast<const float*>(&m->pair_obb[pkA]) + 4);
    const int codeA = __builtin_bit_cast(int, rec0.x);
    const v3 saA = mk3(rec0.y, rec0.z, rec0.w), sbA = mk3(rec1.x, rec1.y, rec1.z);
    const unsigned m1A = m->pair[pkA].m1, m2A = m->pair[pkA].m2;
    const int obA = m->pair[pkA].ob, dimA = m->pair[pkA].condim;
    const unsigned long long boxes = wave_ballot(lane < nhere && ((codeA >> 16) & 255) == (JG_BOX | (JG_BOX << 4)));
    // candidates that go through MPR (neither plane-* nor box-box): two in a row can share a wave (mpr_pair2)
    const unsigned long long hullc = wave_ballot(lane < nhere && ((codeA >> 16) & 15) != JG_PLANE && ((codeA >> 16) & 255) != (JG_BOX | (JG_BOX << 4)));
    // Separating directions (hull pairs).  A pair that MPR found apart ended on a direction d with  max over (G1 - G2) of x . d <= 0; as
    // long as one support query along the cached d still gives < -10 um the geoms are provably apart and MPR -- which would say the same
    // after ~5 queries -- is not run.  Per (env, pair) one float4 in global memory (w = 1: valid), fetched here with the pair records; any
    // stale or foreign value is harmless (it is re-validated by that one query), so the cache needs no hand-off discipline and cannot
    // change a result: only misses are skipped.
    v4 sepA; sepA.x = sepA.y = sepA.z = sepA.w = 0.f;
    float* const seprow = A.sepdir ? A.sepdir + (size_t)env * (4 * JMAXPAIR) : nullptr;
    if (seprow && lane < nhere && ((codeA >> 16) & 15) != JG_PLANE && ((codeA >> 16) & 255) != (JG_BOX | (JG_BOX << 4))) sepA = ld4(seprow + 4 * pkA);
    JSTAMP_NARROW(11);
#if JACO_MPR_PAIRS
    // Hull candidates, two at a time (option "mpr_pairs"; mpr_pair2): the finger hulls' pairs with the object sit between box-box pairs of the finger
    // pads in candidate order, so they are run up front, one per half wave, and their results parked -- 8 floats per candidate in the early stages'
    // scratch area, dead since the pair list was dealt out -- for the in-order loop below, which appends the contacts exactly where the one-pair path
    // would have.  An odd one out (and every candidate when fewer than two are eligible) takes the one-pair path there.
    float* const stash = s.early_scratch;   // [64][8]: hit, depth, pos, dir
    static_assert(JSCRATCH >= 512, "MPR result stash");
    unsigned long long parked = 0ull;
    if (A.mpr_pairs && popc64(hullc) >= 2) {
      // (eligible: every hull mesh of the pair has its support table -- all but the 1 511-vertex base mesh, which is scanned)
      const int gA1 = codeA & 255, gA2 = (codeA >> 8) & 255;
      const bool okA = lane < nhere && !((((codeA >> 16) & 15) == JG_MESH && m->g_cellR[gA1] == 0) || (((codeA >> 20) & 15) == JG_MESH && m->g_cellR[gA2] == 0));
      unsigned long long todo = hullc & wave_ballot(okA);
      while (popc64(todo) >= 2) {
        lane = wave_opaque_i(lane);
        const int c1 = ffs64(todo); todo &= todo - 1ull;
        const int c2 = ffs64(todo); todo &= todo - 1ull;
        const int cc = lane < 32 ? c1 : c2;
        const int pk2 = wave_shfl_i(pkA, cc), code2 = wave_shfl_i(codeA, cc);
        const MprGeom H1 = mpr_geom(m, s, code2 & 255, (code2 >> 16) & 15), H2 = mpr_geom(m, s, (code2 >> 8) & 255, (code2 >> 20) & 15);
        v4 sd; sd.x = wave_shfl(sepA.x, cc); sd.y = wave_shfl(sepA.y, cc); sd.z = wave_shfl(sepA.z, cc); sd.w = wave_shfl(sepA.w, cc);
        bool hit2, sv2, ran2;
        float depth2;
        v3 dir2, pos2, sep2;
        mpr_pair2(A, m, H1, H2, sd, true, lane, &hit2, &depth2, &dir2, &pos2, &sep2, &sv2, &ran2);
        if ((lane & 31) == 0) {
          if (seprow && ran2 && (sv2 || sd.w != 0.f)) {
            v4 o; o.x = sep2.x; o.y = sep2.y; o.z = sep2.z; o.w = sv2 ? 1.f : 0.f;
            *reinterpret_cast<v4*>(seprow + 4 * pk2) = o;
          }
          float* st = stash + 8 * cc;
          st[0] = hit2 ? 1.f : 0.f; st[1] = depth2; st[2] = pos2.x; st[3] = pos2.y; st[4] = pos2.z; st[5] = dir2.x; st[6] = dir2.y; st[7] = dir2.z;
#ifdef JACO_EMULATED
          emu_counter[3]++; emu_counter[4] += hit2;
#endif
        }
#ifdef JACO_EMULATED
        if (lane == 0) emu_counter[10]++;
#endif
        parked |= (1ull << c1) | (1ull << c2);
      }
      wave_sync();   // the parked results are visible
      JSTAMP_NARROW(14);
    }
#else
    const unsigned long long parked = 0ull;
    (void)hullc;
#endif
    for (int c = 0; c < nhere;) {
      lane = wave_opaque_i(lane);   // (lane-id predicates of the narrowphase routines stay inside the loop: physics_kernel.h stage_newton)
      // contact buffer full and a bigger tier to hand the env to: the rest of the narrowphase would be thrown away with the substep
      if (L::Caps::MAXEFC < JacoHuge::MAXEFC && (wave_uniform_i((int)flags) & (int)JFLAG_CON_OVERFLOW)) break;
      if ((boxes >> c) & 1ull) {   // a run of up to four box-box pairs, one 16-lane row each
        const int run = ffs64(~(boxes >> c) | 16ull);
        collide_box_box4(m, s, c, run, pkA, codeA, m1A, m2A, obA, dimA, saA, sbA, lane, ncon, flags);
        c += run;
        JSTAMP_NARROW(12);
        continue;
      }
      const int pk = wave_bcast_i(pkA, c), code = wave_bcast_i(codeA, c);
      const unsigned pm1 = (unsigned)wave_bcast_i((int)m1A, c), pm2 = (unsigned)wave_bcast_i((int)m2A, c);
      const int pob = wave_bcast_i(obA, c), pdim = wave_bcast_i(dimA, c);
      const int g1 = code & 255, g2 = (code >> 8) & 255, t1 = (code >> 16) & 15, t2 = (code >> 20) & 15;
      const int before = ncon;
      if (t1 == JG_PLANE) {
        if (t2 == JG_BOX) collide_plane_box(mk3(wave_bcast(sbA.x, c), wave_bcast(sbA.y, c), wave_bcast(sbA.z, c)), s, g1, g2, pk, lane, ncon, flags);
        else if (t2 == JG_SPHERE) collide_plane_sphere(m, s, g1, g2, pk, lane, ncon, flags);
        else if (t2 == JG_MESH) collide_plane_convex(A, m, s, g1, g2, t2, pk, lane, ncon, flags);
        JSTAMP_NARROW(13);
      } else {
        float depth;
        v3 dir, pos;
#ifdef JACO_EMULATED
        const long q0_ = emu_counter[7];
#endif
        const MprGeom G1 = mpr_geom(m, s, g1, t1), G2 = mpr_geom(m, s, g2, t2);
#ifdef JACO_ROW_REUSE
        HullRow H1, H2;
        H1.cell = -1; H2.cell = -1; H1.e.x = H1.e.y = H1.e.z = H1.e.w = 0.f; H2.e = H1.e;
#endif
        const float sw = wave_bcast(sepA.w, c);
        bool hit = false, apart = false, sepvalid = false;
        v3 sep = mk3(0.f, 0.f, 0.f);
#if JACO_MPR_PAIRS
        if ((parked >> c) & 1ull) {   // run up front by the two-pair routine: the result is parked
          const float* st = stash + 8 * c;
          hit = st[0] != 0.f; depth = st[1]; pos = mk3(st[2], st[3], st[4]); dir = mk3(st[5], st[6], st[7]);
          apart = true;
        } else
#endif
        if (sw != 0.f) {   // (wave-uniform)
          const v3 d = mk3(wave_bcast(sepA.x, c), wave_bcast(sepA.y, c), wave_bcast(sepA.z, c));
          const Sup q = mpr_support(A, G1, G2, d, lane JROWPASS);
          apart = dot(q.v, d) < -1e-5f && fabsf(dot(d, d) - 1.f) < 1e-3f;
        }
        if (!apart) {
          hit = mpr_penetration(A, m, G1, G2, lane, &depth, &dir, &pos, &sep, &sepvalid JROWPASS);
          if (seprow && lane == 0 && (sepvalid || sw != 0.f)) {
            v4 o; o.x = sep.x; o.y = sep.y; o.z = sep.z; o.w = sepvalid ? 1.f : 0.f;
            *reinterpret_cast<v4*>(seprow + 4 * pk) = o;
          }
        }
#ifdef JACO_EMULATED
        if (lane == 0 && !((parked >> c) & 1ull)) { emu_counter[3]++; emu_counter[4] += hit; emu_counter[hit ? 5 : 6] += emu_counter[7] - q0_; }   // (CPU diagnostics: tools/mpr_query_stats.py)
#endif
#ifdef JACO_TRACE_MPR
        if (lane == 0) printf("mpr g1 %d g2 %d hit %d depth %g\n", g1, g2, (int)hit, hit ? depth : 0.f);
#endif
        push_contacts(s, hit && lane == 0, -depth, pos, dir, pk, ncon, flags, 1);
        JSTAMP_NARROW(hit ? 14 : 1);
      }
      if (ncon > before) {   // dof chain masks / body ids of the two geoms, for the row builder and the touch stage
        int cc = before + lane;
        if (cc < ncon) { s.c_m1[cc] = pm1; s.c_m2[cc] = pm2; s.c_ob[cc] = pob; s.c_dim[cc] = pdim; }
      }
      c++;
    }
    if (L::Caps::MAXEFC < JacoHuge::MAXEFC && (wave_uniform_i((int)flags) & (int)JFLAG_CON_OVERFLOW)) break;
  }
  if (lane == 0) { s.ncon = ncon; s.ncand = ncand; }
}

// ---------------------------------------------------------------- stage R (contacts): pyramidal rows
// Side rows (light tier, models with a pedestal block).  The 64-row light tier overflows by a handful of rows exactly in the regimes
// real rollouts live in: object on holder (16 rows) + pedestal on floor (16) + the EE's axis sticks lying on the "hand" marker's
// sticks (9 contacts = 36) = 68.  The pedestal's rows touch nothing but the pedestal's own six dofs (dof block [JB1, JNV)), and as
// long as no other row reaches into that block its sub-problem is exactly separable from the rest of the Newton problem.  So when
// the rows do not fit, the contacts whose dof support is the pedestal block alone go to a side buffer of up to JSIDE_ROWS rows x 6
// columns, solved by a small Newton solve of its own (newton_side, physics_kernel.h), and the main buffer keeps the other <= 64.
// The side buffer costs no LDS: it overlays cdof[6..] + cvel, which are dead between this stage and the next substep's tree walk
// (the controller reads cdof[0..5] only).  Virtual row index of side row j: JSIDE_BASE + j.
#define JSIDE_ROWS 16
#define JSIDE_BASE 64
#define JSIDE_J 0        // [JSIDE_ROWS][6] Jacobian columns JB1 .. JNV-1
#define JSIDE_AREF 96    // [JSIDE_ROWS] aref, later the residual staging of the side solve
#define JSIDE_D 112      // [JSIDE_ROWS]
#define JSIDE_F 128      // [JSIDE_ROWS] row weights during the side solve, row forces after it
template <class L>
struct SideRows {
  static constexpr bool on = L::Caps::CONTACT && L::Caps::MAXEFC == 64 && JNV - JB1 == 6 && (JNV - 6) * 6 + JNB * 6 >= JSIDE_F + JSIDE_ROWS;
};
template <class L>
JDEV float* side_buf(L& s) { return &s.cdof[6][0]; }
// row velocity J_row . qvel of pyramid edge e of contact c, from the two bodies' spatial velocities: the contact-frame components of the
// relative point velocity (k < 3) or relative angular velocity (k >= 3), combined as the pyramid edge  v_0 +- mu_k v_k
template <class L>
JDEV float contact_row_vel(const L& s, int c, int e, float mu) {
  const int pd = s.c_dim[c];
  const int kf = pd == 1 ? 0 : 1 + (e >> 1);
  const int obs = s.c_ob[c], b1 = ((obs >> 8) & 0xFF) - 1, b2 = ((obs >> 24) & 0xFF) - 1;
  sv v1, v2;
  v1.a = v1.b = v2.a = v2.b = mk3(0.f, 0.f, 0.f);
  if (b1 >= 0) v1 = ldsv(s.cvel[b1]);
  if (b2 >= 0) v2 = ldsv(s.cvel[b2]);
  v3 cp = ld3(s.c_pos[c]);
  v3 dw = v2.a - v1.a, dv = (v2.b - v1.b) + cross(dw, cp);
  const float* fr = s.c_frame[c];
  float vel = dot(ld3(fr), dv);
  if (pd > 1) {
    const int ax = kf < 3 ? kf : kf - 3;
    const float comp = dot(ld3(fr + 3 * ax), kf < 3 ? dv : dw);
    vel += ((e & 1) ? -1.f : 1.f) * mu * comp;
  }
  return vel;
}
template <class L>
JDEV void stage_contact_rows(const JacoModelDev* m, L& s, int lane, unsigned& flags) {
  constexpr int MAXEFC = L::Caps::MAXEFC;
  constexpr bool SIDE = SideRows<L>::on;
  const int nv = m->nv, ncon = wave_uniform_i(s.ncon), nlim = wave_uniform_i(s.nlimit);
  // per-contact parameters handed to the row lanes through LDS: a contact's chain masks are dead once its own lane has read
  // them (each lane only ever overwrites its own contact's slots), c_fn is not yet in use
  float* pa0 = reinterpret_cast<float*>(s.c_m1);
  float* pbc = reinterpret_cast<float*>(s.c_m2);
  float* sd = side_buf(s);
  int rowbase = nlim, kept_total = 0, nside = 0, side_cand = 0;
  // Jacobian rows: lane = (contact slot 0..2, dof); the slot's contact data is fetched from its owner lane
  constexpr int CPP = JNV <= 21 ? 3 : 64 / JNV;   // contacts per Jacobian pass (lane = contact slot x dof)
  const int cl = lane / JNV, k = lane - cl * JNV;
  const bool dofok = cl < CPP && k < nv;
  const sv S = ldsv(s.cdof[dofok ? k : 0]);
  for (int cb = 0; cb < ncon; cb += 64) {   // lane = contact, 64 at a time (one pass unless the tier holds more than 64 contacts)
    const int ci = cb + lane, nhere = ncon - cb < 64 ? ncon - cb : 64;
    // rows per contact and row offsets
    int dim = 0, nrow = 0;
    float mu0 = 0.f, mu1 = 0.f, mu2 = 0.f;
    unsigned cm1 = 0, cm2 = 0;
    // per-contact row parameters, from the pair's pre-mixed constants: every row of a contact shares the regulariser
    // (pyramidal: 2 mu^2 R of the first row, impratio = 1) and the position part of aref; only the velocity part differs per
    // row.  aref_row = a0 - bcoef * vel_row,  D_row = dinv.
    float a0 = 0.f, bcoef = 0.f, dinv = 0.f;
    if (lane < nhere) {
      const JacoPairParam& P = m->pair[s.c_pair[ci]];
      dim = s.c_dim[ci]; nrow = dim == 1 ? 1 : 2 * (dim - 1);
      mu0 = P.mu[0]; mu1 = P.mu[2]; mu2 = P.mu[3];   // (mu[1] == mu[0], mu[4] == mu[3]: one tangential, one rolling coefficient)
      cm1 = s.c_m1[ci]; cm2 = s.c_m2[ci];
      const float pos = s.c_dist[ci] - P.margin, tran = P.tran;
      float R, imp;
      a0 = row_params(P.solref, P.solimp, pos, 0.f, dim == 1 ? tran : tran + mu0 * mu0 * tran, &R, &imp);
      bcoef = 2.f / fmaxf(JMINVAL, fminf(0.9999f, fmaxf(0.0001f, P.solimp[1])) * P.solref[0]);
      if (dim > 1) R = fmaxf(JMINVAL, 2.f * mu0 * mu0 * R);
      dinv = 1.f / R;
    }
    // which contacts could live in the side buffer: dof support = the pedestal block alone, at most 4 rows; usable when they number at most
    // JSIDE_ROWS rows and no other contact reaches into that block (every tier counts them: "would the light tier cope?" needs the number)
    const unsigned mm = cm1 | cm2;
    bool sidec = false;
    // inclusive prefix sum of the row counts.  A contact has 1, 4 or 10 rows (condim 1 / 3 / 6): three ballots and their prefix counts
    // instead of a six-step shuffle scan (each step an LDS-crossbar round trip)
    const unsigned long long upto = lane >= 63 ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= mine
    const unsigned long long r1m = wave_ballot(nrow == 1), r4m = wave_ballot(nrow == 4), r10m = wave_ballot(nrow == 10);
    const bool rows_std = wave_ballot(nrow != 0 && nrow != 1 && nrow != 4 && nrow != 10) == 0ull;
    int end = rowbase + (rows_std ? popc64(r1m & upto) + 4 * popc64(r4m & upto) + 10 * popc64(r10m & upto) : wave_scan_incl(nrow, lane));
    // (only looked for when the rows exceed the light tier's buffer: nobody asks for the number otherwise)
#ifdef JACO_SIDE_ALWAYS   // A/B build (tools/build_variant.sh sidealways -DJACO_SIDE_ALWAYS): the pedestal's rows go to the side solve whenever they are separable
    if (JNV - JB1 == 6 && cb == 0) {
#else
    if (JNV - JB1 == 6 && cb == 0 && wave_bcast_i(end, nhere - 1) > JSIDE_BASE) {
#endif
      const bool only2 = lane < nhere && mm != 0u && (mm & ((1u << JB1) - 1u)) == 0u && nrow <= 4;
      const bool reach2 = lane < nhere && !only2 && (mm >> JB1) != 0u;
      const unsigned long long o2 = wave_ballot(only2);
      if (o2 != 0ull && wave_ballot(reach2) == 0ull && ncon <= 64) {
        int srows = 0;   // (a handful of contacts: serial over the set bits, wave-uniform)
        for (unsigned long long b = o2; b; b &= b - 1ull) srows += wave_bcast_i(nrow, ffs64(b));
        if (srows <= JSIDE_ROWS) { side_cand = srows; sidec = only2; }
      }
    }
    bool split = false;
    int r0s = 0;
#ifdef JACO_SIDE_ALWAYS
    if (SIDE && side_cand > 0 && wave_bcast_i(end, nhere - 1) - side_cand <= MAXEFC) {
#else
    if (SIDE && side_cand > 0 && wave_bcast_i(end, nhere - 1) > MAXEFC && wave_bcast_i(end, nhere - 1) - side_cand <= MAXEFC) {
#endif
      // the rows do not fit the main buffer, and they do without the pedestal's: split
      split = true;
      end = rowbase + wave_scan_incl(sidec ? 0 : nrow, lane);
      r0s = JSIDE_BASE + wave_scan_incl(sidec ? nrow : 0, lane) - nrow;
      nside = side_cand;
    }
    sidec = sidec && split;
    const unsigned long long fm = wave_ballot(lane < nhere && end <= MAXEFC);
    const int kept = popc64(fm);   // row offsets are monotone, so the contacts that fit form a prefix
    int total = wave_bcast_i(end, kept > 0 ? kept - 1 : 0);
    total = kept > 0 ? total : rowbase;
    const int r0 = sidec ? r0s : end - nrow;
    if (lane < kept) {
      s.c_efc[ci] = r0;
      int blk = ((mm & ((1u << JB0) - 1u)) ? 1 : 0) | (((mm >> JB0) & ((1u << (JB1 - JB0)) - 1u)) ? 2 : 0) | ((mm >> JB1) ? 4 : 0);
      // body-space rows carry their body pair: (fused body + 1) of geom 1 / geom 2, 0 = static (c_ob keeps them at bits 8 / 24)
      const int bodies = L::Caps::WRENCH ? ((((s.c_ob[ci] >> 8) & 31) << 19) | (((s.c_ob[ci] >> 24) & 31) << 24)) : 0;
      if (!sidec) for (int e = 0; e < nrow; e++) {
        s.e_con[r0 + e] = ci | (e << 8) | (blk << 16) | bodies;
        s.e_f[r0 + e] = e < 4 ? mu0 : (e < 6 ? mu1 : mu2);   // friction of the row's pyramid edge (e_f is free until the solver runs)
      }
      pa0[ci] = a0; pbc[ci] = bcoef; s.c_fn[ci] = dinv;
    }
    wave_sync();   // e_con, the per-contact parameters and the row maps are visible
    // per-row parameters of this chunk's main rows, lane = row (before the Jacobian rows: in split mode those overwrite cvel)
    for (int rr = rowbase + lane; rr < total; rr += 64) {
      const int ce = s.e_con[rr], c = ce & 255, e = (ce >> 8) & 255;
      const float vel = contact_row_vel(s, c, e, s.e_f[rr]);
      s.e_aref[rr] = pa0[c] - pbc[c] * vel;
      s.e_D[rr] = s.c_fn[c];
    }
    // ... and of the side rows, by their contact's own lane, into registers (the side buffer overlays cvel, which is still being read)
    float aside[4] = {0.f, 0.f, 0.f, 0.f};
    if (SIDE && sidec && lane < kept) {
#pragma unroll
      for (int e = 0; e < 4; e++) if (e < nrow) aside[e] = a0 - bcoef * contact_row_vel(s, ci, e, mu0);
    }
    if (SIDE && split) {
      wave_sync();   // every read of cvel / cdof[6..] of this stage is done: the side buffer may be written
      if (sidec && lane < kept) {
#pragma unroll
        for (int e = 0; e < 4; e++) if (e < nrow) { sd[JSIDE_AREF + r0 - JSIDE_BASE + e] = aside[e]; sd[JSIDE_D + r0 - JSIDE_BASE + e] = dinv; }
      }
    }
    if (L::Caps::WRENCH) {
      // the rows themselves, lane = row: wrench of pyramid edge e of contact c about the world origin -- linear part dir = n +- mu t_k
      // (k < 3) or n (torsional / rolling edges), angular part pos x dir (+- mu axis_k for k >= 3) -- so that J[r][d] = sgn_d(r) w . S_d
      for (int rr = rowbase + lane; rr < total; rr += 64) {
        const int ce = s.e_con[rr], c = ce & 255, e = (ce >> 8) & 15, pd = s.c_dim[c];
        const int kf = pd == 1 ? 0 : 1 + (e >> 1);
        const float smu = ((e & 1) ? -1.f : 1.f) * s.e_f[rr];
        const float* fr = s.c_frame[c];
        const v3 n = ld3(fr), pos = ld3(s.c_pos[c]);
        v3 lin = n, ang = mk3(0.f, 0.f, 0.f);
        if (kf == 1 || kf == 2) lin = n + ld3(fr + 3 * kf) * smu;
        if (kf >= 3) ang = ld3(fr + 3 * (kf - 3)) * smu;
        ang = ang + cross(pos, lin);
        float* R = s.J + 8 * rr;
        R[0] = ang.x; R[1] = ang.y; R[2] = ang.z; R[3] = lin.x; R[4] = lin.y; R[5] = lin.z; R[6] = 0.f; R[7] = 0.f;
      }
    } else
    for (int c0 = 0; c0 < kept; c0 += CPP) {
      int c = c0 + (cl < CPP ? cl : 0);
      int src = c < 64 ? c : 0;
      int cdim = wave_shfl_i(dim, src), cr0 = wave_shfl_i(r0, src);
      float f0 = wave_shfl(mu0, src), f1 = wave_shfl(mu1, src), f2 = wave_shfl(mu2, src);
      unsigned a1 = (unsigned)wave_shfl_i((int)cm1, src), a2 = (unsigned)wave_shfl_i((int)cm2, src);
      const bool cside = SIDE && cr0 >= JSIDE_BASE;   // (side rows hold the pedestal block's six columns only: their other entries are zero by construction)
      if (dofok && c < kept && (!cside || k >= JB1)) {
        float coef = (float)((int)((a2 >> k) & 1u) - (int)((a1 >> k) & 1u));
        v3 pos = ld3(s.c_pos[cb + c]);
        v3 jp = (S.b + cross(S.a, pos)) * coef, jr = S.a * coef;
        const float* fr = s.c_frame[cb + c];
        float Jc[6];
#pragma unroll
        for (int a = 0; a < 3; a++) { Jc[a] = dot(ld3(fr + 3 * a), jp); Jc[3 + a] = dot(ld3(fr + 3 * a), jr); }
        const int ld = cside ? 6 : JLD;
        float* Jw = cside ? sd + JSIDE_J + (cr0 - JSIDE_BASE) * 6 + (k - JB1) : s.J + cr0 * JLD + k;
        if (cdim == 1) Jw[0] = Jc[0];
        else {
          Jw[0] = Jc[0] + f0 * Jc[1]; Jw[ld] = Jc[0] - f0 * Jc[1];
          Jw[2 * ld] = Jc[0] + f0 * Jc[2]; Jw[3 * ld] = Jc[0] - f0 * Jc[2];
          if (cdim > 3) {
            Jw[4 * ld] = Jc[0] + f1 * Jc[3]; Jw[5 * ld] = Jc[0] - f1 * Jc[3];
            Jw[6 * ld] = Jc[0] + f2 * Jc[4]; Jw[7 * ld] = Jc[0] - f2 * Jc[4];
            Jw[8 * ld] = Jc[0] + f2 * Jc[5]; Jw[9 * ld] = Jc[0] - f2 * Jc[5];
          }
        }
      }
    }
    rowbase = total; kept_total += kept;
    if (kept < nhere) { flags |= JFLAG_EFC_OVERFLOW; break; }   // the row buffer is full: this contact and all later ones are dropped
  }
  const int total = rowbase;
  wave_sync();  // every lane has read the incoming row / contact counts; J rows and row parameters are visible
  if (lane == 0) { s.nefc = total; s.ncon = kept_total; s.nside = nside; s.nside_cand = side_cand; }
}

// ---------------------------------------------------------------- stage T: touch sensors (env_mujoco_util.py:470-475 reads them)
// A contact counts when one of its geoms is on the site's own body and the ray from the contact point along the
// normal, pointing away from that body, meets the site volume (always true for a point inside it).
JDEV bool ray_hits_site(int type, float sx, float sy, float sz, v3 p, v3 d) {
  // 10 um of slack: pad contacts routinely sit exactly on their site's lateral boundary (same footprint), where
  // the inclusive test would be decided by rounding noise; the slack makes that case deterministic.
  sx += 1e-5f; sy += 1e-5f; sz += 1e-5f;
  float t0 = 0.f, t1 = 3.0e38f;
  bool ok = true;
  if (type == JG_BOX) {
    float pp[3] = {p.x, p.y, p.z}, dd[3] = {d.x, d.y, d.z}, ss[3] = {sx, sy, sz};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (fabsf(dd[i]) < JMINVAL) { ok = ok && !(fabsf(pp[i]) > ss[i]); continue; }
      float a = (-ss[i] - pp[i]) / dd[i], b = (ss[i] - pp[i]) / dd[i];
      t0 = fmaxf(t0, fminf(a, b));
      t1 = fminf(t1, fmaxf(a, b));
    }
    return ok && t1 >= t0;
  }
  if (type == JG_CYLINDER) {
    if (fabsf(d.z) < JMINVAL) ok = !(fabsf(p.z) > sy);
    else {
      float a = (-sy - p.z) / d.z, b = (sy - p.z) / d.z;
      t0 = fmaxf(t0, fminf(a, b));
      t1 = fminf(t1, fmaxf(a, b));
    }
    float A = d.x * d.x + d.y * d.y, B = p.x * d.x + p.y * d.y, C = p.x * p.x + p.y * p.y - sx * sx;
    if (A < JMINVAL) ok = ok && !(C > 0.f);
    else {
      float disc = B * B - A * C;
      if (disc < 0.f) return false;
      float sq = sqrtf(disc);
      t0 = fmaxf(t0, (-B - sq) / A);
      t1 = fminf(t1, (-B + sq) / A);
    }
    return ok && t1 >= t0;
  }
  float A = dot(d, d), B = dot(p, d), C = dot(p, p) - sx * sx;
  if (A < JMINVAL) return C <= 0.f;
  float disc = B * B - A * C;
  if (disc < 0.f) return false;
  return (-B + sqrtf(disc)) / A >= 0.f;
}
template <class L>
JDEV void stage_touch(const JacoModelDev* m, L& s, int lane, float* sens) {
  int ncon = wave_uniform_i(s.ncon);
  {   // no contact on a body that carries a touch site (the common case: only object/pedestal/floor contacts): all zero
    const unsigned sb0 = m->sens_bodymask[0], sb1 = m->sens_bodymask[1], sb2 = m->sens_bodymask[2], sb3 = m->sens_bodymask[3];
    bool mine = false;
    for (int ci = lane; ci < ncon; ci += 64) {
      const int obs = s.c_ob[ci];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int ob = (obs >> (16 * h)) & 0xFF;
        const unsigned w = (ob >> 5) == 0 ? sb0 : ((ob >> 5) == 1 ? sb1 : ((ob >> 5) == 2 ? sb2 : sb3));
        mine = mine || ((w >> (ob & 31)) & 1u) != 0u;
      }
    }
    if (!wave_ballot(mine)) { *sens = 0.f; return; }
  }
  for (int ci = lane; ci < ncon; ci += 64) {
    int cd = s.c_dim[ci];
    int nrow = cd == 1 ? 1 : 2 * (cd - 1), r0 = s.c_efc[ci];
    float fn = 0.f;
    if (SideRows<L>::on && r0 >= JSIDE_BASE) { const float* sd = side_buf(s); for (int e = 0; e < nrow; e++) fn += sd[JSIDE_F + r0 - JSIDE_BASE + e]; }
    else for (int e = 0; e < nrow; e++) fn += s.e_f[r0 + e];
    s.c_fn[ci] = fn;
  }
  wave_sync();
  float sum = 0.f;
  if (lane < m->nsensor) {
    int b = m->s_body[lane];
    m3 R = ldm(m->s_mat[lane]);
    v3 p = ld3(m->s_pos[lane]);
    if (b >= 0) { m3 Rb = ldm(s.xmat[b]); p = ld3(s.xpos[b]) + mul(Rb, p); R = mul(Rb, R); }
    int ob = m->s_origbody[lane], type = m->s_type[lane];
    float sx = m->s_size[lane][0], sy = m->s_size[lane][1], sz = m->s_size[lane][2];
    for (int c = 0; c < ncon; c++) {
      int obs = s.c_ob[c];
      bool on1 = (obs & 0xFF) == ob, on2 = ((obs >> 16) & 0xFF) == ob;
      if (!on1 && !on2) continue;
      float fn = s.c_fn[c];
      if (!(fn > JMINVAL)) continue;
      v3 l = mulT(R, ld3(s.c_pos[c]) - p);
      v3 ray = mulT(R, ld3(s.c_frame[c]) * (on2 ? -1.f : 1.f));
      if (ray_hits_site(type, sx, sy, sz, l, ray)) sum += fn;
    }
  }
  *sens = sum;
}
