// Device-side model constants (fp32), built on the host from the fused view of a JACOMDL1 blob
// (mujoco_jaco_amd/modelc/compile.py) and uploaded once per handle.  Plain-old-data: shared by
// the host loader (model_blob.cpp) and the kernels.  Index conventions follow the reference's
// joint discovery (env_script/mujoco.py:60-88): dofs 0..5 arm, 6..8 fingers, then free bodies.
#pragma once

// Layout of the default build (jaco2_curtain_torque.xml and everything that fits inside it, e.g. the arm-only jaco2_reaching_torque.xml).
// A second build of the same sources with -DJNB=13 -DJNV=18 -DJNQ=19 -DJB0=12 -DJB1=18 -DJNSENS=64 (libjaco_env_d12.so, __graft_entry__.build)
// serves the 12-hinge arm -- 6 arm + 6 finger hinges (proximal + sprung distal, jaco2_torque.xml:109-133) in one kinematic tree -- alone
// (jaco2_torque.xml) or with one free object (jaco2_curtain_torque_sensor.xml: static cylinder geoms, 61 touch sensors;
// jaco2_curtain_torque_old.xml).
#ifndef JNB
#define JNB 11        // moving (fused) bodies: 6 links, 3 fingers, object, destination pedestal
#define JNV 21        // dofs
#define JNQ 23        // generalized positions
#endif
#ifndef JB0
#define JB0 9         // dof blocks of the mass matrix (one per kinematic tree): [0,JB0) arm + fingers, [JB0,JB1) object, [JB1,JNV) pedestal
#define JB1 15
#endif
// A third build, -DJNB=20 -DJNV=30 -DJNQ=32 -DJB0=18 -DJB1=24 -DJNU=18 -DJNSENS=40 -DJMAXGEOM=128 -DJMAXPAIR=3584 -DJMAXINNER=12 -DJMAXMPAIR=192
// (libjaco_env_d30.so), serves jaco2_dual_torque.xml at the sim-interface (ctrl) level: two 9-dof arm trees side by side in dof block
// [0, 18) (block diagonal inside it: the factorisations treat it as one block), then the two free objects; 106 collidable geoms and
// 3 332 whitelisted pairs, so the frame, geom and pair stages run in two or more 64-lane passes there.
#ifndef JNU
#define JNU 9         // actuators: 6 motors + 3 finger position servos (xml:341-349)
#endif
#ifndef JNSENS
#define JNSENS 20     // touch sensors (xml:352-374)
#endif
#ifndef JMAXGEOM
#define JMAXGEOM 64   // collidable geoms
#endif
#ifndef JMAXPAIR
#define JMAXPAIR 768  // geom pairs passing the static collision filter
#endif
#ifndef JMAXINNER
#define JMAXINNER 6   // bodies that have children (link1..link6)
#endif
#ifndef JMAXCHAIN
#define JMAXCHAIN 7   // dofs that move one body: the six arm joints + its own finger joint (a free body's 6)
#endif
#ifndef JMAXDESC
#define JMAXDESC 9    // bodies of one subtree: the arm's links and fingers
#endif
#ifndef JMAXMPAIR
#define JMAXMPAIR 128 // structurally non-zero lower-triangle mass-matrix entries (84 for this model)
#endif
#define JNMOCAP 16

enum { JG_PLANE = 0, JG_SPHERE = 2, JG_CYLINDER = 5, JG_BOX = 6, JG_MESH = 7 };
enum { JJ_FREE = 0, JJ_HINGE = 3 };

struct JacoPairParam {
  float mu[5];       // tangent1, tangent2, torsional, rolling1, rolling2 (element-wise max of the two geoms)
  float solref[2];   // mean of the two geoms, timeconst clamped to 2*dt (refsafe)
  float solimp[5];   // mean of the two geoms
  float margin;
  int condim;        // max of the two geoms
  int g1, g2;        // fused geom ids, type(g1) <= type(g2)
  float tran, rot;   // translational / rotational body_invweight0 of the two geoms' bodies, summed (diagApprox of contact rows)
  // what a contact of this pair carries into the row builder, precomputed by the loader (one struct read instead of
  // g_body -> b_chainmask / g_origbody chains per contact):
  unsigned m1, m2;   // dof chain masks of the two geoms' bodies (0: static)
  int ob;            // original (unfused) body id | (fused body + 1) << 8 for g1; the same for g2 in the upper half
};

// what the oriented-box cull needs about a pair, in one 32-byte record (two 16-byte loads per surviving pair instead of a
// pair_code load followed by dependent g_size loads)
struct alignas(16) JacoPairObb {
  int code;          // = pair_code
  float sa[3];       // box half sizes of g1 (sphere: r, r, r)
  float sb[3];       // ... of g2
  int pad;
};

struct JacoModelDev {
  int nbody, nv, nq, nu, ngeom, npair, nsensor, nhullvert, nmocap;
  int plane_chunks;   // 64-pair chunks [0, plane_chunks) hold every pair whose first geom is a plane (pairs are in geom order)
  float timestep, gravity[3], tolerance, meaninertia, mpr_tolerance;
  int iterations, ls_iterations, mpr_iterations;
  int mpr_output;   // 1 (default): portal-plane normal + support depth; 0: libccd's closest point of the final portal triangle
  float ls_tolerance;
  float timestep_lo;   // timestep - (double)(float)timestep: the integrators carry the state compensated (hi + lo floats) and advance it by the fp64 step
  int compensated;     // 1 (default): qpos / qvel carried as hi + lo floats (physics_kernel.h, stage E); 0: plain fp32 state (comparison)

  // bodies (parents precede children)
  int b_parent[JNB], b_jtype[JNB], b_qadr[JNB], b_dadr[JNB], b_limited[JNB];
  float b_pos[JNB][3], b_mat[JNB][9], b_axis[JNB][3], b_qpos0[JNB];
  float b_qpos0_lo[JNB];   // joint reference angle: fp64 value - b_qpos0 (ref 3.14 is 1.05e-7 away from its float)
  float b_mass[JNB], b_com[JNB][3], b_inertia[JNB][6];  // xx yy zz xy xz yz about the CoM, body frame
  float b_range[JNB][2], b_solref[JNB][2], b_solimp[JNB][5];  // joint-limit solver parameters
  unsigned b_chainmask[JNB];                       // bit d set: dof d moves body b
  unsigned b_descmask[JNB];                        // bit x set: body x is in the subtree rooted at b (b included)
  int ninner, inner_body[JMAXINNER];               // bodies with children: the only ones whose composite sums differ from their own
  int nmpair, mpair[JMAXMPAIR];                    // (dof d | ancestor-or-self dof j << 8): one lane per mass-matrix entry

  // dofs
  int d_body[JNV], d_parent[JNV];
  float d_damping[JNV], d_invweight[JNV];
  float d_stiffness[JNV], d_springref[JNV];   // joint springs (hinge dofs): qfrc_passive -= stiffness (qpos - springref); jaco2_torque.xml:109-133
  int d_qadr[JNV];                            // qpos address of a hinge dof (-1: free-joint dof)
  int has_springs;
  int has_damping;   // 0 none, 1 finger joints only (dofs 6..8: the shared-elimination Euler solve applies), 2 anywhere in block 0

  // actuators: force = position ? kp*(clamp(ctrl) - qpos) : ctrl, then clamped (xml:341-349)
  int a_dof[JNU], a_qadr[JNU], a_position[JNU], a_ctrllimited[JNU], a_forcelimited[JNU];
  float a_kp[JNU], a_ctrlrange[JNU][2], a_forcerange[JNU][2];

  // collidable geoms; body -1 = static (g_pos/g_mat are then world poses)
  int g_body[JMAXGEOM], g_type[JMAXGEOM], g_vertadr[JMAXGEOM], g_vertnum[JMAXGEOM], g_origbody[JMAXGEOM], g_mocap[JMAXGEOM];
  int g_cellR[JMAXGEOM], g_celladr[JMAXGEOM];   // hull support table: cube-map resolution (0 = none) and first entry (float4 units)
  float g_pos[JMAXGEOM][3], g_mat[JMAXGEOM][9], g_size[JMAXGEOM][3], g_rbound[JMAXGEOM], g_invweight[JMAXGEOM][2];

  // geoms riding on the two markers the task layer moves every env step ("hand" = 0, "subgoal_reach" = 1): g_marker is the
  // marker index or -1; for those geoms g_lpos / g_lmat hold the pose in the marker's frame (g_pos / g_mat: XML rest pose)
  int g_marker[JMAXGEOM];
  float g_lpos[JMAXGEOM][3], g_lmat[JMAXGEOM][9];
  float marker_rest[2][12];   // rest pose of the two markers (position, rotation row-major): where sim.reset() puts them

  JacoPairParam pair[JMAXPAIR];
  int pair_code[JMAXPAIR];   // g1 | g2 << 8 | type(g1) << 16 | type(g2) << 20, for the lane-per-pair broadphase
  JacoPairObb pair_obb[JMAXPAIR];

  // touch sites, one per sensor, in sensordata order
  int s_body[JNSENS], s_type[JNSENS], s_origbody[JNSENS];
  float s_pos[JNSENS][3], s_mat[JNSENS][9], s_size[JNSENS][3];
  unsigned sens_bodymask[4];   // bit b (of 128): original body b carries a touch site (contacts elsewhere cannot reach a sensor)

  // named frames the task layer reads: body id + local pos + local rotation
  int ee_body, eeobj_body;
  float ee_pos[3], ee_mat[9], eeobj_pos[3], eeobj_mat[9], base_pos[3];
  int obj_body, dest_body;
};
