// Device-side constant tables shared by the kernels (raptor_hip.hip) and the host set-up code.
// Plain-old-data, laid out once per model / problem in HBM and read through uniform (scalar) loads.
#ifndef RAPTOR_DEVICE_TYPES_H
#define RAPTOR_DEVICE_TYPES_H

#define RAPTOR_MAX_NV 36          // Digit-v3 is the largest configuration (nv = 36)
#define RAPTOR_MAX_SPHERES 24     // Kinova: 17 (+3 with gripper)
#define RAPTOR_MAX_DEGREE 15

// pinocchio::Model fields consumed by the hot path, flattened.
struct DevModel {
    int nv;
    int serial_chain;  // parent[i] == i-1 for all i
    int parent[RAPTOR_MAX_NV];
    int jtype[RAPTOR_MAX_NV];
    double plR[RAPTOR_MAX_NV][9];  // jointPlacements rotation (row-major)
    double plp[RAPTOR_MAX_NV][3];
    double mass[RAPTOR_MAX_NV];
    double lever[RAPTOR_MAX_NV][3];
    double Ic[RAPTOR_MAX_NV][6];  // inertia about COM: xx, xy, xz, yy, yz, zz
    double neg_gravity[3];        // base acceleration -a_grav (linear part)
    double armature[RAPTOR_MAX_NV];
    double damping[RAPTOR_MAX_NV];
    double friction[RAPTOR_MAX_NV];
};

// eval_h torque rows (kinova_hess_torque_kernel.h): the serial chain's constants per link, every link frame re-labelled cyclically so
// that its joint axis is the third one (a proper rotation).  c[l] = { rows I, J, K of jointPlacement^T in the parent's labels (9),
// placement translation in the parent's labels (3), mass, lever (3), inertia about the COM xx xy xz yy yz zz (6), joint sign, pad }
struct DevHessLinks {
    double c[8][24];
};

// Trajectory: q_j(t_i) = sum_k coef(k, j) * basis[i][0][k]  (and [1] = d/dt, [2] = d2/dt2 bases).
// The basis tables depend only on the sample index and are built once on the host with the
// reference's arithmetic (BezierCurves.cpp:117-141, Polynomials.cpp:54-68).
struct DevTraj {
    int type;       // RAPTOR_TRAJ_*
    int degree;     // degree + 1 basis functions
    int K;          // free coefficients per joint = width of the basis tables: degree + 1 minus the rows pinned by
    int nfixed;     // BezierCurves::constrainInitialPosition / Velocity (0, 1, 2); their contribution is a per-(sample, joint) offset
    int N;          // samples
    int Nact;       // joints driven by the trajectory
    int varLength;  // decision variables (columns of pg_pz)
    // ArmourBezierCurves: coefficient rows 0..2 are constants, rows 3..5 are q0 + z
    double armour_c[3][RAPTOR_MAX_NV];
    double armour_q0[RAPTOR_MAX_NV];
};

// Kinova obstacle constraints (KinovaCustomizedConstraints + BoxCollisionAvoidance)
struct DevSpheres {
    int num_spheres;
    int num_obstacles;
    int joint_id[RAPTOR_MAX_SPHERES];  // Pinocchio joint id (1-based) the sphere is attached to
    int ro_slot[RAPTOR_MAX_NV];        // link -> slot of its world pose in the kernel's LDS exchange area (-1: no sphere)
    double offset[RAPTOR_MAX_SPHERES][3];
    double radius[RAPTOR_MAX_SPHERES];
    int slot[RAPTOR_MAX_SPHERES];      // = ro_slot[joint_id - 1]: one table look-up instead of two dependent ones in the kernel
};

// one oriented box: R (row-major), centre, full edge lengths as the reference keeps them; cl = R^T centre and hs = size / 2.0
// (exact) are what the distance loop of the fused Kinova kernel reads, laid out first so that they arrive in one run of scalar loads
struct DevBox {
    double R[9];
    double cl[3];
    double hs[3];
    double center[3];
    double size[3];
};

// Floating-base robot with a planted stance foot, re-rooted at the stance-foot link (legged_kernel.h).
// Bodies are numbered in DFS pre-order from that root, so subtree(g) = [g, body_end[g]).
#define RAPTOR_LEG_MAX_BODIES 32
#define RAPTOR_LEG_MAX_DEPTH 20
#define RAPTOR_LEG_MAX_IND 12
struct DevLegged {
    int nb;                                   // bodies (links that carry dynamics), body 0 = stance-foot link
    int ni, nd, nv, K;                        // independent / dependent joints, model nv, free coefficients per joint
    int body_link[RAPTOR_LEG_MAX_BODIES];     // original link (= joint) index whose inertia the body carries
    int body_joint[RAPTOR_LEG_MAX_BODIES];    // original joint connecting the body to its re-rooted parent (-1: root)
    int body_rev[RAPTOR_LEG_MAX_BODIES];      // 1: that joint is crossed child -> parent (axis flips sign)
    int body_rparent[RAPTOR_LEG_MAX_BODIES];
    int body_end[RAPTOR_LEG_MAX_BODIES];
    int body_depth[RAPTOR_LEG_MAX_BODIES];
    int body_traj[RAPTOR_LEG_MAX_BODIES];     // trajectory index (0..ni-1) driving the joint, -1: none
    int path[RAPTOR_LEG_MAX_BODIES][RAPTOR_LEG_MAX_DEPTH];  // path[g][d-1] = ancestor of g at depth d (d = 1..depth)
    int ind_body[RAPTOR_LEG_MAX_IND];         // body whose joint is independent joint t
    int ind_joint[RAPTOR_LEG_MAX_IND];        // model joint index of independent joint t
    int dep_joint[RAPTOR_MAX_NV];             // model joint index of dependent joint d
    int joint_ind[RAPTOR_MAX_NV];             // model joint -> independent index or -1
    int joint_dep[RAPTOR_MAX_NV];             // model joint -> dependent index or -1
    int torso_body;                           // body carried by the last floating-base joint
    double R0[9], o0[3];                      // world pose of the root body frame: T_des * endT^-1
    double Ades[9];                           // inverse of d(rpy)/d(omega) at R_des; lambda_rot = Ades^T n_p
    double p_foot[3];                         // contact point (stance_foot_T_des.p)
    double mu, gamma, Lx, Ly;                 // RectangleSurfaceContactConstraints
    // closed loops (Digit): a tree joint that is not driven by the trajectory is a function of <= 2 independent joints.
    // Every tree joint carries <= 2 chain-rule weight slots (G, Hv, Hq) towards minimal (independent) columns.
    int nrec;                                 // dependent loop joints per sample-eval (0: Talos / G1, 18: Digit)
    int body_rec[RAPTOR_LEG_MAX_BODIES];      // record of the body's joint in the loop buffer, -1: driven by the trajectory
    int body_col[RAPTOR_LEG_MAX_BODIES][2];   // minimal column of weight slot 0 / 1 (-1: unused)
    int col_nsupp[RAPTOR_LEG_MAX_IND];        // tree joints that move with independent joint t ...
    int col_supp_body[RAPTOR_LEG_MAX_IND][8]; // ... their bodies ...
    int col_supp_slot[RAPTOR_LEG_MAX_IND][8]; // ... and which of their weight slots points at t
    // the same tables packed for the core kernel, which reads them once into registers instead of chasing them through
    // dependent loads inside its loops.  path8: one body per byte.  col_pack entry (16 bits): body | end(body) << 5 | slot << 11 |
    // valid << 12.  flat entry (32 bits): body | end(body) << 5 | slot << 11 | last-of-column << 12 | mode << 13 | column of set 0 << 15 |
    // column of set 1 << 19 (4 bits each, 15 = none; mode 0 / 1: into set 0 / 1 with weight slot `slot`, 2: into both sets).
    // The 5-bit body field needs RAPTOR_LEG_MAX_BODIES <= 32, the 4-bit column fields RAPTOR_LEG_MAX_IND <= 15 (static_asserts in legged_setup).
    unsigned long long path8[RAPTOR_LEG_MAX_BODIES][3];     // path[g][0 .. depth-1], one byte each
    unsigned long long col_pack[RAPTOR_LEG_MAX_IND][2];     // col_supp_*[t][0..7], 16 bits each, flag = entry valid
    int flat_n, flat[64];                                   // all (column, support joint) pairs in column order, flag = last of its column
    // world kinematics by pointer doubling (legged_kernel.h phase A): per body, for round r = 0..4, its ancestor 2^r levels up
    // (the root included) as 5 bits + a valid bit, 6 bits per round; nrounds = rounds needed for the deepest body
    unsigned jump_pk[RAPTOR_LEG_MAX_BODIES];
    int nrounds;
    int ind_ndep[RAPTOR_LEG_MAX_IND];         // dependent loop joints of independent joint t
    int ind_dep_body[RAPTOR_LEG_MAX_IND][6];
    int ind_slot[RAPTOR_LEG_MAX_IND];         // slot of t in those joints' records (0: driver a, 1: driver b)
    int ind_loop_col[RAPTOR_LEG_MAX_IND][2];  // minimal columns of the drivers (a, b) of t's loop, -1: none
    int dep_col[RAPTOR_MAX_NV][2];            // by dependent index (>= 6): minimal columns of its two P_dep entries
    int dep_body[RAPTOR_MAX_NV];              // by dependent index (>= 6): the body whose joint it is (0 for the floating base)
    // robot-specific "customized" block (swing-foot / torso rows): rows are linear in 12 pose quantities of ONE sample,
    // X[0..5] = swing-foot x y z roll pitch yaw, X[6..11] = floating-base coordinates q[0..5]
    int ncust, off_cust;                      // rows (0: block absent), offset in Optimizer row order
    int sf_body;                              // body that carries the swing foot
    double sf_R[9], sf_p[3];                  // swing-foot end transform
    int off_tau, off_q, off_contact;          // block offsets in Optimizer row order
    int nperiod, off_period;                  // reset-map block (period_kernel.h): rows (0: absent), offset
    // per-sample-eval record written by the core kernel (doubles):
    //   tau[ni] lambda[6] q[nv] v[nv] a[nv] | D[it][x][ni] (x = q, v, a; rows = actuators) | L[x][6][it] |
    //   Pb[6][it] (floating-base rows of P_dep) | Pl[nd - 6][2] (closed-loop rows of P_dep, columns dep_col)
    //   sf[6] swing-foot x y z roll pitch yaw | Psf[6][it] its derivative per unit actuator displacement
    //   (`it` innermost where the core kernel's column lanes are the writers: their stores are runs of 8 * ni bytes)
    int c_tau, c_lam, c_q, c_v, c_a, c_D, c_L, c_P, c_Pl, c_sf, c_Psf, c_stride;
};

// one row of the customized block: g = wrap?( ca * X[qa] + cb * X[qb] + add ), X of sample `sample`
struct DevCustRow {
    int sample, qa, qb, wrap;
    double ca, cb, add;
};

// Reset-map ("periodicity") block, period_kernel.h.  The model's own tree (rooted at the world, floating-base joints
// included) in DFS pre-order: lane = joint = the link it moves, subtree(g) = [g, end[g]).
#define RAPTOR_PERIOD_MAX_LOOPS 4
struct DevPeriod {
    int nj;                                   // joints (= model nv)
    int off;                                  // first row of the block in Optimizer order
    int lane_joint[64];                       // lane -> model joint
    int joint_lane[RAPTOR_MAX_NV];
    int end[64], depth[64];
    int path[64][RAPTOR_LEG_MAX_DEPTH];       // lanes from the root down to and including this one
    unsigned long long path8[64][RAPTOR_LEG_MAX_DEPTH / 4];  // the same as bytes, then the model joints of those lanes as bytes
    int old_foot_lane, new_foot_lane;         // stance foot of the step, stance foot after the swap
    double endR[9], endp[3];                  // stance_foot_endT after the swap
    int id1[RAPTOR_LEG_MAX_IND], id2[RAPTOR_LEG_MAX_IND];  // JOINT_MAP as model joints
    double sign;                              // g3 = q_0(id1) + sign * q_minus(id2)
    // closed loops (Digit): loop l = leg * 2 + (0: knee, 1: toe); per decision vector a record
    //   J[nr][nvl] | J v+ [nr] | J^T lambda [nvl] | D(J v+)[nr][nx] | D(J^T lambda)[nvl][nx]      (digit_period_loops_kernel)
    int nloops, loop_stride;
    int loop_nvl[RAPTOR_PERIOD_MAX_LOOPS], loop_nr[RAPTOR_PERIOD_MAX_LOOPS], loop_nx[RAPTOR_PERIOD_MAX_LOOPS];
    int loop_row0[RAPTOR_PERIOD_MAX_LOOPS];   // first constraint row (= first lambda entry) of the loop
    int loop_col[RAPTOR_PERIOD_MAX_LOOPS][2]; // minimal columns of its drivers
    int loop_off[RAPTOR_PERIOD_MAX_LOOPS];    // offset of its record
    int joint_loop[RAPTOR_MAX_NV], joint_var[RAPTOR_MAX_NV];  // model joint -> loop / local variable (-1: none)
    int joint_rec[RAPTOR_MAX_NV];             // dependent loop joint -> its record in the per-sample loop buffer
    int row_loop[24], row_local[24];          // closure row -> loop / row inside the loop
};

// where each constraint block of this problem starts in the Optimizer's row order (-1: block absent)
struct DevLayout {
    int m;        // rows per decision vector
    int n;        // columns (= varLength)
    int off_q;    // JointLimits rows        [N * Nact]
    int off_qd;   // VelocityLimits rows     [N * Nact]
    int off_tau;  // TorqueLimits rows       [N * Nact]
    int off_obs;  // obstacle rows           [N * num_spheres]
};

#endif
