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

// Trajectory: q_j(t_i) = sum_k coef(k, j) * basis[i][0][k]  (and [1] = d/dt, [2] = d2/dt2 bases).
// The basis tables depend only on the sample index and are built once on the host with the
// reference's arithmetic (BezierCurves.cpp:117-141, Polynomials.cpp:54-68).
struct DevTraj {
    int type;       // RAPTOR_TRAJ_*
    int degree;     // K = degree + 1 basis functions
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
};

// one oriented box: R (row-major), centre, half sizes stay as the reference keeps them (full size / 2.0)
struct DevBox {
    double R[9];
    double center[3];
    double size[3];
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
