/*
 * raptor_hip.h -- C ABI of the MI355X (gfx950) implementation of RAPTOR's constraint-evaluation
 * hot path:  z -> trajectory -> fp64 inverse dynamics + d(tau)/dz -> constraint rows g / pg_pz.
 *
 * The reference (roahmlab/RAPTOR) has no FFI; its plug-in surface is the public-member C++ classes
 * Trajectories / InverseDynamics / Constraints / Optimizer.  Every entry point below names the
 * reference interface it stands in for (file:line under the reference checkout), and
 * INTEGRATION.md shows the adapter a RAPTOR maintainer would add on top of it.
 *
 * Conventions
 *   - all floating point is IEEE binary64, all indices are int (as in the reference);
 *   - matrices handed across the boundary are ROW-MAJOR unless stated otherwise
 *     (Optimizer::eval_jac_g emits pg_pz row-major: Optimization/src/Optimizer.cpp:480-485);
 *   - every function returns RAPTOR_OK or an error code; raptor_last_error() gives the text.
 *     Codes mirror the exception classes the reference throws (std::invalid_argument,
 *     std::runtime_error), no exception ever crosses this boundary;
 *   - "d_" pointers are device (HBM) pointers, everything else is host memory;
 *   - objects are not thread-safe, exactly like the reference's.  A raptor_problem also owns
 *     device scratch (per-sample records, partial results) shared by all of its entry points: ASYNCHRONOUS calls on one
 *     problem must be ordered after one another by the caller -- the same hip_stream, or an event / synchronisation
 *     between streams (raptor_eval_batch on one stream and raptor_eval_costs / raptor_eval_h_batch for the SAME problem on
 *     another, unordered, race on that scratch).  A *_host call is ordered by the library itself: every asynchronous entry
 *     point records an event on the caller's stream once its work is enqueued, and every *_host entry point waits for the
 *     problem's latest such event before it touches the scratch.  Different problems are independent.
 *   - *_host entry points are synchronous.  raptor_eval_batch_host, raptor_eval_batch_status_host and raptor_eval_h_batch_host
 *     (the per-iteration calls of a solver) stage through HBM on a private non-blocking stream of the problem and wait for
 *     that stream only (small transfers through a pinned buffer), never for the whole device; the other *_host entry
 *     points (component views, costs, objective, step link, dual arm, capsules) run on the null stream and end with
 *     hipDeviceSynchronize.
 *
 * The same signatures are exported by the CPU oracle (oracle/liboracle.so) with the prefix
 * oracle_ instead of raptor_ (host pointers everywhere); the parity tests call both.
 */
#ifndef RAPTOR_HIP_H
#define RAPTOR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RAPTOR_OK 0
#define RAPTOR_ERR_INVALID_ARGUMENT 1 /* reference: throw std::invalid_argument */
#define RAPTOR_ERR_RUNTIME 2          /* reference: throw std::runtime_error   */
#define RAPTOR_ERR_HIP 3              /* a HIP runtime call failed             */
#define RAPTOR_ERR_UNSUPPORTED 4      /* valid in the reference, not built here yet */

/* Trajectories/include/Trajectories.h:17-20 (TimeDiscretization) */
#define RAPTOR_TIME_UNIFORM 0
#define RAPTOR_TIME_CHEBYSHEV 1

/* trajectory primitives on the hot path */
#define RAPTOR_TRAJ_ARMOUR_BEZIER 0 /* Trajectories/src/ArmourBezierCurves.cpp:44-163 */
#define RAPTOR_TRAJ_BEZIER 1        /* Trajectories/src/BezierCurves.cpp:21-182      */
#define RAPTOR_TRAJ_POLYNOMIAL 2    /* Trajectories/src/Polynomials.cpp:15-91 (used by the reference's ID tests) */
#define RAPTOR_TRAJ_PIECEWISE_BEZIER 3 /* Trajectories/src/PiecewiseBezierCurves.cpp:5-255 (KinovaLongerHorizonOptimizer) */

/* what raptor_eval_batch computes: G only == Constraints::compute(z,false) (eval_g),
 * JAC == Constraints::compute(z,true) (eval_jac_g).  */
#define RAPTOR_EVAL_G 1u
#define RAPTOR_EVAL_JAC 2u

/*
 * Flat rigid-body model = the fields of pinocchio::Model the hot path reads
 * (KinematicsDynamics/src/CustomizedInverseDynamics.cpp:43-59, ForwardKinematics.cpp:70,137).
 * Joint i here is Pinocchio joint i+1; every joint has one degree of freedom (nq == nv).
 */
typedef struct raptor_model_desc {
    int nv;
    const int* parent;         /* [nv]  model.parents[i+1]-1, -1 = universe                    */
    const int* jtype;          /* [nv]  convertPinocchioJointType codes (Transform.cpp:5-54)   */
    const double* placement_R; /* [nv*9] model.jointPlacements[i+1].rotation(), row-major     */
    const double* placement_p; /* [nv*3] model.jointPlacements[i+1].translation()             */
    const double* mass;        /* [nv]   model.inertias[i+1].mass()                           */
    const double* lever;       /* [nv*3] model.inertias[i+1].lever()                          */
    const double* inertia;     /* [nv*9] model.inertias[i+1].inertia().matrix(), row-major    */
    double gravity[3];         /* model.gravity.linear()                                       */
    const double* armature;    /* [nv] model.armature  (NULL = zeros)                          */
    const double* damping;     /* [nv] model.damping   (NULL = zeros)                          */
    const double* friction;    /* [nv] model.friction  (NULL = zeros)                          */
} raptor_model_desc;

/* Constructor arguments of the trajectory classes (T, N, Nact, discretisation, degree [, atp]). */
typedef struct raptor_traj_desc {
    int type;                /* RAPTOR_TRAJ_*                                                  */
    int degree;              /* ignored for ARMOUR_BEZIER (fixed 5)                            */
    double T;                /* duration                                                       */
    int N;                   /* number of time samples                                         */
    int time_discretization; /* RAPTOR_TIME_*                                                  */
    int Nact;                /* actuated joints driven by the trajectory                       */
    const double* q0;        /* [Nact] ARMOUR_BEZIER: ArmourTrajectoryParameters.  BEZIER: non-null = constrainInitialPosition(q0) */
    const double* q_d0;      /* [Nact] BEZIER: non-null (with q0) = constrainInitialVelocity(q_d0); each drops Nact variables
                                (BezierCurves.cpp:39-69, second coefficient row q0 + q_d0 T / 5.0 as written at :86)         */
    const double* q_dd0;     /* [Nact]                                                          */
    const double* qT;        /* [Nact] PIECEWISE_BEZIER only: end position; NULL = optimised, stored at the end of z.  There q0 is
                                the begin position (NULL = optimised too), `degree` the number of interior control points
                                (degree + 1 quintic pieces, N a multiple of degree + 1), time_discretization is ignored
                                (always LinSpaced, PiecewiseBezierCurves.cpp:22)                                             */
} raptor_traj_desc;

/* Arguments of KinovaOptimizer::set_parameters
 * (Examples/Kinova/CollisionAvoidanceTrajectory/src/KinovaOptimizer.cpp:18-116). */
typedef struct raptor_kinova_desc {
    raptor_traj_desc traj;               /* ARMOUR_BEZIER (KinovaOptimizer) or PIECEWISE_BEZIER (KinovaLongerHorizonOptimizer:
                                            Examples/Kinova/CollisionAvoidanceTrajectory/src/KinovaLongerHorizonOptimizer.cpp:18-122,
                                            same four blocks), Nact == model nv == 7 */
    int num_obstacles;
    const double* box_center;            /* [O*3] boxCenters                                    */
    const double* box_rpy;               /* [O*3] boxOrientation (roll,pitch,yaw)               */
    const double* box_size;              /* [O*3] boxSize (full edge lengths)                   */
    int include_gripper;                 /* adds spheres 17..19 on joint 7                      */
    double collision_buffer;
    const double* joint_limits_buffer;   /* [7] rad, NULL = 0                                   */
    const double* velocity_limits_buffer;/* [7] rad/s, NULL = 0                                 */
    const double* torque_limits_buffer;  /* [7] N m,  NULL = 0                                  */
} raptor_kinova_desc;

/*
 * The robot-specific "customized" kinematic block (swing-foot and torso rows):
 *   DigitCustomizedConstraints  Examples/Digit/src/DigitCustomizedConstraints.cpp:35-200   (N*10 + 4 rows)
 *   TalosCustomizedConstraints  Examples/Talos/src/TalosCustomizedConstraints.cpp:35-200   (N*11 + 2 rows; G1 twin)
 * `enabled == 0` leaves the block out.  The fields are the reference's GaitParameters plus the constants its two
 * variants differ in.
 */
typedef struct raptor_gait_desc {
    int enabled;
    int swing_joint_id;             /* model.getJointId(swing foot name), Pinocchio id (1-based); Digit: derived, ignored */
    double swingfoot_endT_p[3];     /* Talos / G1: leftfoot_endT.p == rightfoot_endT.p; Digit: compiled in              */
    double eps_torso_angle;
    double swingfoot_midstep_z_des;
    double swingfoot_begin_x_des, swingfoot_begin_y_des, swingfoot_end_x_des, swingfoot_end_y_des;
    double torso_height_lb;         /* 0.95 Talos, 0.55 G1 (g6_lb); Digit: 1.05 compiled in, ignored                     */
} raptor_gait_desc;

/*
 * The reset-map ("periodicity") block that closes a single walking step:
 *   DigitSingleStepPeriodicityConstraints  Examples/Digit/src/DigitSingleStepPeriodicityConstraints.cpp:6-226
 *   TalosSingleStepPeriodicityConstraints  Examples/Talos/src/TalosSingleStepPeriodicityConstraints.cpp:6-226 (G1 twin)
 * rows: [H (v+ - v-) - J^T lambda (nv) | J v+ (n_dep) | position reset (Nact) | velocity reset (Nact) | contact on lambda (7)]
 * with q-, v- the last sample, J the constraint Jacobian after the stance swap, v+ = z.segment(varLength - nv - n_dep, nv)
 * and lambda = z.tail(n_dep): `extra_variables` must be nv + n_dep.  `enabled == 0` leaves the block out.
 */
#define RAPTOR_PERIOD_LINK_ONLY 2   /* period.enabled: build the tables for raptor_step_link_eval_batch, add no rows */
typedef struct raptor_period_desc {
    int enabled;                    /* 0: absent, 1: the single-step block, RAPTOR_PERIOD_LINK_ONLY (multiple-step members) */
    int swing_contact_joint_id;     /* contact_joint_id after dcPtr_->reinitialize() (Pinocchio id); Digit: derived    */
    double swing_foot_endT_p[3];    /* stance_foot_endT.p after the swap; Digit: compiled in                           */
    const int* joint_id1;           /* [num_independent] getJointId(JOINT_MAP[i][0]) - 1; Digit: derived               */
    const int* joint_id2;           /* [num_independent] getJointId(JOINT_MAP[i][1]) - 1                               */
    double mirror_sign;             /* g3 = q_0(id1) + mirror_sign * q_minus(id2): -1 Talos / G1; Digit: +1, derived   */
} raptor_period_desc;

/*
 * Floating-base robot whose six base joints are made dependent by a planted stance foot: the arguments of
 * TalosSingleStepOptimizer::set_parameters / G1SingleStepOptimizer::set_parameters that reach the hot path
 * (Examples/Talos/src/TalosSingleStepOptimizer.cpp:17-124, TalosDynamicsConstraints.cpp:6-42, same for Unitree-G1).
 * Blocks built: [torque limits N*Nact | joint limits N*nv | contact N*7 | customized N*11 + 2 (optional) |
 * reset map nv + 6 + 2 Nact + 7 (optional)]
 * (TorqueLimits, ConstrainedJointLimits, RectangleSurfaceContactConstraints, *CustomizedConstraints,
 * *SingleStepPeriodicityConstraints).
 */
typedef struct raptor_legged_desc {
    raptor_traj_desc traj;          /* RAPTOR_TRAJ_BEZIER, Nact = num_independent                                 */
    int extra_variables;            /* appended to varLength: NUM_JOINTS + NUM_DEPENDENT_JOINTS (v_reset, lambda_reset) */
    int num_dependent;              /* 6                                                                         */
    const int* dependent_ids;       /* [num_dependent]   model.getJointId(name) - 1 of Px,Py,Pz,Rx,Ry,Rz          */
    int num_independent;
    const int* independent_ids;     /* [num_independent] model.getJointId(name) - 1 of the actuated joints        */
    int fk_start_joint;             /* model.getJointId("Rz")  (TalosDynamicsConstraints.cpp:214)                  */
    int contact_joint_id;           /* model.getJointId(LEFT_FOOT_NAME or RIGHT_FOOT_NAME)                        */
    double stance_foot_endT_p[3];   /* stance_foot_endT.p      (TalosDynamicsConstraints.cpp:34)                   */
    double stance_foot_T_des_R[9];  /* stance_foot_T_des.R row-major (default identity)                            */
    double stance_foot_T_des_p[3];
    const double* torque_limits_lower; /* [num_independent] */
    const double* torque_limits_upper;
    const double* joint_limits_lower;  /* [nv] */
    const double* joint_limits_upper;
    double mu, gamma, Lx, Ly;       /* rectangleContactSurfaceParams(MU, GAMMA, FOOT_WIDTH, FOOT_LENGTH)           */
    int stance_leg;                 /* 'L' or 'R' (only the customized block reads it)                             */
    raptor_gait_desc gait;          /* fourth block: TalosCustomizedConstraints / G1CustomizedConstraints          */
    raptor_period_desc period;      /* fifth block: TalosSingleStepPeriodicityConstraints / G1 twin                */
} raptor_legged_desc;

/*
 * Digit-v3 single-step problem: the arguments of DigitSingleStepOptimizer::set_parameters that reach the hot path
 * (Examples/Digit/src/DigitSingleStepOptimizer.cpp:17-124) plus the joint ids DigitDynamicsConstraints looks up by name
 * (Examples/Digit/include/DigitDynamicsConstraints.h:19-57).  The closed-loop constants (rod / anchor end points,
 * stance-foot end transform, initial-guess fit) are compiled in, as the reference hard-codes them
 * (DigitDynamicsConstraints.h:178-198, DigitDynamicsConstraints.cpp:255-314).
 * Blocks built: [torque limits N*12 | joint limits N*36 | contact N*7 | customized N*10 + 4 (optional) |
 * reset map 91 (optional)].
 */
typedef struct raptor_digit_desc {
    raptor_traj_desc traj;          /* RAPTOR_TRAJ_BEZIER, Nact = 12                                               */
    int extra_variables;            /* NUM_JOINTS + NUM_DEPENDENT_JOINTS appended to varLength                    */
    int base_ids[6];                /* model.getJointId(name) - 1 of Px, Py, Pz, Rx, Ry, Rz                        */
    int leg_ids[2][15];             /* [0] left, [1] right: hip_roll, hip_yaw, hip_pitch, achilles_rod, ach2, knee,
                                       tarsus, toe_A, toe_A_rod, A2, toe_B, toe_B_rod, B2, toe_pitch, toe_roll     */
    int stance_leg;                 /* 'L' or 'R'                                                                  */
    double stance_foot_T_des_R[9];  /* row-major; default Transform(3, -pi/2) (DigitSingleStepOptimizer.h:50)      */
    double stance_foot_T_des_p[3];
    const double* torque_limits_lower; /* [12] */
    const double* torque_limits_upper;
    const double* joint_limits_lower;  /* [36] radians */
    const double* joint_limits_upper;
    double mu, gamma, Lx, Ly;
    raptor_gait_desc gait;          /* fourth block: DigitCustomizedConstraints                                    */
    raptor_period_desc period;      /* fifth block: DigitSingleStepPeriodicityConstraints (only `enabled` is read) */
} raptor_digit_desc;

typedef struct raptor_model raptor_model;     /* opaque */
typedef struct raptor_problem raptor_problem; /* opaque */

/* Text of the last error raised on the calling thread. */
const char* raptor_last_error(void);

/* Replaces holding a pinocchio::Model (InverseDynamics ctor, InverseDynamics.cpp:5-11). */
int raptor_model_create(const raptor_model_desc* desc, raptor_model** out);
void raptor_model_destroy(raptor_model* model);

/*
 * Inverse-dynamics-only problem: Trajectories + InverseDynamics + TorqueLimits rows.
 * One constraint block "torque": g[i*nv+j] = tau(i)(j), pg_pz rows = ptau_pz(i)
 * (KinematicsDynamics/src/InverseDynamics.cpp:73-139, Constraints/src/TorqueLimits.cpp:18-43).
 * Mirrors how the reference's own TestInverseDynamics drives the class
 * (Tests/KinematicsDynamics/TestInverseDynamics.cpp:13-46).
 */
int raptor_id_problem_create(const raptor_model* model, const raptor_traj_desc* traj, raptor_problem** out);

/* KinovaOptimizer::set_parameters: blocks [joint limits | velocity limits | torque limits | obstacles]
 * (KinovaOptimizer.cpp:82-109). */
int raptor_kinova_problem_create(const raptor_model* model, const raptor_kinova_desc* desc, raptor_problem** out);

/* TalosSingleStepOptimizer / G1SingleStepOptimizer set_parameters: ConstrainedInverseDynamics + the three blocks above
 * (KinematicsDynamics/src/ConstrainedInverseDynamics.cpp:45-184, DynamicsConstraints.cpp:36-103). */
int raptor_legged_problem_create(const raptor_model* model, const raptor_legged_desc* desc, raptor_problem** out);

/* DigitSingleStepOptimizer::set_parameters: DigitConstrainedInverseDynamics (closed-loop DigitDynamicsConstraints,
 * Examples/Digit/src/DigitDynamicsConstraints.cpp:242-400) + the three blocks above. */
int raptor_digit_problem_create(const raptor_model* model, const raptor_digit_desc* desc, raptor_problem** out);

/* ConstrainedInverseDynamics::compute results per sample, host buffers (ConstrainedInverseDynamics.h:44-56); any may be NULL:
 *   q, v, a [B][N][nv] (dependent joints filled in), tau [B][N][Nact], lambda [B][N][6] = lambda.tail(6),
 *   the stance-foot contact wrench (the rows RectangleSurfaceContactConstraints reads),
 *   ptau_pz [B][N][Nact][n], plambda_pz [B][N][6][n]  (row-major) */
int raptor_constrained_id_eval_host(raptor_problem* problem, const double* Z, int B, double* q, double* v, double* a,
                                    double* tau, double* lambda, double* ptau_pz, double* plambda_pz);

/*
 * ConstrainedInverseDynamics::lambda(i) and plambda_pz(i) with ALL n_dep rows, as the reference's members hold them
 * (KinematicsDynamics/include/ConstrainedInverseDynamics.h:44-56, src/ConstrainedInverseDynamics.cpp:154, 169-171):
 *     lambda     [B][N][n_dep]        plambda_pz [B][N][n_dep][n]   (may be NULL)
 * Talos / Unitree-G1: n_dep = 6, the contact wrench (the same numbers as raptor_constrained_id_eval_host).  Digit: n_dep = 24,
 * rows 0..17 the multipliers of the closed-loop (rod) constraints in the reference's constraint order
 * (Examples/Digit/src/DigitDynamicsConstraints.cpp:579-596), rows 18..23 the contact wrench.  Component view (host buffers,
 * synchronous, allocates its own scratch): no constraint or cost of the reference reads rows 0..17, so raptor_eval_batch
 * never computes them.
 */
int raptor_constrained_id_lambda_host(raptor_problem* problem, const double* Z, int B, double* lambda, double* plambda_pz);

void raptor_problem_destroy(raptor_problem* problem);

/* Optimizer::get_nlp_info (KinovaOptimizer.cpp:121-147): n = varLength, m = sum of block sizes.
 * block_offsets (may be NULL) receives num_blocks+1 row offsets, blocks in constraintsPtrVec_ order. */
int raptor_problem_layout(const raptor_problem* problem, int* n, int* m, int* num_blocks, int* block_offsets);

/* Constraints::compute_bounds of every block, concatenated (Optimizer::get_bounds_info). */
int raptor_problem_bounds(const raptor_problem* problem, double* g_lb, double* g_ub);

/* Trajectories::tspan (Trajectories/src/Trajectories.cpp:54-64). */
int raptor_problem_tspan(const raptor_problem* problem, double* tspan);

/*
 * The hot path.  For each of the B decision vectors (d_Z is B x n row-major) evaluate every
 * constraint block and write
 *     d_G   [B x m]      g in Optimizer order            (Optimizer::eval_g,     Optimizer.cpp:361-385)
 *     d_JAC [B x m x n]  pg_pz row-major, Optimizer order (Optimizer::eval_jac_g, Optimizer.cpp:467-486)
 * d_G may be NULL when only JAC is wanted and vice versa; flags says what to compute.
 * Asynchronous on hip_stream (a hipStream_t passed as void*, NULL = the default stream).
 */
int raptor_eval_batch(raptor_problem* problem, const double* d_Z, int B, unsigned flags,
                      double* d_G, double* d_JAC, void* hip_stream);

/* Same with host buffers (what Optimizer::eval_g / eval_jac_g hand over); copies through HBM, synchronous. */
int raptor_eval_batch_host(raptor_problem* problem, const double* Z, int B, unsigned flags,
                           double* G, double* JAC);

/*
 * Multi-start support (no counterpart in the reference, which solves one problem per process).
 * raptor_shard_range: the block partition of `total` independent problems over `world` ranks (one process per GPU): rank r
 *   owns [first, first + count), blocks of ceil(total / world), the last ones short or empty.  Pure arithmetic, no GPU.
 * raptor_violation_batch: the per-problem record the ranks exchange -- the worst bound violation of each decision vector,
 *   d_violation[b] = max_r max(g_lb[r] - g[b][r], g[b][r] - g_ub[r], 0), NaN if a row is NaN; a problem is feasible in the
 *   sense of Optimizer::eval_g (Optimization/src/Optimizer.cpp:374-378) when d_violation[b] <= constr_viol_tol.
 *   d_G = the [B x m] output of raptor_eval_batch; asynchronous on hip_stream, stream-ordered after the call that wrote d_G.
 *   Allocates and uploads nothing: the bounds are put on the device by the *_problem_create call.
 * A C++ host gathers d_violation (B doubles per rank, padded to ceil(total / world)) with one ncclAllGather: INTEGRATION.md,
 * or calls raptor_eval_batch_sharded below, which does the three steps.
 */
int raptor_shard_range(long long total, int world, int rank, long long* first, long long* count);
int raptor_violation_batch(raptor_problem* problem, const double* d_G, int B, double* d_violation, void* hip_stream);

/*
 * raptor_eval_batch_sharded: one sweep of a multi-start batch over the ranks of a node (no counterpart in the
 * reference).  Every rank holds the same d_Z_all [total x n]; this rank (one process per GPU) evaluates its own block
 * [first, first + count) of raptor_shard_range through raptor_eval_batch (d_G [count x m], d_JAC [count x m x n]: this rank's rows
 * only, Jacobians stay on the GPU that produced them), writes the per-problem record of raptor_violation_batch into its slot of
 * d_rec_all and completes d_rec_all [world x per], per = ceil(total / world) (padded slots: 0), with ONE in-place ncclAllGather on
 * `nccl_comm` -- the path's only exchange.  d_rec_all[r * per + k] = worst bound violation of problem r * per + k on every rank.
 *   nccl_comm  an ncclComm_t of the caller's RCCL (RCCL's API is NCCL's), created over the `world` ranks.  RCCL is not linked into
 *              this library: ncclAllGather is looked up in the process at the first call (the caller has loaded librccl.so.1 to
 *              create the communicator).  NULL is accepted with world == 1 only and skips the collective.
 * Asynchronous on hip_stream like raptor_eval_batch; the collective is enqueued on the same stream.
 */
int raptor_eval_batch_sharded(raptor_problem* problem, const double* d_Z_all, long long total, int world, int rank, unsigned flags,
                              double* d_G, double* d_JAC, double* d_rec_all, void* nccl_comm, void* hip_stream);

/*
 * raptor_eval_batch plus a per-z report of the dependent-joint root finding (SURVEY.md 8b `int* status`).  The reference
 * runs gsl_multiroot_fdfsolver_hybridsj for at most 50 iterations per time sample and silently accepts whatever it holds at the
 * end (Examples/Digit/src/DigitDynamicsConstraints.cpp:344-367); so does this library -- the rows of a z whose closed loops
 * did not close are "what the solver returned" -- but here the caller can see it:
 *   status[b]     bit RAPTOR_STATUS_NOT_CONVERGED: a closed-loop solve of some sample of z_b ended at the iteration cap with
 *                 sum|c| >= 1e-14;  bit RAPTOR_STATUS_NONFINITE: one produced a NaN / Inf (singular loop Jacobian);  0: all closed
 *   iterations[b] (may be NULL) the largest iteration count over the samples and loops of z_b
 * Problems without a root finder (Kinova, Talos, Unitree-G1: closed forms) report 0 / 0.
 */
#define RAPTOR_STATUS_NOT_CONVERGED 1
#define RAPTOR_STATUS_NONFINITE 2
int raptor_eval_batch_status(raptor_problem* problem, const double* d_Z, int B, unsigned flags, double* d_G, double* d_JAC,
                             int* d_status, int* d_iterations, void* hip_stream);
int raptor_eval_batch_status_host(raptor_problem* problem, const double* Z, int B, unsigned flags, double* G, double* JAC,
                                  int* status, int* iterations);

/*
 * Component views used by the parity tests, host buffers, synchronous:
 *   Trajectories::compute -> q, q_d, q_dd  [B][N][Nact]   (Trajectories.h:71-76); any may be NULL
 *   InverseDynamics::compute -> tau [B][N][nv], ptau_pz [B][N][nv][n] row-major (InverseDynamics.h:83-85)
 */
int raptor_trajectory_eval_host(raptor_problem* problem, const double* Z, int B,
                                double* q, double* q_d, double* q_dd);
int raptor_inverse_dynamics_eval_host(raptor_problem* problem, const double* Z, int B,
                                      double* tau, double* ptau_pz);

/*
 * TaperedCapsuleCollision<F>::computeDistance, batched (Constraints/src/TaperedCapsuleCollision.cpp:43-289).
 * P points to 4 endpoints x 3 coordinates (tc1_point_1, tc1_point_2, tc2_point_1, tc2_point_2) per pair,
 * dP to the four 3 x F Jacobians (row-major, may be NULL: distance only), R to the 4 radii.
 *   d_P [n_pairs][4][3], d_dP [n_pairs][4][3][F], d_R [n_pairs][4] -> d_dist [n_pairs], d_grad [n_pairs][F]
 */
int raptor_capsule_distance_batch(const double* d_P, const double* d_dP, const double* d_R,
                                  int n_pairs, int F, double* d_dist, double* d_grad, void* hip_stream);
int raptor_capsule_distance_batch_host(const double* P, const double* dP, const double* R,
                                       int n_pairs, int F, double* dist, double* grad);

/*
 * The block that links walking step s to step s + 1 in the multiple-step optimizers:
 * DigitMultipleStepPeriodicityConstraints::compute (Examples/Digit/src/DigitMultipleStepPeriodicityConstraints.cpp:57-230) and
 * its Talos / Unitree-G1 twins.  `curr` is step s, created with period.enabled (its reset-map tables: stance swap, contact
 * parameters); `next` is step s + 1 (the other stance leg; the last link of a periodic gait takes step 0).  Rows as in the
 * single-step block except the reset rows: g3 = q_next(0)[independent] - q-[independent], g4 = v_next(0)[independent] - v+[independent].
 *   Zcurr, Znext [B][n] -> G [B][mp], JAC = pg_pz [B][mp][n] (w.r.t. Zcurr), JAC2 = rows of pg3_pz2, pg4_pz2 [B][2 Nact][n]
 *   (w.r.t. Znext), mp = nv + n_dep + 2 Nact + 7.  The steps' own blocks come from raptor_eval_batch on each step's problem,
 *   as DigitMultipleStepOptimizer calls its DigitSingleStepOptimizer members (DigitMultipleStepOptimizer.cpp:36-60).
 */
int raptor_step_link_eval_batch(raptor_problem* curr, raptor_problem* next, const double* d_Zcurr, const double* d_Znext, int B,
                                unsigned flags, double* d_G, double* d_JAC, double* d_JAC2, void* hip_stream);
int raptor_step_link_eval_batch_host(raptor_problem* curr, raptor_problem* next, const double* Zcurr, const double* Znext, int B,
                                     unsigned flags, double* G, double* JAC, double* JAC2);

/*
 * Arm-arm collision rows of DualKinovaOptimizer::eval_g / eval_jac_g
 * (Examples/Kinova/CollisionAvoidanceTrajectory/src/DualKinovaOptimizer.cpp:269-305, 449-510): for two Kinova problems with
 * the same N, per sample the four pairs (tapered capsule 1, 2 of robot 1) x (tapered capsule 1, 2 of robot 2), each capsule
 * spanned by two collision spheres (KinovaCustomizedConstraints.h:58-63; the gripper extends the last capsule).
 *   Z [B][numVars1 + numVars2] (robot 1's variables first) -> G [B][N*4], JAC [B][N*4][numVars1 + numVars2] row-major.
 * The members may be ARMOUR_BEZIER or PIECEWISE_BEZIER problems (the reference's members are KinovaLongerHorizonOptimizers, :41-42).
 * The two robots' own blocks come from raptor_eval_batch on each problem, as the reference calls the two KinovaOptimizers.
 */
int raptor_dual_kinova_arm_arm_batch(raptor_problem* problem1, raptor_problem* problem2, const double* d_Z, int B,
                                     unsigned flags, double* d_G, double* d_JAC, void* hip_stream);
int raptor_dual_kinova_arm_arm_batch_host(raptor_problem* problem1, raptor_problem* problem2, const double* Z, int B,
                                          unsigned flags, double* G, double* JAC);

/*
 * Per-kernel timing with HIP events recorded on the evaluation stream, for measurement harnesses
 * (the reference keeps per-callback wall-clock counters instead: Optimization/src/Optimizer.cpp:717-754).
 * Between _begin and _end every raptor_eval_batch records events around its kernel launches, up to
 * max_evaluations calls; _end waits for them and returns the average duration in milliseconds of the
 * evaluation kernel (id_ms_avg) and of any follow-up kernel (obs_ms_avg; 0 since the Kinova path is one
 * fused kernel).
 */
int raptor_profile_begin(raptor_problem* problem, int max_evaluations);
int raptor_profile_end(raptor_problem* problem, int* evaluations, double* id_ms_avg, double* obs_ms_avg);
/*
 * Same, three intervals (ms_avg[3]): the events of an evaluation are start | split 1 | split 2 | end.  Legged problems whose
 * core kernel also writes the pg_pz rows (the fused form, Digit): closed-loop kernel | core kernel | reset-map kernels.
 * Every other path: kernels before the split | 0 | kernels after it (raptor_profile_end returns the first two added up).
 */
int raptor_profile_end_split(raptor_problem* problem, int* evaluations, double* ms_avg);

/*
 * Costs::compute for the cost classes the single-step optimizers use (Costs/include/Costs.h:25-45):
 *   RAPTOR_COST_MINIMIZE_POWER                 MinimizePower::compute               Costs/src/MinimizePower.cpp:12-71
 *   RAPTOR_COST_MINIMIZE_TORQUE                MinimizeTorque::compute              Costs/src/MinimizeTorque.cpp:12-63
 *   RAPTOR_COST_MINIMIZE_INITIAL_VELOCITY      MinimizeInitialVelocity::compute     Costs/src/MinimizeInitialVelocity.cpp:10-41
 *   RAPTOR_COST_MINIMIZE_INITIAL_ACCELERATION  MinimizeInitialAcceleration::compute (same form on q_dd(0))
 * F [B][num_costs] = the members `f`, GRAD [B][num_costs][n] = `grad_f` (NULL: compute_derivatives = false).  The weighted sum
 * of Optimizer::eval_f / eval_grad_f (Optimization/src/Optimizer.cpp:201-295) stays with the caller.  These four: legged /
 * Digit problems only (the torques are ConstrainedInverseDynamics::tau).
 */
#define RAPTOR_COST_MINIMIZE_POWER 0
#define RAPTOR_COST_MINIMIZE_TORQUE 1
#define RAPTOR_COST_MINIMIZE_INITIAL_VELOCITY 2
#define RAPTOR_COST_MINIMIZE_INITIAL_ACCELERATION 3
/* the costs of KinovaLongerHorizonOptimizer (KinovaLongerHorizonOptimizer.cpp:114-119), trajectory only, any problem kind;
 * not to be mixed with the four above in one call:
 *   RAPTOR_COST_MINIMIZE_PATH_LENGTH  MinimizePathLength::compute  Costs/src/MinimizePathLength.cpp:10-61
 *   RAPTOR_COST_MINIMIZE_JERK         MinimizeJerk::compute        Costs/src/MinimizeJerk.cpp:10-61      */
#define RAPTOR_COST_MINIMIZE_PATH_LENGTH 4
#define RAPTOR_COST_MINIMIZE_JERK 5
int raptor_eval_costs(raptor_problem* problem, const double* d_Z, int B, int num_costs, const int* kinds, double* d_F,
                      double* d_GRAD, void* stream);
int raptor_eval_costs_host(raptor_problem* problem, const double* Z, int B, int num_costs, const int* kinds, double* F,
                           double* GRAD);
/* Costs::hess_f of MinimizePathLength / MinimizeJerk (MinimizePathLength.cpp:43-52, MinimizeJerk.cpp:43-52) over
 * PiecewiseBezierCurves: independent of z (the costs are quadratic), upper triangle row by row, [n (n + 1) / 2].  What
 * Optimizer::eval_hess_f sums with costsWeightVec_ (Optimizer.cpp:298-337) before eval_h scales it by obj_factor. */
int raptor_cost_hessian_host(raptor_problem* problem, int kind, double* HESS);

/*
 * KinovaOptimizer's objective (Examples/Kinova/CollisionAvoidanceTrajectory/src/KinovaOptimizer.cpp:152-250):
 *   f = 0.5 * sum_j d_j^2, d_j = q(tplan_n)(j) - q_des(j), wrapped to (-pi, pi] on the continuous joints 0, 2, 4, 6;
 *   grad_f(j) = d_j * pq_pz(tplan_n)(j, j);  hess_f = diag(pq_pz(tplan_n)(j, j)^2) (enters raptor_eval_h_batch).
 * _set_objective stands in for the q_des_input / tplan_n_input arguments of set_parameters (:28-29, 39-40).
 *   Z [B][7] -> F [B], GRAD [B][7] (NULL: eval_f only)
 */
int raptor_kinova_set_objective(raptor_problem* problem, const double* q_des, int tplan_n);
int raptor_kinova_eval_objective(raptor_problem* problem, const double* d_Z, int B, double* d_F, double* d_GRAD, void* hip_stream);
int raptor_kinova_eval_objective_host(raptor_problem* problem, const double* Z, int B, double* F, double* GRAD);

/*
 * Optimizer::eval_h (Optimization/src/Optimizer.cpp:499-635) for the optimizer that enables it, KinovaOptimizer
 * (KinovaOptimizer.cpp:37): the Hessian of the Lagrangian
 *     obj_factor * hess_f + sum_r lambda_r * pg_pz_pz(r)
 * over the rows of every constraint block -- Constraints::compute(z, true, true) of JointLimits / VelocityLimits (zero),
 * TorqueLimits (InverseDynamics::ptau_pz_pz, InverseDynamics.cpp:141-171) and KinovaCustomizedConstraints
 * (KinovaCustomizedConstraints.cpp:117-155, BoxCollisionAvoidance.cpp:375-475) -- as `values`: the upper triangle row by row,
 * iRow = i, jCol = j >= i (:531-538, :621-627).
 *   Z [B][n], obj_factor [B] (NULL: 0 for every z, eval_hess_f is skipped as at :552), lambda [B][m] -> H [B][n (n + 1) / 2]
 * KinovaLongerHorizonOptimizer problems (PIECEWISE_BEZIER) are evaluated too: obj_factor must be NULL there (that optimizer's
 * objective is a weighted sum of Costs: the caller adds obj_factor * sum_k weight_k * raptor_cost_hessian_host(k)).
 * Other problem kinds return RAPTOR_ERR_RUNTIME (the reference's eval_h returns false for them, :513-515).
 */
int raptor_eval_h_batch(raptor_problem* problem, const double* d_Z, int B, const double* d_obj_factor, const double* d_lambda,
                        double* d_H, void* hip_stream);
int raptor_eval_h_batch_host(raptor_problem* problem, const double* Z, int B, const double* obj_factor, const double* lambda,
                             double* H);

/*
 * Allocation-free variant of raptor_eval_h_batch.  It has NO counterpart in the reference: it exists for callers that cannot
 * allocate device memory while they call -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same arguments, same kernels and bit-identical results as raptor_eval_h_batch in its small-batch form; the partial-Hessian
 * buffer that entry point allocates on demand is here a fixed workspace inside the library (172 KB per GPU, held until the
 * process ends; its per-GPU address is looked up in raptor_init or on the first call on a device).
 * Limits: KinovaOptimizer (ARMOUR_BEZIER) problems only; B <= RAPTOR_H_NOALLOC_MAX_B, larger batches return
 * RAPTOR_ERR_INVALID_ARGUMENT.  All calls on one GPU share that workspace: calls that overlap in time on one GPU (different
 * streams or threads) race on it; calls ordered on one stream do not.  Nothing in it has to be cleared between calls.
 */
#define RAPTOR_H_NOALLOC_MAX_B 112
int raptor_eval_h_batch_noalloc(raptor_problem* problem, const double* d_Z, int B, const double* d_obj_factor,
                                const double* d_lambda, double* d_H, void* hip_stream);

/* Measurement aid with no counterpart in the reference: `blocks` workgroups each fill one contiguous chunk of
 * `chunk_doubles` doubles with 16-byte stores -- the write pattern of the Jacobian kernels, used by scripts/bw_probe.py
 * to calibrate the achievable HBM write rate next to the roofline numbers. */
int raptor_probe_fill(double* d_p, long long blocks, long long chunk_doubles, void* stream);

/* Device selection for the calling process (one process per GPU): hipSetDevice + probe. */
int raptor_init(int device);

#ifdef __cplusplus
}
#endif
#endif /* RAPTOR_HIP_H */
