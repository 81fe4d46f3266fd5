// Digit-v3 closed-loop constants (Examples/Digit/include/DigitDynamicsConstraints.h:19-57, 178-198 and
// DigitDynamicsConstraints_clean.cpp:40-57): which joints close which loop, and the rod / anchor end points.
// Shared by the HIP library (host set-up) and the CPU oracle.
#ifndef RAPTOR_DIGIT_CONSTANTS_H
#define RAPTOR_DIGIT_CONSTANTS_H

#include <cmath>

// role order of raptor_digit_desc::leg_ids[leg][...]
enum DigitLegJoint {
    DG_HIP_ROLL = 0, DG_HIP_YAW, DG_HIP_PITCH, DG_ACHILLES_ROD, DG_ACH2, DG_KNEE, DG_TARSUS, DG_TOE_A, DG_TOE_A_ROD,
    DG_A2, DG_TOE_B, DG_TOE_B_ROD, DG_B2, DG_TOE_PITCH, DG_TOE_ROLL, DG_LEG_JOINTS
};
// dependentJointNames / independentJointNames order within one leg (DigitDynamicsConstraints.h:19-57)
static const int DIGIT_LEG_DEPENDENT[9] = {DG_ACHILLES_ROD, DG_ACH2, DG_TARSUS, DG_TOE_A_ROD, DG_A2, DG_TOE_B_ROD, DG_B2, DG_TOE_PITCH, DG_TOE_ROLL};
static const int DIGIT_LEG_INDEPENDENT[6] = {DG_HIP_ROLL, DG_HIP_YAW, DG_HIP_PITCH, DG_KNEE, DG_TOE_A, DG_TOE_B};

struct DigitLoop {
    int start, endA, endB;      // roles: FK(start -> endA) * rod end  ==  FK(start -> endB) * anchor
    double rod[3];              // rod end point in endA's frame
    double anchor[2][3];        // anchor point in endB's frame, [leg: 0 left, 1 right]
    int ndrv, drv[2];           // driving (independent) joints
    int ndep, depj[6];          // joints the loop determines (shared between the two toe loops)
};

// knee anchor: Transform(deg2rad(rpy), p1) * Transform(p2) -> p1 + Rx(r) Ry(p) Rz(y) p2   (Transform.cpp:70-77)
static inline void digit_knee_anchor(int leg, double out[3]) {
    const double sg = leg == 0 ? 1.0 : -1.0;
    const double d2r = M_PI / 180.0;
    const double r = sg * 4.47 * d2r, p = 0.32 * d2r, y = sg * 155.8 * d2r;
    const double p1[3] = {-0.01766, -sg * 0.029456, 0.00104}, p2[3] = {0.113789, -sg * 0.011056, 0.0};
    const double cr = std::cos(r), sr = std::sin(r), cp = std::cos(p), sp = std::sin(p), cy = std::cos(y), sy = std::sin(y);
    // R = Rx Ry Rz
    const double R[3][3] = {{cp * cy, -cp * sy, sp},
                            {cr * sy + sr * sp * cy, cr * cy - sr * sp * sy, -sr * cp},
                            {sr * sy - cr * sp * cy, sr * cy + cr * sp * sy, cr * cp}};
    for (int i = 0; i < 3; i++) out[i] = p1[i] + R[i][0] * p2[0] + R[i][1] * p2[1] + R[i][2] * p2[2];
}

// toe-A, toe-B, knee loops of one leg.  The knee anchor is filled by digit_knee_anchor().
static const DigitLoop DIGIT_LOOPS[3] = {
    {DG_TARSUS, DG_A2, DG_TOE_ROLL, {0.17 * 2, 0, 0}, {{0.0179, -0.009551, -0.054164}, {0.0179, 0.009551, -0.054164}},
     2, {DG_TOE_A, DG_TOE_B}, 6, {DG_TOE_A_ROD, DG_A2, DG_TOE_B_ROD, DG_B2, DG_TOE_PITCH, DG_TOE_ROLL}},
    {DG_TARSUS, DG_B2, DG_TOE_ROLL, {0.144 * 2, 0, 0}, {{-0.0181, -0.009551, -0.054164}, {-0.0181, 0.009551, -0.054164}},
     2, {DG_TOE_A, DG_TOE_B}, 6, {DG_TOE_A_ROD, DG_A2, DG_TOE_B_ROD, DG_B2, DG_TOE_PITCH, DG_TOE_ROLL}},
    {DG_HIP_PITCH, DG_ACH2, DG_TARSUS, {0.25 * 2, 0, 0}, {{0, 0, 0}, {0, 0, 0}},
     1, {DG_KNEE, -1}, 3, {DG_ACHILLES_ROD, DG_ACH2, DG_TARSUS, -1, -1, -1}},
};

// stance-foot end transform (DigitDynamicsConstraints_clean.cpp:40-57), [leg: 0 left, 1 right]
static inline void digit_stance_foot_endT(int leg, double R[9], double p[3]) {
    const double s3 = std::sin(M_PI / 3);
    if (leg == 0) {
        const double r[9] = {0, 1, 0, -0.5, 0, s3, s3, 0, 0.5};
        for (int i = 0; i < 9; i++) R[i] = r[i];
        p[0] = 0; p[1] = -0.05456; p[2] = -0.0315;
    } else {
        const double r[9] = {0, -1, 0, 0.5, 0, -s3, s3, 0, 0.5};
        for (int i = 0; i < 9; i++) R[i] = r[i];
        p[0] = 0; p[1] = 0.05456; p[2] = -0.0315;
    }
}

#endif
