// World-frame spatial algebra for the gfx950 kernels.
//
// Design note (MI355X-first, not a translation of the reference): the reference's inverse dynamics
// (pinocchio::rnea / Featherstone link-frame recursion, CustomizedInverseDynamics.cpp:133-302) carries
// 6x6 Plucker transforms between link frames.  Here every spatial quantity is expressed in ONE frame
// (world axes, world origin).  The backward sweep then needs no transforms at all -- a joint torque is a
// dot product of the joint's world-frame motion axis with the suffix sum of link wrenches -- and the
// same world-frame joint axes/origins are exactly what the sphere Jacobians of the collision rows
// need.  Each lane owns one (sample, decision-variable column) pair and keeps its state in VGPRs;
// model constants arrive through uniform scalar loads.
#ifndef RAPTOR_SPATIAL_DEV_H
#define RAPTOR_SPATIAL_DEV_H

#include <hip/hip_runtime.h>

#define RD __device__ __forceinline__

struct V3 {
    double x, y, z;
};
RD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
RD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
RD V3 operator*(double s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
RD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// a + s*b
RD V3 axpy(V3 a, double s, V3 b) { return V3{a.x + s * b.x, a.y + s * b.y, a.z + s * b.z}; }

// 3x3 rotation, row-major, columns are the child-frame axes expressed in the parent/world frame
struct M3 {
    double m[9];
};
RD V3 mul(const M3& R, V3 v) {
    return V3{R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
              R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
RD V3 mulT(const M3& R, V3 v) {
    return V3{R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
              R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z};
}
RD M3 mul(const M3& A, const double* B) {  // A * B, B row-major 3x3 (uniform constants)
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B[j] + A.m[3 * i + 1] * B[3 + j] + A.m[3 * i + 2] * B[6 + j];
    return C;
}
RD V3 col(const M3& R, int j) { return V3{R.m[j], R.m[3 + j], R.m[6 + j]}; }
// column j for a j that differs between lanes: blended with 0 / 1 weights, so that R stays in registers (a select between
// array elements is turned back into an indexed load from a stack copy by the compiler)
RD V3 col_sel(const M3& R, int j) {
    const double w0 = (j == 0) ? 1.0 : 0.0, w1 = (j == 1) ? 1.0 : 0.0, w2 = (j == 2) ? 1.0 : 0.0;
    return V3{R.m[0] * w0 + R.m[1] * w1 + R.m[2] * w2, R.m[3] * w0 + R.m[4] * w1 + R.m[5] * w2, R.m[6] * w0 + R.m[7] * w1 + R.m[8] * w2};
}

// spatial motion vector [angular; linear] / force vector [moment; force], world frame at world origin
struct SV {
    V3 a, l;
};
RD SV operator+(SV p, SV q) { return SV{p.a + q.a, p.l + q.l}; }
RD SV operator-(SV p, SV q) { return SV{p.a - q.a, p.l - q.l}; }
RD SV operator*(SV p, double s) { return SV{p.a * s, p.l * s}; }
RD SV sv_zero() { return SV{V3{0, 0, 0}, V3{0, 0, 0}}; }
#ifdef RAPTOR_DOT_TREE
RD double dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }
#else
// one chain of six fused multiply-adds (1 multiply + 5 FMA) instead of two three-term products and an add: the kernels are bound by
// VALU issue, and their 6-vector products are the bulk of the fp64 instructions
RD double dot(SV p, SV q) {
    return __builtin_fma(p.a.x, q.a.x, __builtin_fma(p.a.y, q.a.y, __builtin_fma(p.a.z, q.a.z,
           __builtin_fma(p.l.x, q.l.x, __builtin_fma(p.l.y, q.l.y, p.l.z * q.l.z)))));
}
#endif
// motion x motion   (Spatial.h crm)
RD SV crm(SV v, SV m) { return SV{cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)}; }
// motion x* force   (Spatial.h crf)
RD SV crf(SV v, SV f) { return SV{cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }

// sin and cos of a joint angle together.  Joint angles are bounded (|x| < 2^17 here, in practice a few turns), so the argument
// reduction is three fused multiply-adds against a three-part pi/2 (each product exact inside the fma: the reduced argument is good
// to ~1.5e-16 absolute up to that bound) followed by the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]
// (coefficients of the fdlibm k_sin / k_cos kernels, < 1 ulp there).  ~35 VALU instructions against ~170 for the library's
// full-range sincos; beyond the bound (or non-finite) the library routine is used.
RD void sincos_joint(double x, double* sn, double* cs) {
    if (!(fabs(x) < 131072.0)) {
        sincos(x, sn, cs);
        return;
    }
    const double k = rint(x * 6.36619772367581382433e-01);  // 2 / pi
    double r = fma(-k, 1.57079632679489655800e+00, x);       // pi / 2, leading 53 bits
    r = fma(-k, 6.12323399573676603587e-17, r);              // next 53 bits
    r = fma(-k, -1.49738490485916983327e-33, r);             // and the next
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                         -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                         2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// rigid-body inertia in world axes: mass, COM position c, rotational inertia about the COM (symmetric)
struct Inertia {
    double mass;
    V3 c;
    double xx, xy, xz, yy, yz, zz;
};
RD SV apply(const Inertia& I, SV x) {
    V3 ac = x.l + cross(x.a, I.c);
    V3 l = ac * I.mass;
    V3 n = V3{I.xx * x.a.x + I.xy * x.a.y + I.xz * x.a.z, I.xy * x.a.x + I.yy * x.a.y + I.yz * x.a.z,
              I.xz * x.a.x + I.yz * x.a.y + I.zz * x.a.z} +
           cross(I.c, l);
    return SV{n, l};
}
// R * Ic * R^T for a symmetric Ic given as (xx,xy,xz,yy,yz,zz)
RD void rotate_inertia(const M3& R, const double* Ic, Inertia& out) {
    // T = R * Ic
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double r0 = R.m[3 * i], r1 = R.m[3 * i + 1], r2 = R.m[3 * i + 2];
        T[3 * i] = r0 * Ic[0] + r1 * Ic[1] + r2 * Ic[2];
        T[3 * i + 1] = r0 * Ic[1] + r1 * Ic[3] + r2 * Ic[4];
        T[3 * i + 2] = r0 * Ic[2] + r1 * Ic[4] + r2 * Ic[5];
    }
    out.xx = T[0] * R.m[0] + T[1] * R.m[1] + T[2] * R.m[2];
    out.xy = T[0] * R.m[3] + T[1] * R.m[4] + T[2] * R.m[5];
    out.xz = T[0] * R.m[6] + T[1] * R.m[7] + T[2] * R.m[8];
    out.yy = T[3] * R.m[3] + T[4] * R.m[4] + T[5] * R.m[5];
    out.yz = T[3] * R.m[6] + T[4] * R.m[7] + T[5] * R.m[8];
    out.zz = T[6] * R.m[6] + T[7] * R.m[7] + T[8] * R.m[8];
}

// Joint kinematics in the world frame.  A = R_parent_world * placement rotation, o = joint origin.
// On return R is the child's world rotation, o the child's origin, s the joint's motion axis
// (Transform.cpp:107-159 joint transforms; Spatial.h:28-108 motion subspaces).
RD void joint_world(int jt, double q, double cq, double sq, const M3& A, M3& R, V3& o, SV& s) {
    const int aj = jt < 0 ? -jt : jt;
    const double sg = jt < 0 ? -1.0 : 1.0;
    const double ss = sg * sq;  // sin of the signed angle
    R = A;
    if (aj == 1) {  // about x: columns y,z mix
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double y = A.m[3 * i + 1], z = A.m[3 * i + 2];
            R.m[3 * i + 1] = y * cq + z * ss;
            R.m[3 * i + 2] = z * cq - y * ss;
        }
        V3 ax = col(A, 0) * sg;
        s = SV{ax, cross(o, ax)};
    } else if (aj == 2) {  // about y
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double x = A.m[3 * i], z = A.m[3 * i + 2];
            R.m[3 * i] = x * cq - z * ss;
            R.m[3 * i + 2] = z * cq + x * ss;
        }
        V3 ax = col(A, 1) * sg;
        s = SV{ax, cross(o, ax)};
    } else if (aj == 3) {  // about z
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double x = A.m[3 * i], y = A.m[3 * i + 1];
            R.m[3 * i] = x * cq + y * ss;
            R.m[3 * i + 1] = y * cq - x * ss;
        }
        V3 ax = col(A, 2) * sg;
        s = SV{ax, cross(o, ax)};
    } else {  // prismatic along axis aj-4
        V3 ax = col_sel(A, aj - 4) * sg;
        o = axpy(o, q, ax);
        s = SV{V3{0, 0, 0}, ax};
    }
}

#endif
