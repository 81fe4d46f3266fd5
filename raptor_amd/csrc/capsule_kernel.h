// Tapered-capsule pair distance + analytic gradient, one lane per capsule pair.
// Follows TaperedCapsuleCollision<F>::distanceInternal (Constraints/src/TaperedCapsuleCollision.cpp:43-274)
// branch for branch, including its quirks:
//   - the gradient is always computed when Jacobians are given (:162);
//   - t_test[8], t_test[9] use the UN-rooted radicand in one term (:131-134);
//   - for the edge-edge cases 8/9 pu_pz is set with the same expression as pt_pz (:257-258).
// The per-column gradient terms are independent given the per-pair scalars, so columns are streamed
// instead of materialising the reference's VecF temporaries.  Two phases per 64-pair wavefront: lane = pair for the 14
// candidates and the per-pair scalars (left in LDS), then the wave walks the (pair, column) entries of its 64 pairs in flat
// order -- consecutive lanes read consecutive columns of dP and write consecutive entries of the gradient (with a lane per pair
// every load touched 64 cache lines 96 F bytes apart: fine at F = 14, eight-fold L2 traffic at F = 112).
#ifndef RAPTOR_CAPSULE_KERNEL_H
#define RAPTOR_CAPSULE_KERNEL_H

#include "spatial_dev.h"

// TaperedCapsuleCollision.cpp:7-22
RD void solve_quadratic_dev(double a, double b, double c, double* s1, double* s2) {
    const double disc = (b * b) - (4 * a * c);
    if (disc < 0) {
        *s1 = -1.0;
        *s2 = -1.0;
        return;
    }
    const double sq = sqrt(disc);
    const double twoa = 2 * a;
    *s1 = (-b + sq) / twoa;
    *s2 = (-b - sq) / twoa;
}

#define CAPS_BLOCK 64
#define CAPS_REC 34  // per-pair scalars of the gradient phase, in doubles
__global__ __launch_bounds__(CAPS_BLOCK) void capsule_kernel(const double* __restrict__ P, const double* __restrict__ dP,
                                                             const double* __restrict__ Rr, const int n_pairs, const int F,
                                                             double* __restrict__ dist_out, double* __restrict__ grad_out) {
    __shared__ double prec[CAPS_BLOCK * CAPS_REC];
    const int pair0 = blockIdx.x * CAPS_BLOCK;
    const int idx = pair0 + threadIdx.x;
    const bool want_grad = dP != nullptr && grad_out != nullptr && F > 0;
    if (idx < n_pairs) {
    const double* p = P + (size_t)idx * 12;
    const V3 p11 = v3(p[0], p[1], p[2]), p12 = v3(p[3], p[4], p[5]);
    const V3 p21 = v3(p[6], p[7], p[8]), p22 = v3(p[9], p[10], p[11]);
    const double r11 = Rr[idx * 4], r12 = Rr[idx * 4 + 1], r21 = Rr[idx * 4 + 2], r22 = Rr[idx * 4 + 3];

    const double r1 = r12 - r11;
    const double r2 = r22 - r21;
    const V3 d1 = p12 - p11, d2 = p22 - p21, d12 = p21 - p11;
    const double a = dot(d1, d1), b = dot(d1, d2), c = dot(d1, d12);
    const double e = dot(d2, d2), f = dot(d2, d12), g = dot(d12, d12);

    double u_test[14], t_test[14];
    u_test[10] = 0.0; u_test[11] = 0.0; u_test[12] = 1.0; u_test[13] = 1.0;
    t_test[10] = 0.0; t_test[11] = 1.0; t_test[12] = 0.0; t_test[13] = 1.0;
    if (fabs(r2) > 1e-10) {
        solve_quadratic_dev(e * (e - r2 * r2), 2 * f * (e - r2 * r2), f * f - r2 * r2 * g, &u_test[0], &u_test[1]);
        solve_quadratic_dev(e * (e - r2 * r2), 2 * (f - b) * (e - r2 * r2), (f - b) * (f - b) - r2 * r2 * (g - 2 * c + a), &u_test[2], &u_test[3]);
    } else {
        u_test[0] = -f * (e - r2 * r2) / (e * (e - r2 * r2));
        u_test[1] = -1;
        u_test[2] = -(f - b) * (e - r2 * r2) / (e * (e - r2 * r2));
        u_test[3] = -1;
    }
    t_test[0] = 0; t_test[1] = 0; t_test[2] = 1; t_test[3] = 1;
    u_test[4] = 0; u_test[5] = 0; u_test[6] = 1; u_test[7] = 1;
    if (fabs(r1) > 1e-10) {
        solve_quadratic_dev(a * (a - r1 * r1), -2 * c * (a - r1 * r1), c * c - r1 * r1 * g, &t_test[4], &t_test[5]);
        solve_quadratic_dev(a * (a - r1 * r1), -2 * (b + c) * (a - r1 * r1), (b + c) * (b + c) - r1 * r1 * (e + 2 * f + g), &t_test[6], &t_test[7]);
    } else {
        t_test[4] = c * (a - r1 * r1) / (a * (a - r1 * r1));
        t_test[5] = -1;
        t_test[6] = (b + c) * (a - r1 * r1) / (a * (a - r1 * r1));
        t_test[7] = -1;
    }
    const double rad2 = (g * b * b - 2 * b * c * f + e * c * c + a * f * f - a * e * g) * (b * b + 2 * b * r1 * r2 + e * r1 * r1 + a * r2 * r2 - a * e);
    if (rad2 < 0) {
        t_test[8] = -1.0; t_test[9] = -1.0; u_test[8] = -1.0; u_test[9] = -1.0;
    } else {
        const double sr = sqrt(rad2);
        const double den = (a * a * e * e - a * a * e * r2 * r2 - 2 * a * b * b * e + a * b * b * r2 * r2 - 2 * a * b * e * r1 * r2 - a * e * e * r1 * r1 + b * b * b * b + 2 * b * b * b * r1 * r2 + b * b * e * r1 * r1);
        t_test[8] = (c * r2 + f * r1) / (a * r2 + b * r1) + ((b * r2 + e * r1) * (a * r2 * sr - b * b * b * c + b * r1 * rad2
                    + a * a * f * r2 * r2 + a * b * b * f - a * a * e * f - a * b * c * r2 * r2 - b * c * e * r1 * r1 + a * e * f * r1 * r1 - 2 * b * b * c * r1 * r2 + a * b * c * e + 2 * a * b * f * r1 * r2)) / ((a * r2 + b * r1) * den);
        t_test[9] = (c * r2 + f * r1) / (a * r2 + b * r1) - ((b * r2 + e * r1) * (b * b * b * c + a * r2 * sr + b * r1 * rad2
                    - a * a * f * r2 * r2 - a * b * b * f + a * a * e * f + a * b * c * r2 * r2 + b * c * e * r1 * r1 - a * e * f * r1 * r1 + 2 * b * b * c * r1 * r2 - a * b * c * e - 2 * a * b * f * r1 * r2)) / ((a * r2 + b * r1) * den);
        u_test[8] = (a * r2 * sr - b * b * b * c + b * r1 * sr
                     + a * a * f * r2 * r2 + a * b * b * f - a * a * e * f - a * b * c * r2 * r2 - b * c * e * r1 * r1 + a * e * f * r1 * r1 - 2 * b * b * c * r1 * r2 + a * b * c * e + 2 * a * b * f * r1 * r2) / den;
        u_test[9] = -(b * b * b * c + a * r2 * sr + b * r1 * sr
                      - a * a * f * r2 * r2 - a * b * b * f + a * a * e * f + a * b * c * r2 * r2 + b * c * e * r1 * r1 - a * e * f * r1 * r1 + 2 * b * b * c * r1 * r2 - a * b * c * e - 2 * a * b * f * r1 * r2) / den;
    }

    // first minimum over the admissible candidates (:141-156)
    int ind = 0;
    double best = INFINITY, uStar = u_test[0], tStar = t_test[0];
#pragma unroll
    for (int k = 0; k < 14; k++) {
        double dk = INFINITY;
        if (u_test[k] <= 1 && u_test[k] >= 0 && t_test[k] <= 1 && t_test[k] >= 0) {
            const V3 dv = d1 * t_test[k] - d2 * u_test[k] - d12;
            dk = sqrt(dot(dv, dv)) - r1 * t_test[k] - r2 * u_test[k];
        }
        if (k == 0 || dk < best) {
            best = dk;
            ind = k;
            uStar = u_test[k];
            tStar = t_test[k];
        }
    }
    dist_out[idx] = best - r11 - r21;
    if (want_grad) {
        // ---- per-pair scalars of the gradient (:162-271): norm = |dist_vec|, slope_t / slope_u = half the derivative of the squared
        // separation along t / u at the optimum ----
        const V3 dist_vec = d1 * tStar - d2 * uStar - d12;
        double norm = sqrt(dot(dist_vec, dist_vec));
        if (norm == 0) norm = 1e-6;
        const double slope_t = a * tStar - b * uStar - c;
        const double slope_u = -b * tStar - e * uStar + f;
        const double gradient_factor = 1 / (norm * norm * norm);
        double* o = prec + threadIdx.x * CAPS_REC;
        o[0] = d1.x; o[1] = d1.y; o[2] = d1.z; o[3] = d2.x; o[4] = d2.y; o[5] = d2.z; o[6] = d12.x; o[7] = d12.y; o[8] = d12.z;
        o[9] = tStar; o[10] = uStar; o[11] = norm; o[12] = slope_t; o[13] = slope_u; o[14] = gradient_factor;
        o[15] = a / norm - gradient_factor * slope_t * slope_t;   // L11
        o[16] = e / norm - gradient_factor * slope_u * slope_u;     // L22
        o[17] = -b / norm - gradient_factor * slope_t * slope_u;   // L12
        o[18] = r1; o[19] = r2; o[20] = (double)ind;
    }
    }
    if (!want_grad) return;
    __syncthreads();

    // ---- gradient, flat over the (pair, column) entries of this wavefront's pairs ----
    const int npw = (n_pairs - pair0 < CAPS_BLOCK) ? n_pairs - pair0 : CAPS_BLOCK;
    const int total = npw * F;
    const float invF = 1.0f / (float)F;
    for (int t = threadIdx.x; t < total; t += CAPS_BLOCK) {
        int pl = (int)(((float)t + 0.5f) * invF);  // t / F: exact here (t < 2^16 F-blocks of at most 64 pairs)
        if (pl * F > t) pl--;
        if ((pl + 1) * F <= t) pl++;
        const int i = t - pl * F;
        const double* o = prec + pl * CAPS_REC;
        const V3 d1 = v3(o[0], o[1], o[2]), d2 = v3(o[3], o[4], o[5]), d12 = v3(o[6], o[7], o[8]);
        const double tStar = o[9], uStar = o[10], norm = o[11], slope_t = o[12], slope_u = o[13], gradient_factor = o[14];
        const double L11 = o[15], L22 = o[16], L12 = o[17], r1 = o[18], r2 = o[19];
        const int ind = (int)o[20];
        const V3 dist_vec = d1 * tStar - d2 * uStar - d12;
        const V3 dnorm_vec = d1 * tStar * tStar - d2 * tStar * uStar - d12 * tStar;
        const V3 d1md2md12 = d1 - d2 - d12;
        const V3 hvec = d1 * -tStar + d2 * uStar + d12;
        const double* q11 = dP + (size_t)(pair0 + pl) * 12 * F;
        const double* q12 = q11 + 3 * F;
        const double* q21 = q11 + 6 * F;
        const double* q22 = q11 + 9 * F;
        const V3 pd1 = v3(q12[i] - q11[i], q12[F + i] - q11[F + i], q12[2 * F + i] - q11[2 * F + i]);
        const V3 pd2 = v3(q22[i] - q21[i], q22[F + i] - q21[F + i], q22[2 * F + i] - q21[2 * F + i]);
        const V3 pd12 = v3(q21[i] - q11[i], q21[F + i] - q11[F + i], q21[2 * F + i] - q11[2 * F + i]);
        const double pa = 2 * dot(d1, pd1);
        const double pb = dot(d1, pd2) + dot(d2, pd1);
        const double pc = dot(d12, pd1) + dot(d1, pd12);
        const double pe = 2 * dot(d2, pd2);
        const double pf = dot(d2, pd12) + dot(d12, pd2);
        const double dslope_t = pa * tStar - pb * uStar - pc;
        const double dslope_u = -pb * tStar + pe * uStar + pf;
        const double dnorm_col = (dot(dnorm_vec, pd1) - dot(dnorm_vec, pd2) - dot(d1md2md12, pd12)) / (norm * norm);
        const double dL1 = (tStar == 1) ? -(dslope_t * norm - slope_t * dnorm_col) : 0.0;
        const double dL2 = (tStar == 0) ? -(dslope_u * norm - slope_u * dnorm_col) : 0.0;
        const double dL3 = (uStar == 1) ? (dslope_t * norm - slope_t * dnorm_col) : 0.0;
        const double dL4 = (uStar == 0) ? (dslope_u * norm - slope_u * dnorm_col) : 0.0;
        const V3 comb = pd1 * -tStar + pd2 * uStar + pd12;
        const double Hin = dot(hvec, comb);
        double pt = 0.0, pu = 0.0;
        if (ind <= 3) {
            const double H2 = dslope_u / norm + dL2 - dL4 - slope_u * Hin * gradient_factor;
            pu = -H2 / L22;
        } else if (ind <= 7) {
            const double H1 = dslope_t / norm + dL1 - dL3 - slope_t * Hin * gradient_factor;
            pt = -H1 / L11;
        } else if (ind <= 9) {
            const double H2 = dslope_u / norm + dL2 - dL4 - slope_u * Hin * gradient_factor;
            const double H1 = dslope_t / norm + dL1 - dL3 - slope_t * Hin * gradient_factor;
            pt = (L12 * H2 - L22 * H1) / (L11 * L22 - L12 * L12);
            pu = (L12 * H2 - L22 * H1) / (L11 * L22 - L12 * L12);
        }
        const V3 colv = pd1 * tStar + d1 * pt - pd2 * uStar - d2 * pu - pd12;
        grad_out[(size_t)pair0 * F + t] = dot(dist_vec, colv) / norm - r1 * pt - r2 * pu;
    }
}

#endif
