// Issue cost of fp64 VALU instructions on gfx950, one wavefront alone on a SIMD: cycles per instruction for independent chains of
// v_fma_f64, v_mul_f64, v_add_f64, v_max_f64, v_cndmask (64-bit select = 2), 64-bit DPP move (= 2 v_mov_b32_dpp).
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_probe.hip -o build/probe/valu_rate_probe && build/probe/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 65536
template <int KIND>
__global__ void probe(double* out, long long* cyc, double seed) {
    double x[8];
    for (int k = 0; k < 8; k++) x[k] = seed + k + threadIdx.x;
    const double m = 1.0000001, c = 1e-9;
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) x[k] = __builtin_fma(x[k], m, c);
            else if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(m));
            else if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c));
            else if (KIND == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[k]) : "v"(c));
            else if (KIND == 4) x[k] = __longlong_as_double(__builtin_amdgcn_update_dpp(0, (int)__double_as_longlong(x[k]), 0x111, 0xf, 0xf, true) |
                                                            ((long long)__builtin_amdgcn_update_dpp(0, (int)(__double_as_longlong(x[k]) >> 32), 0x111, 0xf, 0xf, true) << 32));
            else if (KIND == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(c));
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int waves_per_simd) {
    double* out; long long* cyc;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd;  // one workgroup per CU: waves_per_simd wavefronts on each SIMD
    hipMalloc(&out, blocks * threads * sizeof(double)); hipMalloc(&cyc, blocks * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    long long h; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // every SIMD runs waves_per_simd wavefronts of ITER * 8 instructions: SIMD-cycles per instruction at 2.4 GHz from the wall time
    const double per_simd = (double)ITER * 8.0 * waves_per_simd;
    printf("%-24s %d wave(s)/SIMD: %.2f clock64 ticks per instruction per wave; %.3f ms -> %.2f cycles (at 2.4 GHz) per instruction per SIMD\n", name,
           waves_per_simd, (double)h / (ITER * 8.0), ms, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 6; w++) {
        run<0>("fma (compiler)", w); run<5>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_max_f64", w);
        run<4>("64-bit dpp move (2 ops)", w);
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    printf("clock rate %d kHz, wall clock rate %d kHz (clock64 counts the latter on gfx9: scale by the ratio)\n", clk, wall);
    return 0;
}
