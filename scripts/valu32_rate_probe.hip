// Issue cost of 32-bit VALU instructions on gfx950 (companion of valu_rate_probe.hip): eight independent registers per lane, one
// instruction kind per kernel, 1 / 2 / 4 wavefronts per SIMD; SIMD-ns per wave-instruction from the launch's wall time and the
// clock64 tick rate of the same launch.  (A bare `v_cndmask_b32 ..., vcc` with VCC never written in the kernel measures ~23 cycles:
// an artefact of the probe, not of the instruction -- the v_cmp + v_cndmask pair below is the meaningful row, 3.7 cycles each.)
//   hipcc --offload-arch=gfx950 -O3 scripts/valu32_rate_probe.hip -o build/probe/valu32_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define ITER 65536

template <int KIND>
__global__ void probe(int* out, long long* cyc, int seed) {
    int y[8];
    for (int k = 0; k < 8; k++) y[k] = seed + k + (int)threadIdx.x;
    const int a = seed * 3 + (int)threadIdx.x, msk = (threadIdx.x & 1) ? -1 : 0;
    unsigned long long smask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[k]) : "v"(a));
            else if (KIND == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a), "s"(smask));
            else if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 4) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(y[k]) : "v"(msk), "v"(a));
            else if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 7) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 9) asm volatile("v_max_i32 %0, %0, %1" : "+v"(y[k]) : "v"(a));
            else if (KIND == 10) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[k]) : "v"(a) : "vcc");
            else if (KIND == 11) asm volatile("v_cmp_gt_i32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(y[k]) : "v"(a) : "s20", "s21");
            else if (KIND == 12) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(y[k]), "v"(a) : "vcc");
            else if (KIND == 13) y[k] = (y[k] > a + it) ? y[k] - 1 : a;  // what the compiler makes of a select
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    int s = 0;
    for (int k = 0; k < 8; k++) s += y[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int KIND>
void run(const char* name, int waves_per_simd) {
    int* out; long long* cyc;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd;
    CHECK(hipMalloc(&out, blocks * threads * sizeof(int))); CHECK(hipMalloc(&cyc, 2 * blocks * sizeof(long long)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long h[2]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double n_inst = (double)ITER * 8.0;
    const double tick_hz = (double)h[0] / ((double)h[1] / 1e8);
    const double ns = ms * 1e6 / (n_inst * waves_per_simd);
    printf("%-22s %d wave(s)/SIMD: %6.2f ticks/instr/wave, clock64 at %.3f GHz; %.3f ns per instruction per SIMD = %.2f cycles\n", name,
           waves_per_simd, (double)h[0] / n_inst, tick_hz * 1e-9, ns, ns * tick_hz * 1e-9);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}
int main() {
    for (int w : {1, 2, 4}) {
        run<1>("v_cndmask_b32_e64 (sgpr)", w); run<2>("v_add_u32", w); run<3>("v_and_b32", w);
        run<4>("v_bfi_b32", w); run<5>("v_mov_b32", w); run<6>("v_fma_f32", w); run<7>("v_xor_b32", w); run<8>("v_mul_lo_u32", w);
        run<9>("v_max_i32", w);
        run<10>("v_cmp + v_cndmask vcc (2)", w); run<11>("v_cmp + v_cndmask sgpr (2)", w); run<12>("v_cmp_gt_i32 vcc", w); run<13>("compiled select", w);
    }
    return 0;
}
