// Issue cost of fp64 instructions on gfx950: SIMD-cycles per wave-instruction for independent chains of v_fma_f64, v_mul_f64,
// v_add_f64, v_max_f64, a 64-bit DPP move (= 2 v_mov_b32_dpp) and v_mfma_f64_16x16x4_f64 (32-bit instructions: valu32_rate_probe.hip), with 1..4 wavefronts
// resident per SIMD (one workgroup per CU).
// The clock is MEASURED, not assumed: every launch brackets its loop with clock64() (s_memtime) and wall_clock64() (s_memrealtime,
// a constant 100 MHz counter: hipDeviceAttributeWallClockRate), so ticks-per-instruction comes with the tick rate of clock64 in
// that very launch; the wall time of the launch (HIP events) gives SIMD-seconds per instruction independently.  A fully
// pipelined fp64 VALU instruction of one wavefront alone is 8 shader cycles by construction (64 lanes, 16 lanes per cycle, half
// rate), which calibrates the shader clock from the 1-wave rows: sclk = 8 / (seconds per instruction).
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate_probe.hip -o build/probe/valu_rate_probe && build/probe/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define ITER 65536
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void probe(double* out, long long* cyc, double seed) {
    double x[8];
    for (int k = 0; k < 8; k++) x[k] = seed + k + threadIdx.x;
    v4d acc[4];
    for (int k = 0; k < 4; k++) acc[k] = v4d{seed, seed + 1, seed + 2, seed + 3};
    const double m = 1.0000001, c = 1e-9;
    int sel[8];
    for (int k = 0; k < 8; k++) sel[k] = (int)threadIdx.x + k;
    const long long w0 = wall_clock64();
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
            else if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(sel[k]) : "v"(it));
            else if (KIND == 7) acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[k], m, acc[k & 3], 0, 0, 0);  // four independent accumulators
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    double s = 0.0;
    for (int k = 0; k < 8; k++) s += (double)sel[k];
    for (int k = 0; k < 8; k++) s += x[k];
    for (int k = 0; k < 4; k++) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}
static int g_wall_khz = 100000;
template <int KIND>
void run(const char* name, int waves_per_simd) {
    double* out; long long* cyc;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd;  // one workgroup per CU: waves_per_simd wavefronts on each SIMD
    if (threads > 1024) { std::fprintf(stderr, "%d lanes per workgroup cannot be launched\n", threads); std::exit(1); }
    CHECK(hipMalloc(&out, blocks * threads * sizeof(double))); CHECK(hipMalloc(&cyc, 2 * blocks * sizeof(long long)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0);
        CHECK(hipGetLastError());  // a launch that fails must not be recorded as a 0.004 ms data point
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    long long h[2]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    const double n_inst = (double)ITER * 8.0;
    const double tick_hz = (double)h[0] / ((double)h[1] / (g_wall_khz * 1e3));  // clock64 ticks per second in this launch
    const double sec_per_inst_simd = ms * 1e-3 / (n_inst * waves_per_simd);    // SIMD-seconds per wave-instruction (wall time)
    printf("%-26s %d wave(s)/SIMD: %6.2f clock64 ticks/instr/wave, clock64 at %.3f GHz (vs s_memrealtime); %.3f ms -> %.3f ns per instruction per SIMD\n",
           name, waves_per_simd, (double)h[0] / n_inst, tick_hz * 1e-9, ms, sec_per_inst_simd * 1e9);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}
int main() {
    int clk = 0, wall = 0;
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    CHECK(hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0));
    if (wall > 0) g_wall_khz = wall;
    printf("reported clock rate %d kHz, wall clock (s_memrealtime) rate %d kHz\n", clk, g_wall_khz);
    for (int w = 1; w <= 4; w++) {
        run<0>("fma (compiler)", w); run<5>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_max_f64", w);
        run<4>("64-bit dpp move (2 ops)", w); run<7>("v_mfma_f64_16x16x4_f64", w);
    }
    printf("shader clock from the 1-wave v_fma_f64 row: 8 cycles / (ns per instruction) GHz; an fp64 wave-instruction at w waves/SIMD then costs\n"
           "(ns per instruction per SIMD) x that clock cycles.  One v_mfma_f64_16x16x4_f64 = 2048 flops per wave-instruction (16*16*4*2).\n");
    return 0;
}
