// What one wavefront alone on its SIMD sustains in fp64 (gfx950), and what the chip clocks at while every SIMD does it:
// v_fma_f64 in CH independent dependency chains (1..8), SGPR or VGPR coefficient, and mixes with conversions / fp32 — in
// core cycles per instruction (s_memtime), the sustained core clock (s_memtime over s_memrealtime, 100 MHz) and wall-clock
// nanoseconds per instruction (HIP events), for a grid of 64 workgroups (a quarter of the CUs) and of 256 x 4 wavefronts
// (one per SIMD, the whole chip: the headline launch's shape) and 256 x 8 (two per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int CH, int OP>
__global__ __launch_bounds__(256) void k(float *out, long *cyc, int n, double cs) {
    double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) d[i] = 0.5 + threadIdx.x * 1e-3 + i, f[i] = 0.25f + i;
    const double cd = 1.0000001, ed = 1e-9;
    const float cf = 1.0000001f, ef = 1e-9f;
    long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 96; ++j) {
            const int c = j % CH;
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(cd), "v"(ed));
            if (OP == 1) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[c]) : "s"(cs), "v"(ed));          // SGPR coefficient
            if (OP == 2) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[c]) : "s"(cs), "v"(ed));              // accumulate only
            if (OP == 3) {   // the lean kernel's mix: 6 fp64 FMA per (cvt + fp32)
                if (j % 8 == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(d[c]));
                else if (j % 8 == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(ef));
                else asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(d[c]) : "s"(cs), "v"(ed));
            }
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(cf), "v"(ef));
            if (OP == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(cd));
            if (OP == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(ed));
        }
    }
    long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x % 64 == 0) {
        cyc[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0;
        cyc[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = r1 - r0;
    }
    double s = 0; float sf = 0;
    for (int i = 0; i < 8; ++i) s += d[i], sf += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s + sf;
}
template <int CH, int OP> void run(const char *name, float *d, long *c, int grid) {
    const int n = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<CH, OP><<<grid, 256>>>(d, c, n, 1.0000001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) k<CH, OP><<<grid, 256>>>(d, c, n, 1.0000001);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(2 * grid * 4);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cpi, clk;
    for (int i = 0; i < grid * 4; ++i) cpi.push_back((double)h[2 * i] / n / 96), clk.push_back((double)h[2 * i] / h[2 * i + 1] * 0.1);
    std::sort(cpi.begin(), cpi.end()); std::sort(clk.begin(), clk.end());
    const int waves_per_simd = grid <= 256 ? 1 : grid / 256;
    printf("%-26s chains %d grid %4d: %.2f cycles/instr (median wave)  clock %.2f GHz  wall %.3f ns/instr/wave\n", name, CH, grid,
           cpi[cpi.size() / 2], clk[clk.size() / 2], ms / 5 * 1e6 / n / 96 / waves_per_simd);
    fflush(stdout);
}
int main() {
    float *d; long *c;
    (void)hipMalloc(&d, 2048 * 256 * 4); (void)hipMalloc(&c, 2 * 2048 * 4 * 8);
    for (int grid : {64, 256, 512}) {
        run<1, 0>("v_fma_f64 vgpr coef", d, c, grid); run<2, 0>("v_fma_f64 vgpr coef", d, c, grid); run<3, 0>("v_fma_f64 vgpr coef", d, c, grid);
        run<4, 0>("v_fma_f64 vgpr coef", d, c, grid); run<6, 0>("v_fma_f64 vgpr coef", d, c, grid); run<8, 0>("v_fma_f64 vgpr coef", d, c, grid);
        run<1, 1>("v_fma_f64 sgpr coef", d, c, grid); run<2, 1>("v_fma_f64 sgpr coef", d, c, grid); run<4, 1>("v_fma_f64 sgpr coef", d, c, grid); run<8, 1>("v_fma_f64 sgpr coef", d, c, grid);
        run<4, 2>("v_fmac_f64 sgpr coef", d, c, grid); run<8, 2>("v_fmac_f64 sgpr coef", d, c, grid);
        run<4, 3>("6 fma64 : cvt : add32", d, c, grid); run<8, 3>("6 fma64 : cvt : add32", d, c, grid);
        run<1, 4>("v_fma_f32", d, c, grid); run<4, 4>("v_fma_f32", d, c, grid); run<8, 4>("v_fma_f32", d, c, grid);
        run<4, 5>("v_mul_f64", d, c, grid); run<4, 6>("v_add_f64", d, c, grid);
    }
    return 0;
}
