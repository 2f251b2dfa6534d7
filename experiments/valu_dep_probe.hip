// Dependent-issue latency of VALU instructions for one wavefront alone on its SIMD (gfx950): a chain of N dependent
// instructions, against the same instructions split over 2 / 4 independent chains (64 instructions per loop trip, so
// the loop overhead is < 5 %).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int OP>
__global__ void k(float *out, long *cyc, int n) {
    float f[4]; double d[4];
    for (int i = 0; i < 4; ++i) f[i] = 0.5f + threadIdx.x * 1e-3f + i, d[i] = 0.5 + threadIdx.x * 1e-3 + i;
    const float cf = 1.0000001f, ef = 1e-9f;
    const double cd = 1.0000001, ed = 1e-9;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int c = j % CH;
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(cf), "v"(ef));
            if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(cd), "v"(ed));
            if (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[c]));
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[c]) : "v"(cf));
        }
    }
    long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = f[0] + f[1] + f[2] + f[3] + (float)(d[0] + d[1] + d[2] + d[3]);
}
template <int CH, int OP> double run(float *d, long *c) {
    const int n = 256; long h;
    k<CH, OP><<<256, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    k<CH, OP><<<256, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    return (double)h / n / 64;
}
int main() {
    float *d; long *c;
    (void)hipMalloc(&d, 256 * 64 * 4); (void)hipMalloc(&c, 32);
    printf("cycles per instruction, 1 / 2 / 4 independent chains\n");
    printf("v_fma_f32      %.2f %.2f %.2f\n", run<1, 0>(d, c), run<2, 0>(d, c), run<4, 0>(d, c));
    printf("v_fma_f64      %.2f %.2f %.2f\n", run<1, 1>(d, c), run<2, 1>(d, c), run<4, 1>(d, c));
    printf("v_sqrt_f32     %.2f %.2f %.2f\n", run<1, 2>(d, c), run<2, 2>(d, c), run<4, 2>(d, c));
    printf("v_rcp_f32      %.2f %.2f %.2f\n", run<1, 3>(d, c), run<2, 3>(d, c), run<4, 3>(d, c));
    printf("v_cndmask_b32  %.2f %.2f %.2f\n", run<1, 4>(d, c), run<2, 4>(d, c), run<4, 4>(d, c));
    return 0;
}
