// Issue cost (cycles per instruction, one wavefront alone on its SIMD, gfx950) of the VALU instructions the quad
// kernels are made of: independent streams of 8 accumulators, 1024 x 8 instructions each, s_memtime around them.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int V>
__global__ void k(double *out, long *cyc, int n) {
    double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) d[i] = 1.0 + threadIdx.x * 1e-3 + i, f[i] = 0.5f + threadIdx.x * 1e-3f + i;
    const double c = 1.0000001, e = 1e-9;
    const float cf = 1.0000001f, ef = 1e-9f;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(c), "v"(e));
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(ef));
#define CVT6432(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define CVT3264(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 7]));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(ef), "v"(cf));
#define ACCW(i) asm volatile("v_accvgpr_write_b32 a%c1, %0" : : "v"(f[i]), "n"(i));
#define ACCR(i) asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(f[i]) : "n"(i));
        if (V == 0) { REP8(FMA64) }
        if (V == 1) { REP8(FMA32) }
        if (V == 2) { REP8(CVT6432) }
        if (V == 3) { REP8(CVT3264) }
        if (V == 4) { REP8(MOV64) }
        if (V == 5) { REP8(ADD64) }
        if (V == 6) { REP8(MUL64) }
        if (V == 7) { REP8(MED3) }
        if (V == 8) { REP8(ACCW) }
        if (V == 9) { REP8(ACCW) REP8(ACCR) }
    }
    long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += d[i] + f[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int V> void run(const char *name, double *d, long *c, int per) {
    const int n = 1024; long h;
    k<V><<<256, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    k<V><<<256, 64>>>(d, c, n); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %.2f cycles per instruction\n", name, (double)h / n / per);
}
int main() {
    double *d; long *c;
    (void)hipMalloc(&d, 256 * 64 * 8); (void)hipMalloc(&c, 32);
    run<0>("v_fma_f64", d, c, 8); run<1>("v_fma_f32", d, c, 8); run<2>("v_cvt_f32_f64", d, c, 8); run<3>("v_cvt_f64_f32", d, c, 8);
    run<4>("v_mov_b64", d, c, 8); run<5>("v_add_f64", d, c, 8); run<6>("v_mul_f64", d, c, 8); run<7>("v_med3_f32", d, c, 8);
    run<8>("v_accvgpr_write_b32", d, c, 8); run<9>("v_accvgpr_write+read", d, c, 16);
    return 0;
}
