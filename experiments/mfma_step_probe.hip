// What one rollout step of the mfmac kernel costs in isolation (gfx950): two products chained on x, the next step's
// product with t issued behind them, result -> fp32 -> LDS hand-over cells + step counter.  Variants: (0) as is,
// (1) without the LDS stores, (2) without the pre-issued t product, (3) with two polling wavefronts in the workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
typedef volatile float __attribute__((address_space(3))) lds_vf;
typedef volatile int __attribute__((address_space(3))) lds_vi;
template <int V>
__global__ void k(double *out, long *cyc, int n, int stride) {
    __shared__ float cells[64 * 64 * 3 + 64];
    __shared__ int step;
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64 * 3 + 64; i += blockDim.x) cells[i] = 1e-3f * i;
    if (threadIdx.x == 0) step = 0;
    __syncthreads();
    if (wave != 0) {
        float acc = 0.f;
        for (int kk = 0; kk < n; ++kk) {
            while (*(lds_vi *)&step < kk + 1) __builtin_amdgcn_s_sleep(1);
            acc += *(lds_vf *)&cells[((kk & 63) * 3 + (wave - 1)) * 64 + l];
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
        return;
    }
    const double a0 = 1e-3 + l * 1e-5, a1 = 2e-3, a2 = 3e-3;
    double xa = 0.5, xb = 0.25;
    float t = cells[l], tn = cells[64 + l];
    d4 zero = {0, 0, 0, 0};
    d4 cpre = mma(a2, (double)t, zero);
    long t0 = __builtin_amdgcn_s_memtime();
    for (int kk = 0; kk < n; ++kk) {
        d4 c = mma(a0, xa, cpre);
        c = mma(a1, xb, c);
        if (V != 2) cpre = mma(a2, (double)tn, zero);
        tn = cells[((kk + 2) & 63) * 192 + l];
        xa = c[0], xb = c[1];
        if (V != 1) {
            float *p = &cells[(kk & 63) * 192 + l];
            *(lds_vf *)p = (float)xa;
            *(lds_vf *)(p + 64) = (float)xb;
            *(lds_vf *)(p + 128) = (float)c[2];
            *(lds_vi *)&step = kk + 1;
        }
    }
    asm volatile("" : "+v"(xa));
    long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = xa + xb;
}
template <int V>
void run(const char *name, int threads, double *d, long *c) {
    const int n = 4096;
    long h;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<256, threads>>>(d, c, n, 0); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<V><<<256, threads>>>(d, c, n, 0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-34s ticks per step %.1f   kernel %.3f ms -> %.1f ns per step\n", name, (double)h / n, ms, ms * 1e6 / n);
}
int main() {
    double *d; long *c;
    hipMalloc(&d, 1024 * 256 * 8); hipMalloc(&c, 32);
    run<0>("step as in the kernel, 1 wave", 64, d, c);
    run<1>("no LDS stores, 1 wave", 64, d, c);
    run<2>("no pre-issued t product, 1 wave", 64, d, c);
    run<0>("as in the kernel + 2 pollers", 192, d, c);
    return 0;
}
