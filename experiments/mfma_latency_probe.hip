// Latency of dependent v_mfma_f64_16x16x4_f64 on gfx950, one wavefront per SIMD (what the LDS-resident kernels are bound
// by): (a) chained through the accumulator, (b) chained through the B operand (result -> next product's operand),
// (c) independent products.  Cycles per product from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__global__ void k(double *out, long *cyc, int n) {
    const double a = 1.0 + threadIdx.x * 1e-3, b0 = 0.5;
    d4 c = {0, 0, 0, 0};
    long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) c = mma(a, b0, c);                       // (a) accumulator chain
    double s = c[0];
    asm volatile("" : "+v"(s));
    long t1 = __builtin_amdgcn_s_memtime();
    double b = b0;
    for (int i = 0; i < n; ++i) {                                        // (b) operand chain
        d4 z = {0, 0, 0, 0};
        d4 r = mma(a, b, z);
        b = r[0] * 1e-3;
    }
    asm volatile("" : "+v"(b));
    long t2 = __builtin_amdgcn_s_memtime();
    d4 c1 = {0, 0, 0, 0}, c2 = c1, c3 = c1, c4 = c1;
    for (int i = 0; i < n; i += 4) {                                     // (c) four independent chains
        c1 = mma(a, b0, c1);
        c2 = mma(a, b0, c2);
        c3 = mma(a, b0, c3);
        c4 = mma(a, b0, c4);
    }
    double s2 = c1[0] + c2[0] + c3[0] + c4[0];
    asm volatile("" : "+v"(s2));
    long t3 = __builtin_amdgcn_s_memtime();
    // (d) operand chain with a float round trip, as the kernels do: result -> cvt f32 -> cvt f64 -> operand
    double bb = b0;
    for (int i = 0; i < n; ++i) {
        d4 z = {0, 0, 0, 0};
        d4 r = mma(a, bb, z);
        r = mma(a, bb + 1.0, r);
        r = mma(a, bb + 2.0, r);
        bb = r[0] * 1e-3;
    }
    asm volatile("" : "+v"(bb));
    long t4 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cyc[0] = t1 - t0, cyc[1] = t2 - t1, cyc[2] = t3 - t2, cyc[3] = t4 - t3;
    }
    out[blockIdx.x * 64 + threadIdx.x] = s + b + s2 + bb;
}
int main() {
    double *d; long *c, h[4];
    hipMalloc(&d, 1024 * 64 * 8); hipMalloc(&c, 32);
    const int n = 4096;
    for (int blocks : {1, 1024}) {
        k<<<blocks, 64>>>(d, c, n); hipDeviceSynchronize();
        k<<<blocks, 64>>>(d, c, n); hipDeviceSynchronize();
        hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on this part (constant clock): convert with the shader clock separately
        printf("blocks %4d: memtime ticks per product: acc chain %.2f | operand chain %.2f | independent %.2f | 3-chain+operand (per 3) %.2f\n",
               blocks, (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n);
    }
    return 0;
}
