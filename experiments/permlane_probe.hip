// Which lanes do v_permlane16_swap_b32 / v_permlane32_swap_b32 exchange on gfx950, and does the "both operands equal"
// sum trick hold?  (csrc/admm_mfmac.hip.h sums over the four lanes j, 16 + j, 32 + j, 48 + j of an instance)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float inst_sum(float v) {
    // (inline asm: through __builtin_amdgcn_permlane16_swap the compiler of this image loses the second result when
    // both are consumed by one add — it emits v_add v0, v0, v0 — experiments/permlane_probe.hip.  The s_nop covers the
    // VALU-write -> permlane-read hazard the compiler would otherwise pad for.)
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(unsigned *out, float *fo) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    out[128 + l] = r2[0];
    out[192 + l] = r2[1];
    fo[l] = inst_sum((float)(1 << (l / 16)) * (1.f + (l % 16)));   // expect 15 * (1 + l % 16)
}
int main() {
    unsigned *d, h[256];
    float *fd, fh[64];
    hipMalloc(&d, sizeof h);
    hipMalloc(&fd, sizeof fh);
    k<<<1, 64>>>(d, fd);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(fh, fd, sizeof fh, hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane16_swap vdst(=lane)", "permlane16_swap src (=100+lane)", "permlane32_swap vdst", "permlane32_swap src"};
    for (int a = 0; a < 4; ++a) {
        printf("%s:\n", names[a]);
        for (int i = 0; i < 64; ++i) printf("%4u%s", h[a * 64 + i], i % 16 == 15 ? "\n" : "");
    }
    printf("inst_sum (expect 15 * (1 + lane %% 16) in every row):\n");
    for (int i = 0; i < 64; ++i) printf("%5.0f%s", fh[i], i % 16 == 15 ? "\n" : "");
    return 0;
}
