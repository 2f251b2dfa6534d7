// Probe of v_mfma_f64_16x16x4f64 operand / result layouts on gfx950 (experiment, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;  // 64 lanes
    // assumed: A (16x4): lane holds A[l%16][l/16];  B (4x16): lane holds B[l/16][l%16]
    const double a = A[(l % 16) * 4 + l / 16];
    const double b = B[(l / 16) * 16 + l % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}
int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = (double)(rand() % 17) - 8; B[i] = (double)(rand() % 13) - 6; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    // candidate result layouts
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int j = l % 16;
        if (D[l * 4 + v] != R[(4 * (l / 16) + v) * 16 + j]) ok1 = 0;   // i = 4*(l/16)+v
        if (D[l * 4 + v] != R[((l / 16) + 4 * v) * 16 + j]) ok2 = 0;   // i = l/16 + 4v
    }
    printf("layout i=4*(l/16)+v: %d   layout i=l/16+4v: %d\n", ok1, ok2);
    return 0;
}
