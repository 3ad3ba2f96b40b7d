// Prints the D layout of v_mfma_f64_16x16x4_f64: D[i][j] = 100 i + j from A[i][0] = i, B[0][j] = 1 plus A[i][1] = 1, B[1][j] = j / 100.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4v = double __attribute__((ext_vector_type(4)));
__global__ void k(double* out)
{
    const int l = threadIdx.x, i = l & 15, kk = l >> 4;
    const double a = kk == 0 ? 100.0 * i : kk == 1 ? 1.0 : 0.0;
    const double b = kk == 0 ? 1.0 : kk == 1 ? static_cast<double>(l & 15) : 0.0;
    d4v z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d4v{0, 0, 0, 0}, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = z[v];
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8); k<<<1, 64>>>(d); double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
