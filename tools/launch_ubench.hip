// Launch + teardown cost of an (almost) empty kernel in the shapes the core could use.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int n) { extern __shared__ float s[]; if (n == 12345) { s[threadIdx.x] = n; out[0] = s[0]; } }
static float run(int blocks, int threads, size_t lds)
{
    float* d; hipMalloc(&d, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms / 200 * 1e3f;
}
int main()
{
    printf("empty kernel, back-to-back launches, us per launch:\n");
    printf("  256 x 1024 threads, 135 KB LDS : %.2f\n", run(256, 1024, 135 * 1024));
    printf("  256 x 1024 threads,   0 KB LDS : %.2f\n", run(256, 1024, 0));
    printf(" 1024 x  256 threads,  40 KB LDS : %.2f\n", run(1024, 256, 40000));
    printf(" 1024 x  256 threads,   0 KB LDS : %.2f\n", run(1024, 256, 0));
    printf("  256 x   64 threads,   0 KB LDS : %.2f\n", run(256, 64, 0));
    printf("    1 x   64 threads,   0 KB LDS : %.2f\n", run(1, 64, 0));
    return 0;
}
