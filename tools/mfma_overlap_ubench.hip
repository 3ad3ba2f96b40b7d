// Does MFMA work overlap with VALU work on the same SIMD?  f32-input MFMA vs f16-input MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND, bool DO_MFMA, bool DO_VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    f4 acc[4]; f2 p[8];
    for (int i = 0; i < 4; ++i) acc[i] = f4{seed, seed, seed, seed};
    for (int i = 0; i < 8; ++i) p[i] = f2{seed + i, seed - i};
    const f2 c2 = f2{seed * 0.5f, seed * 0.25f};
    const float a = seed + threadIdx.x, b = seed - threadIdx.x;
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(seed + i); bh[i] = (_Float16)(seed - i); }
    for (int it = 0; it < iters; ++it) {
        if (DO_MFMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            }
        }
        if (DO_VALU) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
        }
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
// wave-specialised variant: even blocks run MFMA only, odd blocks VALU only (2 + 2 waves per SIMD)
template <int KIND>
__global__ __launch_bounds__(256) void kspec(float* out, int iters, float seed, int mode)
{
    f4 acc[4]; f2 p[8];
    for (int i = 0; i < 4; ++i) acc[i] = f4{seed, seed, seed, seed};
    for (int i = 0; i < 8; ++i) p[i] = f2{seed + i, seed - i};
    const f2 c2 = f2{seed * 0.5f, seed * 0.25f};
    const float a = seed + threadIdx.x, b = seed - threadIdx.x;
    h8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(seed + i); bh[i] = (_Float16)(seed - i); }
    const bool do_mfma = (mode == 0) || (mode == 2 && (blockIdx.x & 1) == 0);
    const bool do_valu = (mode == 1) || (mode == 2 && (blockIdx.x & 1) == 1);
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#ifdef VALU_SCALAR
                    asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(p[i].x) : "v"(c2.x));
#else
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
#endif
                }
        }
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> float runspec(float* d, int mode, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kspec<KIND>), dim3(blocks), dim3(256), 0, 0, d, 2000, 1.0f, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kspec<KIND>), dim3(blocks), dim3(256), 0, 0, d, 2000, 1.0f, mode);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int KIND, bool M, bool V> float run(float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * 4), block(256);            // 4 blocks per CU -> 4 waves per SIMD
    hipLaunchKernelGGL((k<KIND, M, V>), grid, block, 0, 0, d, 2000, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, M, V>), grid, block, 0, 0, d, 2000, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 4);
    printf("per iteration: 4 MFMA + 32 v_pk_fma_f32 per wave, 4 waves/SIMD, 2000 iterations\n");
    printf("f32 MFMA 16x16x4 : mfma only %.3f ms | valu only %.3f ms | both %.3f ms\n", run<0, true, false>(d), run<0, false, true>(d), run<0, true, true>(d));
    printf("f16 MFMA 16x16x32: mfma only %.3f ms | valu only %.3f ms | both %.3f ms\n", run<1, true, false>(d), run<1, false, true>(d), run<1, true, true>(d));
    printf("wave-specialised (2 MFMA-only + 2 VALU-only waves per SIMD; alone = 2 waves per SIMD of that kind):\n");
    printf("f32: mfma alone %.3f ms | valu alone %.3f ms | side by side %.3f ms\n", runspec<0>(d, 0, 512), runspec<0>(d, 1, 512), runspec<0>(d, 2, 1024));
    printf("f16: mfma alone %.3f ms | valu alone %.3f ms | side by side %.3f ms\n", runspec<1>(d, 0, 512), runspec<1>(d, 1, 512), runspec<1>(d, 2, 1024));
    return 0;
}
