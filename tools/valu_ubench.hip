// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of fp32 add / fma /
// packed add / packed fma at 1, 2, 4, 8 waves per SIMD (256 CUs fully populated).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, int iters, float seed)
{
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; }
    const float c = seed * 0.5f; const f2 c2 = f2{c, c + 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
                if constexpr (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if constexpr (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(c));
                if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, float* d)
{
    const int iters = 4096;            // 32 instr per iter
    for (int wps : {1, 2, 4, 8}) {
        dim3 grid(256 * 4 * wps / 4), block(256);   // blocks of 4 waves -> one per SIMD; wps blocks per CU... approx
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = double(wps) * iters * 32;
        printf("%-12s waves/SIMD %d: %.3f ms  => %.2f ns/instr/SIMD  (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main()
{
    float* d; hipMalloc(&d, 4);
    run<0>("v_add_f32", d); run<1>("v_fma_f32", d); run<4>("v_fmac_f32", d); run<5>("v_fma_sgpr", d);
    run<2>("v_pk_add_f32", d); run<6>("v_pk_mul_f32", d); run<3>("v_pk_fma_f32", d);
    return 0;
}
