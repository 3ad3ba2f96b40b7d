// Would a 3-way bf16 split of the window fold beat the f32 matrix pipe?  ("levers not taken".)
// The fold is, per tap, a 16 x 8 real matrix applied to 16 frames: two v_mfma_f32_16x16x4_f32 (K = 4 each).  With x and the
// constants split into three bf16 parts each (24 mantissa bits) the products x_i c_j that matter are 6..8 terms x 8 = 48..64
// K-elements = two v_mfma_f32_16x16x32_bf16 per tap.  This measures what such an MFMA costs in a loop shaped like the
// kernel's (4 waves per SIMD, chains of 2 from C = 0, operands from LDS: two ds_read_b128 per MFMA instead of a
// ds_read_b64 + ds_read2_b32 per pair) next to the f32 form.
//   F3 / B3: 16 persistent accumulators x 2, constant register operands (pure issue rate), f32 16x16x4 / bf16 16x16x32
//   F6 / B6: 16 chains of 2 from C = 0, operands from LDS, results folded after the 32 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    __shared__ f4 atab[2 * 16 * 64];                      // 32 KB: A operands of 32 MFMAs (16 B per lane each)
    __shared__ f4 xb[4][2 * 16 * 16];                      // per wave: B operands
    __shared__ float pad[512];
    if (seed == 123.0f) pad[threadIdx.x] = seed;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * 16 * 64; i += 256) atab[i] = f4{seed + i, seed - i, seed, seed};
    for (int i = lane; i < 2 * 16 * 16; i += 64) xb[wv][i] = f4{seed * i, seed, seed, seed};
    __syncthreads();
    f2 tot = {0.0f, 0.0f};
    if (V == 0) {          // F3
        f4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f4{seed, seed, seed, seed};
        const float a = seed + lane, b = seed - lane;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
            }
        for (int i = 0; i < 16; ++i) tot += f2{acc[i].x + acc[i].z, acc[i].y + acc[i].w};
    } else if (V == 1) {   // B3
        f4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f4{seed, seed, seed, seed};
        b8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i + lane); b[i] = (__bf16)(seed - i); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[i], 0, 0, 0);
            }
        for (int i = 0; i < 16; ++i) tot += f2{acc[i].x + acc[i].z, acc[i].y + acc[i].w};
    } else if (V == 2) {   // F6: f32 chains of 2 from C = 0, operands from LDS (A: b64, B: 2 x b32)
        const f2* at = reinterpret_cast<const f2*>(atab) + lane;
        const float* xs = reinterpret_cast<const float*>(xb[wv]) + lane;
        for (int it = 0; it < iters; ++it) {
            f4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f2 a2 = at[i * 64];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, xs[i], f4{0, 0, 0, 0}, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, xs[i + 64], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) tot += f2{acc[i].x + acc[i].z, acc[i].y + acc[i].w};
        }
    } else {               // B6: bf16 chains of 2 from C = 0, operands from LDS (A: b128, B: b128 per MFMA)
        const b8* at = reinterpret_cast<const b8*>(atab) + lane;
        const b8* xs = reinterpret_cast<const b8*>(xb[wv]) + (lane & 15) + 16 * (lane >> 4);
        for (int it = 0; it < iters; ++it) {
            f4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[i * 128], xs[i * 2 % 8], f4{0, 0, 0, 0}, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[i * 128 + 64], xs[(i * 2 + 1) % 8 + 64], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) tot += f2{acc[i].x + acc[i].z, acc[i].y + acc[i].w};
        }
    }
    if (tot.x + tot.y == 12345.678f) out[0] = tot.x;
}

template <int V> void run(const char* name, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 250;
    dim3 grid(256 * 4), block(256);                        // 4 blocks per CU (42 KB LDS each) = 4 waves per SIMD
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d, iters, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (double(iters) * 32 * 4);   // 32 MFMAs per wave and iteration, 4 waves per SIMD
    printf("%-58s %7.3f ms  %6.2f ns/MFMA per SIMD\n", name, ms, per);
}

int main()
{
    float* d; hipMalloc(&d, 4096);
    run<0>("F3 f32 16x16x4, 16 persistent accumulators x2", d);
    run<1>("B3 bf16 16x16x32, 16 persistent accumulators x2", d);
    run<2>("F6 f32 16x16x4 chains of 2 from C=0, operands from LDS", d);
    run<3>("B6 bf16 16x16x32 chains of 2 from C=0, operands from LDS", d);
    run<0>("F3 again", d); run<1>("B3 again", d); run<2>("F6 again", d); run<3>("B6 again", d);
    return 0;
}
