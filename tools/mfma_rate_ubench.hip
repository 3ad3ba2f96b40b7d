// What does one v_mfma_f32_16x16x4_f32 cost inside a loop shaped like the nwin=128 core's first stage?
// 40 KB of LDS per block pins the occupancy at 4 blocks per CU (4 waves per SIMD) like the real kernel; 8 rounds of blocks.
// Reading: V3 = the pure matrix-pipe rate; V7 - V3 = cost of VALU work the MFMAs do not feed (it adds, it does not
// overlap); V5/V10 fold the results in ONE dependent chain, which is latency- not issue-bound -- do not read them as
// "VALU reads of MFMA results are slow".
//   V0: 4 independent accumulators, chained forever (best case)
//   V1: per iteration 16 chains of length 2 starting from C = 0, results folded with v_pk_add (register operands)
//   V2: V1 with the operands read from LDS (A: ds_read_b64 per tap, B: two ds_read_b32 per tap)
// Also reports the shader clock the chip sustains (s_memtime ticks per s_memrealtime tick, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float seed)
{
    __shared__ f2 atab[16 * 64];
    __shared__ float xs[4][256];
    __shared__ float pad[7168];                            // 40 KB per block in total: exactly 4 blocks (16 waves) per CU
    if (seed == 123.0f) pad[threadIdx.x] = seed;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) atab[i] = f2{seed + i, seed - i};
    for (int i = lane; i < 256; i += 64) xs[wv][i] = seed * i;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    f2 tot = {0.0f, 0.0f};
    if (V == 0) {
        f4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f4{seed, seed, seed, seed};
        const float a = seed + lane, b = seed - lane;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) tot += f2{acc[i].x + acc[i].z, acc[i].y + acc[i].w};
    } else if (V == 3) {                                  // 16 persistent accumulators, no VALU at all
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
    } else if (V == 7) {                                  // V3 plus 32 v_pk_add on registers the MFMAs do not touch
        f4 acc[16]; f2 q[8];
        for (int i = 0; i < 16; ++i) acc[i] = f4{seed, seed, seed, seed};
        for (int i = 0; i < 8; ++i) q[i] = f2{seed + i, seed - i};
        const float a = seed + lane, b = seed - lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
        }
        for (int i = 0; i < 16; ++i) tot += f2{acc[i].x + acc[i].z, acc[i].y + acc[i].w};
        for (int i = 0; i < 8; ++i) tot += q[i];
    } else if (V == 9) {                                  // V5 with srcC = a zeroed VGPR quad instead of the inline constant 0
        float av[16], bv[16];
        for (int i = 0; i < 16; ++i) { av[i] = seed + lane + i; bv[i] = seed - lane * i; }
        f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < iters; ++it) {
            f4 acc[16];
            asm volatile("" : "+v"(z));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(av[i]), "+v"(bv[i]));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], z, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[i], av[i], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot += f2{acc[i].x, acc[i].y}; tot += f2{acc[i].z, acc[i].w}; }
        }
    } else if (V == 13) {                                 // V10 with the fold spread over 8 independent chains (like V7, but reading the accumulators)
        f4 acc[16]; f2 q[8];
        for (int i = 0; i < 16; ++i) acc[i] = f4{seed, seed, seed, seed};
        for (int i = 0; i < 8; ++i) q[i] = f2{seed + i, seed - i};
        const float a = seed + lane, b = seed - lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) { q[i & 7] += f2{acc[i].x, acc[i].y}; q[(i + 4) & 7] += f2{acc[i].z, acc[i].w}; }
        }
        for (int i = 0; i < 8; ++i) tot += q[i];
    } else if (V == 10) {                                 // V3 (persistent accumulators) + fold that reads them every iteration
        f4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f4{seed, seed, seed, seed};
        const float a = seed + lane, b = seed - lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot += f2{acc[i].x, acc[i].y}; tot += f2{acc[i].z, acc[i].w}; }
        }
    } else if (V == 4 || V == 5) {                        // chains of 2 from C = 0, constant operands; fold interleaved (4) or after all 32 (5)
        float av[16], bv[16];
        for (int i = 0; i < 16; ++i) { av[i] = seed + lane + i; bv[i] = seed - lane * i; }
        for (int it = 0; it < iters; ++it) {
            f4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(av[i]), "+v"(bv[i]));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[i], av[i], acc[i], 0, 0, 0);
                if (V == 4) { tot += f2{acc[i].x, acc[i].y}; tot += f2{acc[i].z, acc[i].w}; }
            }
            if (V == 5) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) { tot += f2{acc[i].x, acc[i].y}; tot += f2{acc[i].z, acc[i].w}; }
            }
        }
    } else if (V == 6) {                                  // V5 with the operands from LDS
        for (int it = 0; it < iters; ++it) {
            const float* xb = &xs[wv][(it & 3) * 16 + lane];
            const f2* myA = atab + lane;
            f4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f2 a2 = myA[i * 64];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, xb[i], f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, xb[i + 64], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot += f2{acc[i].x, acc[i].y}; tot += f2{acc[i].z, acc[i].w}; }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const float* xb = &xs[wv][(it & 3) * 16 + lane];
            const f2* myA = atab + lane;
#pragma unroll
            for (int nn = 0; nn < 16; ++nn) {
                f2 a2; float b0, b1;
                if (V == 2) { a2 = myA[nn * 64]; b0 = xb[nn]; b1 = xb[nn + 64]; }
                else { a2 = f2{seed + nn, seed - nn}; b0 = tot.x + nn; b1 = tot.y - nn; }
                f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b1, acc, 0, 0, 0);
                tot += f2{acc.x, acc.y}; tot += f2{acc.z, acc.w};
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (tot.x + tot.y == 12345.678f) out[0] = tot.x + pad[lane];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int V> void run(const char* name, float* d, unsigned long long* dc, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * 4 * 8), block(256);                  // 8 rounds of 4 blocks per CU (4 waves per SIMD, LDS-limited)
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d, dc, iters, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), grid, block, 0, 0, d, dc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
    const double mfma_per_simd = 8.0 * 4.0 * iters * 32.0;
    printf("%-44s %.3f ms  %.2f ns/MFMA per SIMD  shader clock %.0f MHz (memtime %llu / realtime %llu @100MHz)\n", name, ms,
           ms * 1e6 / mfma_per_simd, 100.0 * h[0] / (double)h[1], h[0], h[1]);
}
int main()
{
    float* d; unsigned long long* dc; hipMalloc(&d, 4); hipMalloc(&dc, 16);
    printf("32 v_mfma_f32_16x16x4_f32 per wave and iteration, 4 waves/SIMD, 125 iterations x 8 rounds of blocks\n");
    run<0>("V0 4 chained accumulators", d, dc, 125);
    run<1>("V1 16 chains of 2 from C=0 + pk_add fold", d, dc, 125);
    run<2>("V2 = V1 with operands from LDS", d, dc, 125);
    run<3>("V3 16 persistent accumulators x2, no VALU", d, dc, 125);
    run<4>("V4 chains of 2 from C=0, const operands, fold interleaved", d, dc, 125);
    run<5>("V5 = V4, fold after all 32 MFMAs", d, dc, 125);
    run<6>("V6 = V5 with operands from LDS", d, dc, 125);
    run<7>("V7 = V3 + 32 unrelated v_pk_add", d, dc, 125);
    run<9>("V9 = V5 with srcC = zero VGPRs", d, dc, 125);
    run<10>("V10 = V3 + fold reading the accumulators", d, dc, 125);
    run<13>("V13 = V10, fold over 8 independent chains", d, dc, 125);
    run<3>("V3 again", d, dc, 125);
    run<5>("V5 again", d, dc, 125);
    run<6>("V6 again", d, dc, 125);
    return 0;
}
