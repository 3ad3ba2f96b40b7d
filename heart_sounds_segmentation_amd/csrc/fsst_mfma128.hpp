// fsst_mfma128.hpp -- second-generation synchrosqueeze core for the canonical window length
// nwin = 128 (the reference's Kaiser(128) configuration, /root/reference/main.py:153-158).
//
// Why a second kernel: measured on MI355X (profiles/r01_valu_ubench.txt) a 3-operand fp32 FMA
// costs ~4 cycles per wave64 instruction, an add ~2.8, a PACKED v_pk_{add,mul,fma}_f32 ~4.3 for two
// results, one wave per SIMD issues a VALU op only every ~5.4 cycles, and wave-uniform table
// constants cost SALU issue slots.  So this kernel
//   * runs the one dense contraction of the path -- the window multiply fused with the first
//     radix-8 decimation-in-frequency stage,  y_r[n] = sum_{q<8} x[t+n+16q] * C_r[n,q]  (a constant
//     16 x 8 real matrix applied at every tap n, for 16 frames at a time) -- on the fp32 matrix
//     pipe (v_mfma_f32_16x16x4_f32, bit-exact fp32 FMA chain), which is otherwise idle and runs
//     concurrently with the VALU; the constant operand lives in 32 VGPRs for the kernel's lifetime,
//     the frame operand is one contiguous ds_read_b32 per MFMA (hop-1 frames are shifted copies);
//   * finishes with two 16-point FFTs per lane written in packed (re,im) math: 74 v_pk ops each;
//   * 4 lanes per frame: lane group g = lane>>4 owns bin classes {g, 8-g} ({0,4} for g = 0), so a
//     bin k = 8j + r and its conjugate partner 128 - k are always in the same lane (two-for-one
//     real-FFT unpack without cross-lane traffic); the MFMA output fragment
//     D[row = 4g + {0,1,2,3}][frame] = {re, im} of class a, {re, im} of class b is exactly that.
//
// The arithmetic after the FFT (instantaneous-frequency test, cyclic scatter with the
// negative-frequency mirror, band truncation, abs / stack / raw epilogue, fp64 statistics partials)
// is the same as in fsst_kernels.hpp and follows oracle/fsst_oracle.c steps 4-7.
#pragma once
#include <hip/hip_runtime.h>

#include "fsst_kernels.hpp"

#ifndef HSS_MW128
#define HSS_MW128 3
#endif

namespace hssfsst {

using f2 = float __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));

struct Core128Params {
    const float* x;       // [batch][n]
    float* out;
    double* partials;     // [batch][nblk][4]
    const float* atab;    // MFMA A-operand constants [16 taps][2 k-halves][64 lanes]
    int n;
    int klo;
    int K;
    int mode;
    int nblk;             // wave tiles per signal
};

// cos / sin of 2*pi*j/16, j = 0..7
__device__ constexpr float kCos16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508978f,
                                        0.0f, -0.38268343236508973f, -0.70710678118654746f, -0.92387953251128674f};
__device__ constexpr float kSin16[8] = {0.0f, 0.38268343236508978f, 0.70710678118654746f, 0.92387953251128674f,
                                        1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508984f};

constexpr __host__ __device__ int bitrev4(int n) { return ((n & 1) << 3) | ((n & 2) << 1) | ((n & 4) >> 1) | ((n & 8) >> 3); }

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Radix-2 DIT butterfly on packed complex values, twiddle W = exp(-2*pi*i*TW/16).
template <int TW>
__device__ __forceinline__ void bfly16(f2& e, f2& o)
{
    f2 a, b;
    if constexpr (TW == 0) {
        a = e + o; b = e - o;
    } else if constexpr (TW == 4) {                     // W = -i: W o = (o.im, -o.re)
        const f2 t = f2{o.y, -o.x};
        a = e + t; b = e - t;
    } else {                                            // W = wr + i wi, wr = cos, wi = -sin
        constexpr float wr = kCos16[TW], wi = -kSin16[TW];
        const f2 t1 = pk_fma(f2{o.x, o.x}, f2{wr, wi}, e);
        a = pk_fma(f2{o.y, o.y}, f2{-wi, wr}, t1);      // e + W o
        b = pk_fma(e, f2{2.0f, 2.0f}, -a);              // e - W o = 2e - (e + W o)
    }
    e = a; o = b;
}

// In-place 16-point complex FFT (forward); input in bit-reversed order, output natural order.
__device__ __forceinline__ void fft16(f2 (&z)[16])
{
    static_for<4>([&](auto S) {
        constexpr int L = 2 << decltype(S)::value;
        constexpr int H = L / 2;
        constexpr int STEP = 16 / L;
        static_for<8>([&](auto B) {
            constexpr int b = decltype(B)::value;
            constexpr int j = b % H;
            constexpr int a = (b / H) * L + j;
            bfly16<j * STEP>(z[a], z[a + H]);
        });
    });
}

// ---- LDS planes of one 16-frame group: [16 frames][LDF] packed complex (re, im) per kept row, plus
// one dummy column (index K) that swallows the stores of sources whose own row is not kept.
__host__ __device__ constexpr int plane_ldf(int K) { return (K & 1) ? K + 2 : K + 1; }   // odd => b64 conflict-free

struct SrcEntry { float thr; int off; };      // per (source s, array a/b, lane group): threshold, byte offset of the own slot

// Exact (rare) path for a displaced source: oracle/fsst_oracle.c steps 4-6 in fp32.  `row_disp`
// points at this lane's frame row in the displaced plane; the own plane already holds V in the
// source's own slot (unconditional store), so V is first taken out again there.
__device__ __forceinline__ bool displaced_source(f2* row_disp, int klo, int K, int own_off, float kf,
                                                 bool mirror, float num, float den, f2 V)
{
    auto add = [&](int idx, float re, float im) {
        float* q = reinterpret_cast<float*>(row_disp + idx);
        __hip_atomic_fetch_add(q, re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(q + 1, im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    bool added = false;
    if (own_off < K * 8) { add(own_off >> 3, -V.x, -V.y); added = true; }
    float shift = num * __builtin_amdgcn_rcpf(den);
    if (!(fabsf(shift) <= 1.0e6f)) shift = 0.0f;        // NaN / inf / absurd -> 0 (fsst.m: ~isfinite)
    const float a = kf + shift;
    const float r = truncf(a + copysignf(0.5f, a));     // MATLAB round: half away from zero
    const int row = static_cast<int>(r) & 127;
    const int idx = row - klo;
    if (static_cast<unsigned>(idx) < static_cast<unsigned>(K)) { add(idx, V.x, V.y); added = true; }
    if (mirror) {                                       // negative-frequency twin: row -> 128 - row, value conj
        const int idm = ((128 - row) & 127) - klo;
        if (static_cast<unsigned>(idm) < static_cast<unsigned>(K)) { add(idm, V.x, -V.y); added = true; }
    }
    return added;
}

// One one-sided source bin k' held as packed spectrum value X = Z[k'] with conjugate partner
// P = Z[128 - k'] (Z = FFT of x (w + i dw') / 2 with the sign (-1)^k' folded into the constants):
//   V = X + conj(P) = (-1)^k' V[k'],   Vd' = (X - conj(P)) / i,   shift = -Im(Vd'/V) = num / den.
// V is stored unconditionally into the source's own slot (or the dummy column).  `thr`: 1/2 for
// kept rows (the source stays in its own row iff |shift| < 1/2); for rows outside the kept band
// the distance to the band minus 1/2 (closer moves cannot reach a kept row, nor can their mirror),
// so the exact path only runs for sources that may change the output.
__device__ __forceinline__ bool process_source(f2 X, f2 P, SrcEntry ent, char* row_own, f2* row_disp,
                                               int klo, int K, float kf, bool mirror)
{
    const f2 V = X + f2{P.x, -P.y};
    const f2 Vd = f2{X.y, -X.x} + f2{P.y, P.x};
    const f2 sq = V * V;
    const f2 cr = Vd * f2{V.y, V.x};
    const float den = sq.x + sq.y;
    const float num = cr.x - cr.y;
    *reinterpret_cast<f2*>(row_own + ent.off) = V;
    const bool moved = fabsf(num) >= fmaxf(ent.thr * den, 1.0e-37f);
    bool added = false;
    if (__ballot(moved) != 0ull) {
        if (moved) added = displaced_source(row_disp, klo, K, ent.off, kf, mirror, num, den, V);
    }
    return added;
}

// ------------------------------------------------------------------------------------------------
// grid = batch * nblk blocks of ONE wave; wave tile = FPW consecutive frames of one signal, walked
// in groups of 16 frames.  LDS per wave: xs[FPW + 127] | own[16][LDF] f2 | disp[16][LDF] f2 | tab.
// ------------------------------------------------------------------------------------------------
template <int FPW>
__global__ __launch_bounds__(64, HSS_MW128) void fsst_core128_kernel(Core128Params p)
{
    constexpr int XS = ((FPW + 127 + 3) / 4) * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.K, klo = p.klo, n = p.n;
    const int LDF = plane_ldf(K);
    float* xs = smem;
    f2* own_base = reinterpret_cast<f2*>(smem + XS);
    f2* disp_base = own_base + 16 * LDF;
    SrcEntry* tab = reinterpret_cast<SrcEntry*>(disp_base + 16 * LDF);   // [(s*2 + ab)][4 groups], s = 0..8

    const int lane = threadIdx.x;
    const int g = lane >> 4, j = lane & 15;
    const int blk = blockIdx.x % p.nblk;
    const long long b = blockIdx.x / p.nblk;
    const int t0 = blk * FPW;
    const float* xsig = p.x + b * static_cast<long long>(n);

    float A[32];                                         // MFMA A operand: row (lane&15), k (lane>>4)
#pragma unroll
    for (int i = 0; i < 32; ++i) A[i] = p.atab[i * 64 + lane];

    for (int i = lane; i < FPW + 127; i += 64) {         // xs[i] = xpad[t0 + i] = x[t0 + i - 64]
        const int gi = t0 + i - 64;
        xs[i] = (gi >= 0 && gi < n) ? xsig[gi] : 0.0f;
    }
    for (int i = lane; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
    const bool isg0 = (g == 0);
    for (int i = lane; i < 18 * 4; i += 64) {            // entry (s, ab, group)
        const int gg = i & 3, ab = (i >> 2) & 1, s = i >> 3;
        const int r = ab ? ((gg == 0) ? 4 : 8 - gg) : gg;
        const int slot = 8 * s + r - klo;
        const int d = max(slot - (K - 1), -slot);        // > 0: rows outside the kept band
        SrcEntry ent;
        ent.thr = (d > 0) ? static_cast<float>(d) - 0.5f : 0.5f;
        ent.off = (d > 0) ? K * 8 : slot * 8;            // dummy column K
        if (s == 8 && (ab || gg != 0)) { ent.thr = 3.0e38f; ent.off = K * 8; }   // only k' = 64 (class 0) exists
        tab[i] = ent;
    }
    __syncthreads();

    char* row_own = reinterpret_cast<char*>(own_base + j * LDF);
    f2* row_disp = disp_base + j * LDF;
    const SrcEntry* mytab = tab + g;
    const float rAf = static_cast<float>(g), rBf = static_cast<float>(isg0 ? 4 : 8 - g);
    f2 st_s = {0.0f, 0.0f}, st_q = {0.0f, 0.0f};        // (sum re, sum im), (sum re^2, sum im^2)
    bool dirty = false;

    for (int grp = 0; grp < FPW / 16; ++grp) {
        const int tg = t0 + grp * 16;
        if (tg >= n) break;
        const float* xb = xs + grp * 16 + lane;

        // ---- folded window + radix-8 stage on the matrix pipe: 16 taps x 2 k-halves
        f2 za[16], zb[16];
        static_for<16>([&](auto NN) {
            constexpr int nn = decltype(NN)::value;
            f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2 * nn], xb[nn], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2 * nn + 1], xb[nn + 64], acc, 0, 0, 0);
            za[bitrev4(nn)] = f2{acc.x, acc.y};
            zb[bitrev4(nn)] = f2{acc.z, acc.w};
        });
#ifdef HSS_SGB
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        static_for<14>([&](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
        fft16(za);
        fft16(zb);

        // ---- one-sided sources of this lane: classes rA (array a) and rB (array b)
        static_for<8>([&](auto SS) {
            constexpr int s = decltype(SS)::value;
            // partner of a[s]: class 0 -> a[(16-s)&15];  else b[15-s].  partner of b[s]: class 4 -> b[15-s]; else a[15-s]
            const f2 pa0 = za[(16 - s) & 15], pb = zb[15 - s], pa = za[15 - s];
            const f2 PA = f2{isg0 ? pa0.x : pb.x, isg0 ? pa0.y : pb.y};
            const f2 PB = f2{isg0 ? pb.x : pa.x, isg0 ? pb.y : pa.y};
            dirty |= process_source(za[s], PA, mytab[(s * 2 + 0) * 4], row_own, row_disp, klo, K,
                                    rAf + static_cast<float>(8 * s), !(isg0 && s == 0));
            dirty |= process_source(zb[s], PB, mytab[(s * 2 + 1) * 4], row_own, row_disp, klo, K,
                                    rBf + static_cast<float>(8 * s), true);
        });
        // k' = 64 (class 0, j = 8): its own partner; lanes of other groups idle (thr = huge, dummy slot)
        dirty |= process_source(za[8], za[8], mytab[(8 * 2 + 0) * 4], row_own, row_disp, klo, K, 64.0f, false);
        __syncthreads();

        // ---- epilogue for these 16 frames: element f -> (frame jj, kept row k)
        const bool wdirty = __any(dirty);
        const int nvalid = min(16, n - tg);
        if (p.mode == kModeRaw) {
            float2* dst = reinterpret_cast<float2*>(p.out) + (b * K) * static_cast<long long>(n) + tg;
            for (int e = lane; e < K * 16; e += 64) {
                const int k = e >> 4, jj = e & 15;
                if (jj < nvalid) {
                    f2 v = own_base[jj * LDF + k];
                    if (wdirty) v += disp_base[jj * LDF + k];
                    dst[static_cast<long long>(k) * n + jj] = make_float2(v.x, v.y);
                }
            }
        } else {
            const int C = (p.mode == kModeAbs) ? K : 2 * K;
            float* dst = p.out + (b * static_cast<long long>(n) + tg) * C;
            const int total = nvalid * K;
            int jj = lane / K, k = lane - jj * K;
            const int djj = 64 / K, dk = 64 - djj * K;
            for (int e = lane; e < total; e += 64) {
                f2 v = own_base[jj * LDF + k];
                if (wdirty) v += disp_base[jj * LDF + k];
                if (p.mode == kModeAbs) {
                    dst[jj * C + k] = sqrtf(fmaf(v.x, v.x, v.y * v.y));
                } else {
                    dst[jj * C + k] = v.x;
                    dst[jj * C + K + k] = v.y;
                    st_s += v;
                    st_q = pk_fma(v, v, st_q);
                }
                k += dk; jj += djj;
                if (k >= K) { k -= K; ++jj; }
            }
        }
        __syncthreads();
        if (wdirty) {
            for (int i = lane; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
            dirty = false;
            __syncthreads();
        }
    }
    if (p.mode != kModeStack) return;
    const double v0 = wave_sum(static_cast<double>(st_s.x)), v1 = wave_sum(static_cast<double>(st_q.x));
    const double v2 = wave_sum(static_cast<double>(st_s.y)), v3 = wave_sum(static_cast<double>(st_q.y));
    if (lane == 0) {
        double* part = p.partials + (b * p.nblk + blk) * 4;
        part[0] = v0; part[1] = v1; part[2] = v2; part[3] = v3;
    }
}

}  // namespace hssfsst
