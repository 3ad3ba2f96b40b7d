// fsst_dft.hpp -- synchrosqueeze core for ANY window length (odd, non-power-of-two, > 512): the reference takes
// nfft = len(window) for whatever array the caller passes (/root/reference/hss/transforms/synchrosqueeze.py:13-35,48).
//
// No FFT structure is assumed.  For every one-sided source bin k' the two spectra are plain windowed DFTs of the
// hop-1 frame,
//     V [k', t] = sum_n x[t + n] w [n] e^{-2 pi i k' (n + m) / N},      m = floor(N / 2)
//     Vd'[k', t] = sum_n x[t + n] dw'[n] e^{-2 pi i k' (n + m) / N},
// i.e. (V.re, V.im, Vd'.re, Vd'.im) of 4 sources x 16 frames is a (16 x N) by (N x 16) product whose right factor is a
// Hankel matrix of the signal tile: it runs on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains),
// A operand = constants from HBM / L2 (made in float64 by the host; the phase factor e^{-2 pi i m k' / N} of the
// "modified STFT", oracle/fsst_oracle.c step 5, is folded in -- it cancels in Vd'/V), B operand = shifted reads of the
// LDS tile.  The D fragment of a lane is one complete source of one frame.  Cost: 8 nf N flop per frame instead of
// ~6 N log N -- this kernel is the general fallback (nwin = 100: ~0.6 M windows/s), the radix kernels keep the
// power-of-two lengths.  Everything after the spectra follows oracle/fsst_oracle.c steps 4-7 as in the other kernels:
// stay-in-row test without a division, MATLAB rounding, cyclic row modulo N (not a power of two here), the
// negative-frequency twin, rounding ties decided in float64 (fsst_mfma128.hpp "Rounding ties"), abs / stack / raw
// epilogue with pivoted statistics partials.
#pragma once
#include <hip/hip_runtime.h>

#include "fsst_kernels.hpp"
#include "fsst_mfma128.hpp"

namespace hssfsst {

// this kernel's rounding-tie queue: {bin | frame << 16, V.re, V.im} per entry (the MFMA kernel's is more compact)
constexpr int kDftTieQueue = 256;
constexpr int kDftTieWords = 4 + 3 * kDftTieQueue;


struct DftParams {
    const float* x;       // [batch][n] (signal starts xstride apart)
    float* out;
    float* partials;      // [batch][groups][kPartFloats] (STACK)
    const float* atab;    // A operand [source block][k-step][64 lanes]
    const double* wtab;   // float64 {w, dw' (bin units)}[nwin]
    const double* twtab;  // float64 {cos, sin}(2 pi m / nwin)[nwin]
    int n, nwin, nf, klo, K, mode, col0, ncols;
    int nk4;              // k-steps = ceil(nwin / 4)
    int nblk4;            // source blocks = ceil(nf / 4)
    long long nitems;     // batch * groups
    long long xstride;
    float r2scale;
};

// G = 16-frame groups per work item (a tile of 16 G frames): every A-operand load feeds G MFMAs on G independent
// accumulators (the constants come from L2: one group per item is bound by those loads -- nwin 1024: 1.3 k windows/s).
__host__ __device__ constexpr int dft_xs_floats(int nk4, int G = 1) { return ((16 * G + 4 * nk4 + 3) / 4) * 4; }
__host__ __device__ constexpr int dft_wave_lds_floats(int nk4, int K, int G = 1)
{
    return dft_xs_floats(nk4, G) + 2 * 16 * G * plane_ldf(K) + 4 + kDftTieWords;
}

// One wave = one tile of G groups at a time (grid-stride over batch x tiles); blockDim = 64 x (waves that fit the LDS).
template <int G>
__global__ __launch_bounds__(512) void fsst_dft_kernel(DftParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int N = p.nwin, K = p.K, klo = p.klo, nf = p.nf, m = N / 2;
    const int LDP = plane_ldf(K);
    constexpr int F = 16 * G;                              // frames per tile
    const int XS = dft_xs_floats(p.nk4, G);
    float* xs = smem + wv * dft_wave_lds_floats(p.nk4, K, G);
    f2* plane = reinterpret_cast<f2*>(xs + XS);
    int* flag = reinterpret_cast<int*>(plane + F * LDP);
    int* tq = flag + 4;
    if (lane == 0) tq[0] = 0;
    const int ngroups = (p.ncols + 15) >> 4;             // statistics partials stay per 16-frame group
    const int ntiles = (p.ncols + F - 1) / F;
    const int cend = p.col0 + p.ncols;
    const bool even = (N & 1) == 0;

    auto add = [&](int jf, int idx, float re, float im) {       // kept rows only
        if (static_cast<unsigned>(idx) < static_cast<unsigned>(K)) {
            float* q = reinterpret_cast<float*>(plane + jf * LDP + idx);
            __hip_atomic_fetch_add(q, re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(q + 1, im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // source k' of frame jf lands in `row` (two-sided cyclic scatter; its twin N - k' in N - row, conjugated)
    auto land = [&](int jf, int kp, int row, float re, float im) {
        add(jf, row - klo, re, im);
        if (row != kp && kp != 0 && !(even && 2 * kp == N)) {
            const int rm = (row == 0) ? 0 : N - row;
            add(jf, rm - klo, re, -im);
        }
    };

    for (long long item = static_cast<long long>(blockIdx.x) * nwv + wv; item < p.nitems; item += static_cast<long long>(gridDim.x) * nwv) {
        const long long b = item / ntiles;
        const int tidx = static_cast<int>(item - b * ntiles);
        const int tg = p.col0 + tidx * F, tr = tidx * F;
        const float* xsig = p.x + b * p.xstride;
        // stage the zero-padded tile xs[i] = xpad[tg + i] = x[tg + i - m]; sum x^2 for the error bound of displaced cells
        float e2 = 0.0f, s1 = 0.0f, cnt = 0.0f;
        for (int i = lane; i < XS; i += 64) {
            const int gi = tg + i - m;
            const bool in = (i < F + N - 1 && gi >= 0 && gi < p.n);
            const float v = in ? xsig[gi] : 0.0f;
            xs[i] = v;
            e2 = fmaf(v, v, e2); s1 += v; cnt += in ? 1.0f : 0.0f;
        }
        const TileEnergy te = tile_energy(e2, s1, cnt);
        const float R2 = p.r2scale * te.E;
        for (int i = lane; i < F * LDP; i += 64) plane[i] = f2{0.0f, 0.0f};
        wave_sync();

        const float* xb = xs + j + g;                            // B[k = g][frame j] of k-step ks: xs[j + 4 ks + g]
        for (int blk = 0; blk < p.nblk4; ++blk) {
            const float* ab = p.atab + (static_cast<size_t>(blk) * p.nk4) * 64 + lane;
            f4 acc[G];
#pragma unroll
            for (int q = 0; q < G; ++q) acc[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
            for (int ks = 0; ks < p.nk4; ++ks) {
                const float a = ab[ks * 64];
#pragma unroll
                for (int q = 0; q < G; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[4 * ks + 16 * q], acc[q], 0, 0, 0);
            }
            const int kp = 4 * blk + g;                          // this lane's source, frames j + 16 q
#pragma unroll
            for (int q = 0; q < G; ++q)
            if (kp < nf) {
                const int jq = j + 16 * q;
                const float vr = acc[q].x, vi = acc[q].y, dr = acc[q].z, di = acc[q].w;
                const float den = fmaf(vr, vr, fmaf(vi, vi, 1.0e-37f));
                const float num = fmaf(dr, vi, -(di * vr));      // shift = -Im(Vd'/V) = num / den (bins)
                if (fabsf(num) < (0.5f - kTieMargin) * den) {
                    add(jq, kp - klo, vr, vi);                    // stays in its own row
                } else {
                    float shift = num * __builtin_amdgcn_rcpf(den);
                    if (!(fabsf(shift) <= 1.0e6f)) shift = 0.0f; // NaN / inf / absurd -> 0 (fsst.m: ~isfinite)
                    const float a = static_cast<float>(kp) + shift;
                    const float fr = a - floorf(a) - 0.5f, s1 = 1.0f + fabsf(shift);
                    bool queued = false;
                    if (fr * fr * den < kTieErr2 * s1 * s1 * R2 && den > kTieFloor2 * R2) {
                        const int slot = __hip_atomic_fetch_add(tq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (slot < kDftTieQueue) {
                            tq[4 + 3 * slot] = kp | (jq << 16);
                            tq[5 + 3 * slot] = __float_as_int(vr);
                            tq[6 + 3 * slot] = __float_as_int(vi);
                            queued = true;
                        }
                    }
                    if (!queued) {
                        const float r = truncf(a + copysignf(0.5f, a));      // MATLAB round: half away from zero
                        int row = static_cast<int>(r) % N;
                        if (row < 0) row += N;
                        land(jq, kp, row, vr, vi);
                    }
                }
            }
        }
        wave_sync();
        // rounding ties: float64 DFT of the one bin, all lanes (fsst_mfma128.hpp "Rounding ties")
        const int qn = min(__builtin_amdgcn_readfirstlane(tq[0]), kDftTieQueue);
        if (qn > 6) {
            // many cells (tonal input): one cell per lane, its N taps in sequence (fsst_mfma128.hpp resolve_ties)
            for (int base = 0; base < qn; base += 64) {
                const int e = base + lane;
                const bool act = e < qn;
                const int meta = act ? tq[4 + 3 * e] : 0;
                const int kp = meta & 0xffff, jf = meta >> 16;
                double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
                int ti = 0;                                      // (kp * nn) mod N, kept incrementally
#pragma unroll 4
                for (int nn = 0; nn < N; ++nn) {
                    const double x = static_cast<double>(xs[jf + nn]);
                    const double2 wd = reinterpret_cast<const double2*>(p.wtab)[nn];
                    const double2 cs = reinterpret_cast<const double2*>(p.twtab)[ti];
                    ti += kp; if (ti >= N) ti -= N;
                    const double xw = x * wd.x, xd = x * wd.y;
                    vr = fma(xw, cs.x, vr); vi = fma(-xw, cs.y, vi);
                    dr = fma(xd, cs.x, dr); di = fma(-xd, cs.y, di);
                }
                if (act) {
                    double shift = (dr * vi - di * vr) / (vr * vr + vi * vi);
                    if (!(fabs(shift) <= 1.0e6)) shift = 0.0;
                    const double a = static_cast<double>(kp) + shift;
                    const double r = (a >= 0.0) ? floor(a + 0.5) : -floor(0.5 - a);
                    long long row = static_cast<long long>(r) % N;
                    if (row < 0) row += N;
                    land(jf, kp, static_cast<int>(row), __int_as_float(tq[5 + 3 * e]), __int_as_float(tq[6 + 3 * e]));
                }
            }
        } else
        for (int e = 0; e < qn; ++e) {
            const int meta = __builtin_amdgcn_readfirstlane(tq[4 + 3 * e]);
            const int kp = meta & 0xffff, jf = meta >> 16;
            double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
            for (int nn = lane; nn < N; nn += 64) {
                const double x = static_cast<double>(xs[jf + nn]);
                const double2 wd = reinterpret_cast<const double2*>(p.wtab)[nn];
                const unsigned ti = (static_cast<unsigned>(kp) * static_cast<unsigned>(nn)) % static_cast<unsigned>(N);
                const double2 cs = reinterpret_cast<const double2*>(p.twtab)[ti];
                const double xw = x * wd.x, xd = x * wd.y;
                vr = fma(xw, cs.x, vr); vi = fma(-xw, cs.y, vi);
                dr = fma(xd, cs.x, dr); di = fma(-xd, cs.y, di);
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                vr += shfl_xor_f64(vr, off, lane); vi += shfl_xor_f64(vi, off, lane);
                dr += shfl_xor_f64(dr, off, lane); di += shfl_xor_f64(di, off, lane);
            }
            if (lane == 0) {
                double shift = (dr * vi - di * vr) / (vr * vr + vi * vi);
                if (!(fabs(shift) <= 1.0e6)) shift = 0.0;
                const double a = static_cast<double>(kp) + shift;
                const double r = (a >= 0.0) ? floor(a + 0.5) : -floor(0.5 - a);
                long long row = static_cast<long long>(r) % N;
                if (row < 0) row += N;
                land(jf, kp, static_cast<int>(row), __int_as_float(tq[5 + 3 * e]), __int_as_float(tq[6 + 3 * e]));
            }
        }
        const int q_all = __builtin_amdgcn_readfirstlane(tq[0]);  // undecided cells drawn, including those the queue had no room for
        if (qn > 0) {
            if (lane == 0) tq[0] = 0;
            wave_sync();
        }
#ifndef HSS_NO_EXACT
        // ---- "Exact groups" (fsst_mfma128.hpp): the tile is redone in float64 when none of its kept cells reaches
        //      kExactTheta R (the band holds only the far leakage of something outside it: float32 resolves ~4e-7 of the
        //      frame's spectrum norm, not of the band), or when the tie queue overflowed (cells it had no room for were
        //      rounded in float32).  Every (source, frame) cell: float64 DFT, the modified-STFT phase, float64 rounding, land.
        {
            float mxc = 0.0f;
            for (int i = lane; i < F * LDP; i += 64) { const f2 v = plane[i]; mxc = fmaxf(mxc, fmaf(v.x, v.x, v.y * v.y)); }
            const bool quiet = __builtin_amdgcn_ballot_w64(mxc > kExactTheta2 * R2) == 0ull && R2 > 0.0f;
            if (quiet || (te.dcdom && R2 > 0.0f) || q_all > kDftTieQueue) {      // (te.dcdom: an offset with little on top, fsst_kernels.hpp)
                wave_sync();
                for (int i = lane; i < F * LDP; i += 64) plane[i] = f2{0.0f, 0.0f};
                wave_sync();
                const int ncell = nf * F;
                for (int base = 0; base < ncell; base += 64) {
                    const int e = base + lane;
                    const bool act = e < ncell;
                    const int kp = act ? e / F : 0, jf = act ? e - kp * F : 0;
                    double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
                    int ti = 0;
#pragma unroll 4
                    for (int nn = 0; nn < N; ++nn) {
                        const double x = static_cast<double>(xs[jf + nn]);
                        const double2 wd = reinterpret_cast<const double2*>(p.wtab)[nn];
                        const double2 cs = reinterpret_cast<const double2*>(p.twtab)[ti];
                        ti += kp; if (ti >= N) ti -= N;
                        const double xw = x * wd.x, xd = x * wd.y;
                        vr = fma(xw, cs.x, vr); vi = fma(-xw, cs.y, vi);
                        dr = fma(xd, cs.x, dr); di = fma(-xd, cs.y, di);
                    }
                    if (act) {
                        double shift = (dr * vi - di * vr) / (vr * vr + vi * vi);
                        if (!(fabs(shift) <= 1.0e6)) shift = 0.0;
                        const double a = static_cast<double>(kp) + shift;
                        const double r = (a >= 0.0) ? floor(a + 0.5) : -floor(0.5 - a);
                        long long row = static_cast<long long>(r) % N;
                        if (row < 0) row += N;
                        // the plane holds the modified STFT: V e^{-2 pi i m k' / N} (oracle/fsst_oracle.c step 5)
                        const double2 ph = reinterpret_cast<const double2*>(p.twtab)[static_cast<unsigned>((static_cast<long long>(kp) * m) % N)];
                        land(jf, kp, static_cast<int>(row), static_cast<float>(vr * ph.x + vi * ph.y), static_cast<float>(vi * ph.x - vr * ph.y));
                    }
                }
                wave_sync();
            }
        }
#endif

        // ---- epilogue for these F frames, one 16-frame group (= one statistics partial) at a time
        for (int q = 0; q < G; ++q) {
            const int gidx = tidx * G + q;
            if (gidx >= ngroups) break;
            const int tgq = tg + 16 * q, trq = tr + 16 * q;
            const f2* pl = plane + 16 * q * LDP;
            const int nvalid = min(16, cend - tgq);
            if (p.mode == kModeRaw) {
                float2* dst = reinterpret_cast<float2*>(p.out) + (b * K) * static_cast<long long>(p.ncols) + trq;
                for (int e = lane; e < K * 16; e += 64) {
                    const int k = e >> 4, jj = e & 15;
                    if (jj < nvalid) {
                        const f2 v = pl[jj * LDP + k];
                        dst[static_cast<long long>(k) * p.ncols + jj] = make_float2(v.x, v.y);
                    }
                }
            } else if (p.mode == kModeAbs) {
                float* dst = p.out + (b * static_cast<long long>(p.ncols) + trq) * K;
                for (int e = lane; e < nvalid * K; e += 64) {
                    const int jj = e / K, k = e - jj * K;
                    const f2 v = pl[jj * LDP + k];
                    dst[e] = sqrtf(fmaf(v.x, v.x, v.y * v.y));
                }
            } else {
                const int C = 2 * K;
                float* dst = p.out + (b * static_cast<long long>(p.ncols) + trq) * C;
                const f2 piv = f2{pivot_med3(pl[0].x, pl[K >> 1].x, pl[K - 1].x), pivot_med3(pl[0].y, pl[K >> 1].y, pl[K - 1].y)};   // pivot_med3 of the group's frame 0
                float s_re = 0.0f, q_re = 0.0f, s_im = 0.0f, q_im = 0.0f;
                for (int e = lane; e < nvalid * C; e += 64) {
                    const int jj = e / C, c = e - jj * C;
                    float val;
                    if (c < K) { val = pl[jj * LDP + c].x; const float d = val - piv.x; s_re += d; q_re = fmaf(d, d, q_re); }
                    else       { val = pl[jj * LDP + c - K].y; const float d = val - piv.y; s_im += d; q_im = fmaf(d, d, q_im); }
                    dst[e] = val;
                }
                if (p.mode == kModeStack) {
                    const float w = piece_sums(s_re, q_re, s_im, q_im);
                    store_partial(p.partials + (b * ngroups + gidx) * kPartFloats, w, piv.x, piv.y);
                }
            }
        }
        wave_sync();
    }
}

}  // namespace hssfsst
