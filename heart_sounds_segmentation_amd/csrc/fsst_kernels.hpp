// fsst_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the FSST feature path.
//
// What the path computes (reference boundary: /root/reference/hss/transforms/synchrosqueeze.py:48
// `ssq.fsst(x, fs, window)` + the epilogue :50-111; algorithm steps as restated in
// oracle/fsst_oracle.c): for every sample t of a signal, the nwin-point DFT of the zero-padded
// hop-1 frame under the window w (V) and under the derivative window dw (Vd); the instantaneous
// frequency coordinate a = k - Im(Vd/V) * nwin/fs; the cyclic scatter S[round(a) mod nwin, t] +=
// (-1)^k V[k, t]; then band truncation and abs / z-score-stack / raw output.
//
// MI355X mapping (not a translation of an FFT-library call pattern):
//   * one frame per lane, one 64-frame tile per wavefront; the signal tile (+ nwin-1 halo) and the
//     per-lane scatter accumulators live in LDS, so HBM sees 4 B in and 4*out_floats B out per
//     sample and nothing else;
//   * nwin = 32*R.  The first radix-R decimation-in-frequency stage is FOLDED into the window
//     multiply: class r (bins k = R*j + r) is the 32-point FFT of
//         y_r[n] = sum_q x[t + n + 32 q] * C_r[n, q],   C_r[n,q] = win[n+32q] * W_R^{rq} * W_nwin^{rn}
//     with C_r precomputed on the host in fp64 and read through the scalar cache (wave-uniform),
//     so windowing, the first butterfly stage and its twiddles cost one FMA per (output, q);
//   * real input => only classes r <= R/2 are transformed.  The self-conjugate classes r = 0 and
//     r = R/2 carry V and Vd packed in ONE complex FFT (two-for-one, partner bin in the same
//     class); classes 0 < r < R/2 run one FFT for w and one for dw: bins j < 16 are class r's
//     one-sided bins, bins j >= 16 are (conjugated) the one-sided bins of class R - r;
//   * every one-sided source bin k' also feeds its negative-frequency mirror nwin - k' (value
//     conj, row (nwin - row) mod nwin), which reproduces the reference's two-sided cyclic scatter
//     exactly, including wrap-around into the kept band;
//   * the 32-point FFT is a fully unrolled radix-2 DIT in registers, twiddles as constants,
//     6 FMA-class ops per general butterfly (second output as 2e - first);
//   * MFMA is deliberately unused: at fp32 the matrix pipe has no rate advantage and the only
//     dense contraction (the folded first stage) is K <= 16 deep.
#pragma once
// Development hooks (phase ablations, clock probes, alternative tilings: -DHSS_ABLATE, -DHSS_CANON_ABLATE, -DHSS_T16_ABLATE, -DHSS_*_PROBE,
// -DHSS_NO_TIES, -DHSS_NO_EXACT, -DHSS_T16_NO_LAGPRIO, -DHSS_DEV_ONLY128 ...) change results or timing and exist for tools/ only: a build
// that carries one must say so with -DHSS_DEV (tools/dev.sh does); the library that ships is compiled without any of them.
#if !defined(HSS_DEV) && (defined(HSS_ABLATE) || defined(HSS_CANON_ABLATE) || defined(HSS_T16_ABLATE) || defined(HSS_FUSE_PROBE) || \
                          defined(HSS_STREAM_PROBE) || defined(HSS_CLOCKPROBE) || defined(HSS_CANON_PROBE) || defined(HSS_NO_TIES) || \
                          defined(HSS_NO_EXACT) || defined(HSS_T16_NO_LAGPRIO) || defined(HSS_DEV_ONLY128) || defined(HSS_TAIL_ENV) || \
                          defined(HSS_LDS_PAD) || defined(HSS_WPB_CANON) || defined(HSS_NO_GATE) || defined(HSS_T16_PLANES) || defined(HSS_T16_BLKPROBE))
#error "development hook without -DHSS_DEV: the shipped library carries none (tools/dev.sh builds development libraries)"
#endif
#include <hip/hip_runtime.h>

#include <utility>

namespace hssfsst {

constexpr int kModeRaw = 0, kModeAbs = 1, kModeStack = 2, kModeStackUnnorm = 3;

struct CoreParams {
    const float* x;       // [batch][n]
    float* out;           // per mode, see include/hssfsst.h
    float* partials;      // [batch][nblk][kPartFloats] pivoted sums per 64-frame tile (STACK only; see "Statistics")
    const float* ctab;    // class-folded window tables, [(R/2+1) classes][32 n][4R]
    int n;                // samples per signal
    int klo;              // first kept row
    int K;                // kept rows
    int mode;
    int nblk;             // tiles per signal
    int col0;             // first output column (frame centre) of every signal
    int ncols;            // number of output columns (== n for a whole-signal transform)
    int oneplane;         // 1: own and displaced values share one LDS plane (wide bands)
    const double* wtab;   // float64 tables of the rounding-tie path (see Scatter)
    const double* twtab;
    float r2scale;
    long long xstride;    // samples between the starts of consecutive signals (n for a dense batch; smaller for
                          // overlapping frames of one recording, hss/utils/preprocess.py:48-52)
};

// Window tables are read-only for the whole launch and indexed wave-uniformly: the constant
// address space lets hipcc fetch them with scalar loads (s_load_dwordx*) through the scalar cache.
using ctab_ptr = const float __attribute__((address_space(4)))*;

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// cos(2*pi*j/32), sin(2*pi*j/32), j = 0..15
__device__ constexpr float kCos32[16] = {
    1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
    0.70710678118654757f, 0.55557023301960229f, 0.38268343236508984f, 0.19509032201612833f,
    0.0f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f,
    -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f};
__device__ constexpr float kSin32[16] = {
    0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f,
    0.70710678118654746f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f,
    1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254546f,
    0.70710678118654757f, 0.55557023301960218f, 0.38268343236508989f, 0.19509032201612861f};

constexpr __host__ __device__ int bitrev5(int n)
{
    return ((n & 1) << 4) | ((n & 2) << 2) | (n & 4) | ((n & 8) >> 2) | ((n & 16) >> 4);
}

// Radix-2 DIT butterfly with twiddle W = exp(-2*pi*i*TW/32): (e, o) -> (e + W o, e - W o).
template <int TW>
__device__ __forceinline__ void bfly(float& er, float& ei, float& orr, float& oi)
{
    constexpr float c = 0.70710678118654752f;
    float ar, ai, br, bi;
    if constexpr (TW == 0) {
        ar = er + orr; ai = ei + oi; br = er - orr; bi = ei - oi;
    } else if constexpr (TW == 8) {          // W = -i:  W o = (oi, -or)
        ar = er + oi; ai = ei - orr; br = er - oi; bi = ei + orr;
    } else if constexpr (TW == 4) {          // W = (1 - i)/sqrt2:  W o = c (or + oi, oi - or)
        const float sr = orr + oi, si = oi - orr;
        ar = fmaf(c, sr, er); ai = fmaf(c, si, ei); br = fmaf(-c, sr, er); bi = fmaf(-c, si, ei);
    } else if constexpr (TW == 12) {         // W = (-1 - i)/sqrt2: W o = c (oi - or, -(or + oi))
        const float sr = oi - orr, si = orr + oi;
        ar = fmaf(c, sr, er); ai = fmaf(-c, si, ei); br = fmaf(-c, sr, er); bi = fmaf(c, si, ei);
    } else {                                 // W = wr + i wi, wr = cos, wi = -sin
        constexpr float wr = kCos32[TW], wi = -kSin32[TW];
        ar = fmaf(orr, wr, fmaf(-oi, wi, er));
        ai = fmaf(orr, wi, fmaf(oi, wr, ei));
        br = fmaf(2.0f, er, -ar);
        bi = fmaf(2.0f, ei, -ai);
    }
    er = ar; ei = ai; orr = br; oi = bi;
}

// In-place 32-point complex FFT (forward, e^{-i...}); input in bit-reversed order, output natural.
__device__ __forceinline__ void fft32(float (&re)[32], float (&im)[32])
{
    static_for<5>([&](auto S) {
        constexpr int L = 2 << decltype(S)::value;
        constexpr int H = L / 2;
        constexpr int STEP = 32 / L;
        static_for<16>([&](auto B) {
            constexpr int b = decltype(B)::value;
            constexpr int j = b % H;
            constexpr int a = (b / H) * L + j;
            bfly<j * STEP>(re[a], im[a], re[a + H], im[a + H]);
        });
    });
}

// Per-lane scatter context.  Two LDS planes per lane, each a column of a [2K][LD] array:
//   own  -- written exactly once per kept row k by source k itself (plain store, no RMW):
//           (-1)^k V[k] if the source stays in its own row, else 0;
//   disp -- zero-initialised; receives the (rare) displaced sources by read-modify-write.
// The lane owns its columns, so no atomics and a fixed summation order (deterministic output).
template <int LD>
struct Scatter {
    float* own;
    float* disp;
    int klo;
    int K;
    bool oneplane;        // own == disp (wide bands that do not fit two planes in LDS): own-row values are accumulated
    const float* frame;   // this lane's frame in the LDS tile: frame[n] = sample n of the zero-padded hop-1 frame
    const double* wtab;   // float64 {w, dw' (bin units)}[nwin]       } rounding ties, see scatter_source
    const double* twtab;  // float64 {cos, sin}(2 pi m / nwin)[nwin]  }
    float R2;             // error-bound scale of this tile (see fsst_mfma128.hpp "Rounding ties")
    __device__ __forceinline__ void add(int row, float re, float im) const
    {
        const int idx = row - klo;
        if (static_cast<unsigned>(idx) < static_cast<unsigned>(K)) {
            disp[idx * LD] += re;
            disp[(K + idx) * LD] += im;
        }
    }
};

// One one-sided source bin k' (0 <= k' <= nwin/2) with V = p + i q, Vd' = u + i v (Vd' already in
// bin units: the host scales the derivative window by nwin/fs, so -Im(Vd'/V) is a shift in bins).
// Follows oracle/fsst_oracle.c steps 4-6: shift = -Im(Vd/V) (non-finite -> 0), coordinate
// a = k' + shift, MATLAB round (half away from zero), cyclic row, value (-1)^k' V; the mirror
// source nwin - k' has coordinate nwin - a and value conj.
// Fast path: the source stays in row k' iff |shift| < 1/2 iff |num| < den/2 -- no division; its
// mirror then lands in row nwin - k' > nwin/2, outside every kept band.  V == 0 contributes 0
// wherever it lands and is treated as staying.  Only when some lane of the wave has a displaced
// cell does the wave run the exact rounding path for those lanes.
constexpr float kTieMarginG = 1.0f / 64.0f;      // (== kTieMargin, kTieErr2, kTieFloor2 of fsst_mfma128.hpp, where the
constexpr float kTieErr2G = 1.0e-12f;            //  error model behind them is described)
constexpr float kTieFloor2G = 1.0e-12f;

template <int NWIN, int LD>
__device__ __forceinline__ void scatter_source(const Scatter<LD>& sc, float kp, int kpi, float sgn,
                                               bool mirror, float p, float q, float u, float v)
{
    const float den = fmaf(p, p, q * q);
    const float num = fmaf(u, q, -(v * p));
    // (threshold kTieMarginG below 1/2: a |shift| that close to 1/2 is a rounding tie too and is decided in float64 below)
    const bool moved = (fabsf(num) >= (0.5f - kTieMarginG) * den) && (den > 0.0f);
    const float re = sgn * p, im = sgn * q;
    const int slot = kpi - sc.klo;                        // wave-uniform
    if (static_cast<unsigned>(slot) < static_cast<unsigned>(sc.K)) {
        if (sc.oneplane) {                                // wave-uniform
            sc.own[slot * LD] += moved ? 0.0f : re;
            sc.own[(sc.K + slot) * LD] += moved ? 0.0f : im;
        } else {
            sc.own[slot * LD] = moved ? 0.0f : re;
            sc.own[(sc.K + slot) * LD] = moved ? 0.0f : im;
        }
    }
    if (moved) {
        float shift = num * __builtin_amdgcn_rcpf(den);
        if (!(fabsf(shift) <= 1.0e6f)) shift = 0.0f;      // inf / absurd -> 0 (fsst.m: ~isfinite)
        const float a = kp + shift;
        float r = truncf(a + copysignf(0.5f, a));
        const float fr = a - floorf(a) - 0.5f, s1 = 1.0f + fabsf(shift);
        if (fr * fr * den < kTieErr2G * s1 * s1 * sc.R2 && den > kTieFloor2G * sc.R2) {
            // Rounding too close to call in float32 (its estimate of the shift is off by up to ~1e-2 bins for a small
            // cell that moves far): this bin of V and Vd' again by a float64 DFT of the lane's frame, float64 coordinate,
            // MATLAB round.  Divergent and slow (nwin taps), but only such cells pay.
            double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
            for (int n = 0; n < NWIN; ++n) {
                const double x = static_cast<double>(sc.frame[n]);
                const double2 wd = reinterpret_cast<const double2*>(sc.wtab)[n];
                const double2 cs = reinterpret_cast<const double2*>(sc.twtab)[(kpi * n) & (NWIN - 1)];
                const double xw = x * wd.x, xd = x * wd.y;
                vr = fma(xw, cs.x, vr); vi = fma(-xw, cs.y, vi);
                dr = fma(xd, cs.x, dr); di = fma(-xd, cs.y, di);
            }
            double sh = (dr * vi - di * vr) / (vr * vr + vi * vi);
            if (!(fabs(sh) <= 1.0e6)) sh = 0.0;
            const double ad = static_cast<double>(kpi) + sh;
            r = static_cast<float>((ad >= 0.0) ? floor(ad + 0.5) : -floor(0.5 - ad));
        }
        const int row = static_cast<int>(r) & (NWIN - 1);
        sc.add(row, re, im);
        if (mirror) sc.add((NWIN - row) & (NWIN - 1), re, -im);
    }
}

// Self-conjugate class (r = 0 or r = R/2): V and Vd packed in one complex FFT.
//   table row (class, n): [re(q=0..R-1) | im(q=0..R-1)] of 0.5*(w + i dw')[n+32q] * phase
template <int R, bool HALF, int LD>
__device__ __forceinline__ void packed_class(ctab_ptr tab, const float* xs, const Scatter<LD>& sc)
{
    constexpr int NWIN = 32 * R;
    float zr[32], zi[32];
    static_for<32>([&](auto NN) {
        constexpr int n = decltype(NN)::value;
        ctab_ptr c = tab + n * 4 * R;
        float sr = 0.0f, si = 0.0f;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const float xv = xs[n + 32 * q];
            sr = fmaf(xv, c[q], sr);
            si = fmaf(xv, c[R + q], si);
        }
        zr[bitrev5(n)] = sr;
        zi[bitrev5(n)] = si;
    });
    fft32(zr, zi);
    constexpr int NPRIM = HALF ? 16 : 17;
    static_for<NPRIM>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        constexpr int jp = HALF ? (31 - j) : ((32 - j) & 31);
        constexpr int kp = R * j + (HALF ? R / 2 : 0);
        const float p = zr[j] + zr[jp];
        const float q = zi[j] - zi[jp];
        const float u = zi[j] + zi[jp];
        const float v = zr[jp] - zr[j];
        scatter_source<NWIN, LD>(sc, static_cast<float>(kp), kp, (kp & 1) ? -1.0f : 1.0f,
                                 kp != 0 && kp != NWIN / 2, p, q, u, v);
    });
}

// Class pair (r, R - r), 0 < r < R/2: one FFT of the w-branch, one of the dw-branch.
//   table row (class, n): [w re | w im | dw re | dw im], each R wide
template <int R, int LD>
__device__ __forceinline__ void pair_class(ctab_ptr tab, int r, const float* xs, const Scatter<LD>& sc)
{
    constexpr int NWIN = 32 * R;
    float ar[32], ai[32], dr[32], di[32];
    static_for<32>([&](auto NN) {
        constexpr int n = decltype(NN)::value;
        ctab_ptr c = tab + n * 4 * R;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const float xv = xs[n + 32 * q];
            s0 = fmaf(xv, c[q], s0);
            s1 = fmaf(xv, c[R + q], s1);
            s2 = fmaf(xv, c[2 * R + q], s2);
            s3 = fmaf(xv, c[3 * R + q], s3);
        }
        ar[bitrev5(n)] = s0; ai[bitrev5(n)] = s1; dr[bitrev5(n)] = s2; di[bitrev5(n)] = s3;
    });
    fft32(ar, ai);
    fft32(dr, di);
    const float sgn = (r & 1) ? -1.0f : 1.0f;            // (-1)^(R j + r) = (-1)^r (R even here)
    static_for<32>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (j < 16) {                          // k' = R j + r
            scatter_source<NWIN, LD>(sc, static_cast<float>(R * j + r), R * j + r, sgn, true,
                                     ar[j], ai[j], dr[j], di[j]);
        } else {                                         // k' = nwin - (R j + r), conjugated
            scatter_source<NWIN, LD>(sc, static_cast<float>(NWIN - R * j - r), NWIN - R * j - r, sgn,
                                     true, ar[j], -ai[j], dr[j], -di[j]);
        }
    });
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// Statistics of the STACK epilogue (FSST._stack_real_imag, synchrosqueeze.py:78-85: mean and UNBIASED std of
// the real block and of the imag block over all K*n elements, torch.std semantics).
//
// Every core kernel emits one PARTIAL per piece of a signal (a 16-frame group of the MFMA kernel, a 64-frame tile
// of the generic kernel): 8 floats {S1re, S2re, S1im, S2im, p_re, p_im, -, -} with S1 = sum (v - p), S2 = sum (v - p)^2
// over the piece's valid cells and the pivot p = one cell of the piece itself.  Shifting by a value of the data keeps
// the FLOAT32 sums small against the spread (a plain float32 sum x^2 loses the variance when mean^2 >> variance,
// e.g. a DC-offset recording with a band that contains row 0).  From there on everything is float64: a piece's
// moments about zero  sum x = n p + S1,  sum x^2 = S2 + p (2 S1 + n p)  are exact to 1e-16 and are simply added up
// -- in a FIXED order: blocks of kStatBlock (4) consecutive pieces sequentially, then lane (block % 16) over blocks c, c + 16, ...,
// then a butterfly over the 16 block lanes -- so the statistics are run-to-run deterministic, independent of which
// wave produced which partial, and identical whether the sums are formed here (two-kernel path) or inside the fused
// kernel of fsst_mfma128.hpp, whose teams add their pieces in exactly this order.  The float64 cancellation in
// sum x^2 - (sum x)^2 / N is harmless whenever the float32 features themselves still resolve the spread.
// ------------------------------------------------------------------------------------------------
constexpr int kPartFloats = 8;
constexpr int kStatBlock = 4;              // pieces per block of the summation order (= the groups of one 64-frame chunk of
                                           // the team kernel, which publishes one float64 block sum per chunk)

// quantity q of a piece: 0 = sum re, 1 = sum re^2, 2 = sum im, 3 = sum im^2 (s1, s2, pivot of that block of columns)
__device__ __forceinline__ double piece_moment(int q, double s1, double s2, double piv, double cnt)
{
    // explicit fma: the two call sites (stats kernel, fused kernel) must round identically whatever the compiler's
    // contraction choices are
    return (q & 1) ? fma(piv, fma(cnt, piv, 2.0 * s1), s2) : fma(cnt, piv, s1);
}

// (lane passed in: callers inside a register-starved loop hand over an opaque copy so that nothing derived from it is
// hoisted out of their loop)
__device__ __forceinline__ double shfl_xor_f64(double v, int off, int lane)
{
    const int src = (lane ^ off) << 2;
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src, static_cast<int>(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src, static_cast<int>(b >> 32));
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}

// One full wave.  block_sum(blk, q) = sum of quantity q over the pieces of block blk (in piece order), called by lane
// (blk % 16, q) for blk = lane >> 2, (lane >> 2) + 16, ...  total = number of elements per block of columns (K * ncols).
// Returns {mean_re, 1/std_re, mean_im, 1/std_im} (float32, like the reference's float32 tensors; a zero variance gives
// 1/0 = inf and the z-score (v - mean) * inf = NaN for every element, as torch's 0/0).
// stats_finish: the part after the per-lane accumulation (lane (blk % 16, q) holds the sum of its blocks' quantity q);
// split off so that a kernel whose block sums arrive some other way runs the very same
// instructions on the very same numbers as the two-kernel path.
// The value of lane (lane ^ OFF), OFF in {4, 8, 16, 32}, without a trip through the LDS crossbar: row rotations (DPP) inside the
// 16-lane rows, row / half swaps (v_permlane16_swap, v_permlane32_swap) across them.  Which of the two candidates of a level
// comes from lane ^ OFF is read off the lane ids sent through the same instruction (no reliance on a rotation's direction).
template <int OFF>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned v, int lane)
{
    static_assert(OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "butterfly levels of stats_finish");
    const unsigned id = static_cast<unsigned>(lane);
    if constexpr (OFF == 8) {
        return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x128, 0xf, 0xf, false));      // row_ror:8 == xor 8
    } else if constexpr (OFF == 4) {
        const unsigned a = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x124, 0xf, 0xf, false));   // row_ror:4
        const unsigned b = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x12c, 0xf, 0xf, false));   // row_ror:12
        const unsigned ia = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(id), 0x124, 0xf, 0xf, false));
        return ia == (id ^ 4u) ? a : b;
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        const auto ri = __builtin_amdgcn_permlane16_swap(id, id, false, false);
        return ri[0] == (id ^ 16u) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        const auto ri = __builtin_amdgcn_permlane32_swap(id, id, false, false);
        return ri[0] == (id ^ 32u) ? r[0] : r[1];
    }
}
template <int OFF>
__device__ __forceinline__ double xor_lane_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = xor_lane_u32<OFF>(static_cast<unsigned>(b & 0xffffffffll), lane);
    const unsigned hi = xor_lane_u32<OFF>(static_cast<unsigned>(static_cast<unsigned long long>(b) >> 32), lane);
    return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}
// (it, it1 = 1 / total, 1 / (total - 1): correctly rounded divisions wherever they are made -- a kernel that finishes many signals
//  of one shape gets them from the host instead of dividing twice per signal)
__device__ __forceinline__ float4 stats_finish_pre(double acc, double it, double it1, int lane)
{
    // (butterfly over the 16 block lanes: the lower lane's value is always the left operand)
    { const double o = xor_lane_f64<4>(acc, lane);  acc = (lane & 4) ? (o + acc) : (acc + o); }
    { const double o = xor_lane_f64<8>(acc, lane);  acc = (lane & 8) ? (o + acc) : (acc + o); }
    { const double o = xor_lane_f64<16>(acc, lane); acc = (lane & 16) ? (o + acc) : (acc + o); }
    { const double o = xor_lane_f64<32>(acc, lane); acc = (lane & 32) ? (o + acc) : (acc + o); }
    auto from_lane = [&](int l) {
        const long long b = __double_as_longlong(acc);
        const int lo = __builtin_amdgcn_readlane(static_cast<int>(b & 0xffffffffll), l);
        const int hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), l);
        return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
    };
    const double sx_re = from_lane(0), sxx_re = from_lane(1), sx_im = from_lane(2), sxx_im = from_lane(3);
    // (two float64 divisions -- loop-invariant for a caller that finishes many signals of one shape, as the team kernel does
    //  once per chunk -- instead of four, and the square root in float32 of the float64 variance: the result is a float32
    //  1/std either way; measured 5 % of the team kernel with four divisions and two float64 square roots)
    const double mr = sx_re * it, mi = sx_im * it;
    const double vr = fma(-sx_re, mr, sxx_re) * it1, vi = fma(-sx_im, mi, sxx_im) * it1;
    return make_float4(static_cast<float>(mr), 1.0f / sqrtf(static_cast<float>(vr)),
                       static_cast<float>(mi), 1.0f / sqrtf(static_cast<float>(vi)));
}
// The same sums for a caller that needs them in ONE place only (fsst_team16_kernel's resolver: a wave at raised priority that fifteen
// siblings wait for): a DIRECTED reduction towards lanes 0..3 instead of the butterfly -- at every level only the lower partner adds, so
// lanes 0..3 receive exactly the additions the butterfly makes there (same operands, same order: the same bits), and the other lanes'
// values are never read.  One exchange per 32-bit half and level instead of the butterfly's select between two candidates.
__device__ __forceinline__ float4 stats_finish_lead(double acc, double it, double it1, int lane)
{
    auto halves = [](double v, unsigned& lo, unsigned& hi) {
        const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
        lo = static_cast<unsigned>(b); hi = static_cast<unsigned>(b >> 32);
    };
    auto join = [](unsigned lo, unsigned hi) { return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo)); };
    unsigned lo, hi;
    {   // lane + 4 (within the 16-lane row): a rotation by 4 one way or the other -- which way is read off the lane ids, once
        const int from = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_update_dpp(0, lane, 0x124, 0xf, 0xf, false));     // row_ror:4
        halves(acc, lo, hi);
        unsigned plo, phi;
        if (from == 4) {
            plo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), 0x124, 0xf, 0xf, false));
            phi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), 0x124, 0xf, 0xf, false));
        } else {
            plo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), 0x12c, 0xf, 0xf, false));      // row_ror:12
            phi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), 0x12c, 0xf, 0xf, false));
        }
        acc = acc + join(plo, phi);
    }
    {   // lane + 8: row_ror:8 either way
        halves(acc, lo, hi);
        const unsigned plo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(lo), 0x128, 0xf, 0xf, false));
        const unsigned phi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(hi), 0x128, 0xf, 0xf, false));
        acc = acc + join(plo, phi);
    }
    {   // lane + 16: the next row (v_permlane16_swap of a register with itself: the second result holds the odd rows in the even rows' place)
        halves(acc, lo, hi);
        const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        acc = acc + join(rl[1], rh[1]);
    }
    {   // lane + 32: the upper half (v_permlane32_swap: the second result holds the upper half in the lower half's place)
        halves(acc, lo, hi);
        const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        acc = acc + join(rl[1], rh[1]);
    }
    auto from_lane = [&](int l) {
        const long long b = __double_as_longlong(acc);
        const int l0 = __builtin_amdgcn_readlane(static_cast<int>(b & 0xffffffffll), l);
        const int h0 = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), l);
        return __longlong_as_double((static_cast<long long>(h0) << 32) | static_cast<unsigned>(l0));
    };
    const double sx_re = from_lane(0), sxx_re = from_lane(1), sx_im = from_lane(2), sxx_im = from_lane(3);
    const double mr = sx_re * it, mi = sx_im * it;
    const double vr = fma(-sx_re, mr, sxx_re) * it1, vi = fma(-sx_im, mi, sxx_im) * it1;
    return make_float4(static_cast<float>(mr), 1.0f / sqrtf(static_cast<float>(vr)),
                       static_cast<float>(mi), 1.0f / sqrtf(static_cast<float>(vi)));
}
__device__ __forceinline__ float4 stats_finish(double acc, double total, int lane)
{
    return stats_finish_pre(acc, 1.0 / total, 1.0 / (total - 1.0), lane);
}
template <class BlockSum>
__device__ __forceinline__ float4 stats_from_blocks(int nblocks, double total, BlockSum block_sum, int lane)
{
    const int q = lane & 3;
    double acc = 0.0;
    for (int blk = lane >> 2; blk < nblocks; blk += 16) acc += block_sum(blk, q);
    return stats_finish(acc, total, lane);
}

// The two-kernel path: partials in HBM, [nparts][kPartFloats]; fpp = frames per piece.
// (STRIDE: floats between consecutive pieces' partials -- kPartFloats in HBM, 6 for the copy fsst_team16_kernel keeps in LDS)
template <int STRIDE = kPartFloats>
__device__ __forceinline__ float4 signal_stats(const float* part, int nparts, int fpp, int ncols, int K, int lane)
{
    const int nblocks = (nparts + kStatBlock - 1) / kStatBlock;
    return stats_from_blocks(nblocks, static_cast<double>(K) * static_cast<double>(ncols), [&](int blk, int q) {
        double s = 0.0;
        const int g1 = min(nparts, (blk + 1) * kStatBlock);
        for (int g = blk * kStatBlock; g < g1; ++g) {
            const float* pp = part + static_cast<long long>(g) * STRIDE;
            const double cnt = static_cast<double>(min(fpp, ncols - fpp * g)) * static_cast<double>(K);
            const int h = q >> 1;                        // 0 = real block, 1 = imaginary block
            s += piece_moment(q, static_cast<double>(pp[2 * h]), static_cast<double>(pp[2 * h + 1]),
                              static_cast<double>(pp[4 + h]), cnt);
        }
        return s;
    }, lane);
}

// Wave-wide sums of the four per-lane accumulators of a piece in 10 instructions: two half swaps (v_permlane32_swap)
// put the re pair in lanes 0-31 and the im pair in lanes 32-63, a row swap (v_permlane16_swap) gives each of the
// four 16-lane rows one quantity, four DPP rotations finish inside the rows.  Result: every lane of row 0 / 1 / 2 / 3
// holds S1re / S2re / S1im / S2im.  Fixed order => deterministic.
__device__ __forceinline__ float piece_sums(float s_re, float q_re, float s_im, float q_im)
{
    const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s_re), __float_as_uint(s_im), false, false);
    const float u = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q_re), __float_as_uint(q_im), false, false);
    const float v = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    const auto r3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(u), __float_as_uint(v), false, false);
    float w = __uint_as_float(r3[0]) + __uint_as_float(r3[1]);
    w += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(w), 0x128, 0xf, 0xf, false));   // row_ror:8
    w += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(w), 0x124, 0xf, 0xf, false));   // row_ror:4
    w += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(w), 0x122, 0xf, 0xf, false));   // row_ror:2
    w += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(w), 0x121, 0xf, 0xf, false));   // row_ror:1
    return w;
}

// The pivot of a piece: the MEDIAN of three of its first frame's cells -- first, middle and last kept row.  A pivot only helps
// when it lies within a few standard deviations of the piece's mean: with every row kept, a recording riding on an offset has
// ONE row (row 0) hundreds of standard deviations away from all others; as the pivot (it was: "the piece's first cell") it
// made sum (v - p)^2 280 x the variance sum and the float32 partials lost 2e-4 of the standard deviation (Hann(512), all rows,
// 100 + cos: rel-L2 1.1e-4 with an EXACT transform).  The median is never the single outlier, and for a band of one or two
// rows it is a cell of the data as before.
__device__ __forceinline__ float pivot_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// Stores a piece's partial: `w` from piece_sums (rows 0..3), pivot (p_re, p_im) in every lane.
__device__ __forceinline__ void store_partial(float* part, float w, float p_re, float p_im)
{
    // no predicates (each costs a v_cmp + exec save / restore): the 16 lanes of a row store the same value to the same
    // word, even / odd lanes the two pivots
    const int lane = threadIdx.x & 63;
    part[lane >> 4] = w;
    part[4 + (lane & 1)] = (lane & 1) ? p_im : p_re;
}

// A staged tile's energy and whether an OFFSET dominates it.  float32 resolves a feature to ~4e-7 of its frame's spectrum norm;
// when a tile is an offset with little on top (AC amplitude below a tenth of the mean: S1^2 >= 0.99 n E over the n samples
// that are not zero padding) that is not 1e-4 of what the z-score makes of the small cells -- with every row kept the
// offset's own row is the largest cell, so the "no stored cell reaches 1e-2 R" test of "Exact groups" (fsst_mfma128.hpp) does
// not see it (profiles/r03_adversarial_parity.txt: the 8 misses of 1 800).  Such a tile's groups are redone in float64 like the
// quiet ones.  One piece_sums for all three sums: rows 0 / 1 / 2 of the wave carry E, S1, n.
constexpr float kDcTheta = 0.99f;
struct TileEnergy { float E, S1, C; bool dcdom; };   // sum x^2, sum x, samples inside the signal; offset-dominated
__device__ __forceinline__ TileEnergy tile_energy(float e2, float s1, float cnt)
{
    const float w = piece_sums(e2, s1, cnt, 0.0f);
    const float E = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), 0));
    const float S1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), 16));
    const float C = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), 32));
    return TileEnergy{E, S1, C, E > 0.0f && S1 * S1 >= kDcTheta * C * E};
}

// Per-signal statistics from the partials: one wave per signal.
// stats[b] = {mean_re, 1/std_re, mean_im, 1/std_im}.
// (gate != nullptr: the launch is the fallback of a team-kernel exec and runs only if that launch gave up -- *gate == gate_val)
__global__ __launch_bounds__(64) void fsst_stats_kernel(const float* partials, float4* stats, int nparts, int fpp,
                                                        int n, int K, const unsigned* gate = nullptr, unsigned gate_val = 0u)
{
    if (gate != nullptr && *gate != gate_val) return;
    const long long b = blockIdx.x;
    const float4 st = signal_stats(partials + b * nparts * kPartFloats, nparts, fpp, n, K, threadIdx.x & 63);
    if (threadIdx.x == 0) stats[b] = st;
}

// ------------------------------------------------------------------------------------------------
// Core kernel: one block = one TILE-sample stretch of one signal; one lane = one hop-1 frame.
// LDS: xs[TILE + nwin - 1 (+pad)] | own[2K][TILE + 1] | disp[2K][TILE + 1]; red[] aliases xs.
// p.oneplane != 0: own and disp are the same plane (half the LDS; every own-row value then costs a read-modify-write
// instead of a store) -- used by the host when two planes of the kept band do not fit 160 KB (nwin = 512, wide bands).
// ------------------------------------------------------------------------------------------------
template <int R, int TILE>
__global__ __launch_bounds__(TILE, 2) void fsst_core_kernel(CoreParams p)
{
    constexpr int NWIN = 32 * R;
    constexpr int XS = ((TILE + NWIN - 1 + 3) / 4) * 4;
    constexpr int LD = TILE + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* own = smem + XS;

    const int tid = threadIdx.x;
    const int blk = blockIdx.x % p.nblk;
    const long long b = blockIdx.x / p.nblk;
    const int t0 = p.col0 + blk * TILE;
    const int n = p.n;
    const int ncols = p.ncols, tr0 = t0 - p.col0;      // output rows are relative to col0
    const int K = p.K;
    const float* xsig = p.x + b * p.xstride;

    // stage the zero-padded signal tile: xs[i] = xpad[t0 + i] = x[t0 + i - nwin/2]
    float e2 = 0.0f, s1 = 0.0f, cnt = 0.0f;              // sum x^2 of the tile: error-bound scale of displaced cells; sum x, samples
    for (int i = tid; i < TILE + NWIN - 1; i += TILE) {
        const int g = t0 + i - NWIN / 2;
        const bool in = (g >= 0 && g < n);
        const float v = in ? xsig[g] : 0.0f;
        xs[i] = v;
        e2 = fmaf(v, v, e2); s1 += v; cnt += in ? 1.0f : 0.0f;
    }
    static_assert(TILE == 64, "tile_energy reduces one wave");
    const TileEnergy te = tile_energy(e2, s1, cnt);
    const float R2 = p.r2scale * te.E;
    const bool oneplane = p.oneplane != 0;
    float* disp = oneplane ? own : own + 2 * K * LD;
    for (int c = 0; c < 2 * K; ++c) disp[c * LD + tid] = 0.0f;
    __syncthreads();

    const Scatter<LD> sc{own + tid, disp + tid, p.klo, K, oneplane, xs + tid, p.wtab, p.twtab, R2};
    const float* myx = xs + tid;
    ctab_ptr tab = (ctab_ptr)p.ctab;
    packed_class<R, false, LD>(tab, myx, sc);
    if constexpr (R >= 2) packed_class<R, true, LD>(tab + (R / 2) * 32 * 4 * R, myx, sc);
    if constexpr (R >= 4 && R <= 8) {
        static_for<R / 2 - 1>([&](auto RR) {
            constexpr int r = decltype(RR)::value + 1;
            pair_class<R, LD>(tab + r * 32 * 4 * R, r, myx, sc);
        });
    } else if constexpr (R > 8) {
        for (int r = 1; r < R / 2; ++r) pair_class<R, LD>(tab + r * 32 * 4 * R, r, myx, sc);
    }
    __syncthreads();
    // fold the displaced plane into the own plane (each lane its own column: no hazard)
    if (!oneplane) {
        for (int c = 0; c < 2 * K; ++c) own[c * LD + tid] += disp[c * LD + tid];
    }
    __syncthreads();
    float* acc = own;
#ifndef HSS_NO_EXACT
    // ---- "Exact groups" (fsst_mfma128.hpp): no kept cell of the tile reaches 1e-2 R (R^2 = the bound of the tile's spectrum
    //      norms: the band holds only the far leakage of something outside it, and float32 resolves ~4e-7 of the frame's
    //      spectrum norm, not of the band) -> every lane redoes its frame in float64: all one-sided sources, float64 DFT
    //      of V and Vd', the float64 coordinate rounded half away from zero, the two-sided cyclic scatter into the kept rows.
    {
        float mxc = 0.0f;
        for (int k = 0; k < K; ++k) { const float re = acc[k * LD + tid], im = acc[(K + k) * LD + tid]; mxc = fmaxf(mxc, fmaf(re, re, im * im)); }
        if ((__builtin_amdgcn_ballot_w64(mxc > 1.0e-4f * R2) == 0ull || te.dcdom) && R2 > 0.0f) {
            for (int c = 0; c < 2 * K; ++c) acc[c * LD + tid] = 0.0f;
            const int klo = p.klo;
            for (int kp = 0; kp <= NWIN / 2; ++kp) {
                double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
#pragma unroll 4
                for (int nn = 0; nn < NWIN; ++nn) {
                    const double x = static_cast<double>(xs[tid + nn]);
                    const double2 wd = reinterpret_cast<const double2*>(p.wtab)[nn];
                    const double2 cs = reinterpret_cast<const double2*>(p.twtab)[(kp * nn) & (NWIN - 1)];
                    const double xw = x * wd.x, xd = x * wd.y;
                    vr = fma(xw, cs.x, vr); vi = fma(-xw, cs.y, vi);
                    dr = fma(xd, cs.x, dr); di = fma(-xd, cs.y, di);
                }
                double shift = (dr * vi - di * vr) / (vr * vr + vi * vi);
                if (!(fabs(shift) <= 1.0e6)) shift = 0.0;           // V == 0 or absurd -> 0 (fsst.m: ~isfinite)
                const double a = static_cast<double>(kp) + shift;
                const double r = (a >= 0.0) ? floor(a + 0.5) : -floor(0.5 - a);
                const int row = static_cast<int>(static_cast<long long>(r)) & (NWIN - 1);
                const double sg = (kp & 1) ? -1.0 : 1.0;              // the modified-STFT phase for even nwin
                const float re = static_cast<float>(vr * sg), im = static_cast<float>(vi * sg);
                const int idx = row - klo;
                if (static_cast<unsigned>(idx) < static_cast<unsigned>(K)) { acc[idx * LD + tid] += re; acc[(K + idx) * LD + tid] += im; }
                if (kp != 0 && kp != NWIN / 2) {                      // the negative-frequency twin N - k' lands in N - row, conjugated
                    const int idm = ((NWIN - row) & (NWIN - 1)) - klo;
                    if (static_cast<unsigned>(idm) < static_cast<unsigned>(K)) { acc[idm * LD + tid] += re; acc[(K + idm) * LD + tid] -= im; }
                }
            }
        }
    }
#endif

    const int valid = min(TILE, p.col0 + ncols - t0);
    if (p.mode == kModeRaw) {
        // complex64 [K][n], frequency-major: lane tid owns sample t0 + tid
        if (tid < valid) {
            float2* dst = reinterpret_cast<float2*>(p.out) + (b * K) * static_cast<long long>(ncols) + tr0 + tid;
            for (int k = 0; k < K; ++k)
                dst[static_cast<long long>(k) * ncols] = make_float2(acc[k * LD + tid], acc[(K + k) * LD + tid]);
        }
        return;
    }
    // time-major outputs: the tile's block of `valid * C` floats is contiguous in HBM
    const int C = (p.mode == kModeAbs) ? K : 2 * K;
    float* dst = p.out + (b * static_cast<long long>(ncols) + tr0) * C;
    const int total = valid * C;
    int tt = tid / C, c = tid - tt * C;
    const int dtt = TILE / C, dc = TILE - dtt * C;
    static_assert(TILE == 64, "one wave per tile: the statistics partial is a single-wave reduction");
    // pivots of this tile's statistics partial: its first frame's first kept row (a value of the data itself)
    const float p_re = pivot_med3(acc[0], acc[(K >> 1) * LD], acc[(K - 1) * LD]);
    const float p_im = pivot_med3(acc[K * LD], acc[(K + (K >> 1)) * LD], acc[(2 * K - 1) * LD]);
    float s_re = 0.0f, q_re = 0.0f, s_im = 0.0f, q_im = 0.0f;
    for (int e = tid; e < total; e += TILE) {
        float val;
        if (p.mode == kModeAbs) {
            const float re = acc[c * LD + tt], im = acc[(K + c) * LD + tt];
            val = sqrtf(fmaf(re, re, im * im));
        } else {
            val = acc[c * LD + tt];
            if (c < K) { const float d = val - p_re; s_re += d; q_re = fmaf(d, d, q_re); }
            else       { const float d = val - p_im; s_im += d; q_im = fmaf(d, d, q_im); }
        }
        dst[e] = val;
        c += dc; tt += dtt;
        if (c >= C) { c -= C; ++tt; }
    }
    if (p.mode != kModeStack) return;
    // per-tile statistics partial (fixed reduction order => run-to-run deterministic)
    const float w = piece_sums(s_re, q_re, s_im, q_im);
    store_partial(p.partials + (b * p.nblk + blk) * kPartFloats, w, p_re, p_im);
}

// grid = any number of blocks of 256 (the host sizes it to a fraction of the chip so the sweep can
// share the GPU with a concurrently running core kernel); block b handles signals b, b + grid, ...
// With `partials` != nullptr the block first reduces the signal's nblk fp64 partials itself (the arithmetic of
// fsst_stats_kernel, same order, same result) instead of reading `stats`: one launch less per transform.
// `slices` > 1 cuts every signal into that many contiguous pieces, one block each (small batches: a block per
// signal would leave most of the chip idle); the fused reduction is only used with slices == 1.
__global__ __launch_bounds__(256) void fsst_normalize_kernel(float* out, const float4* stats, const float* partials,
                                                             int nblk, int fpp, int n, int K, int nsignals, int slices,
                                                             const unsigned* gate = nullptr, unsigned gate_val = 0u)
{
    if (gate != nullptr && *gate != gate_val) return;    // (see fsst_stats_kernel)
    __shared__ float4 st_sh;
    const int tid = threadIdx.x;
    const int C = 2 * K;
    const int total = n * C;                             // per-signal element count (< 2^31, checked on the host)
    for (int unit = blockIdx.x; unit < nsignals * slices; unit += gridDim.x) {
        const int sig = unit / slices, sl = unit - sig * slices;
        float4 st;
        if (partials != nullptr) {
            if (tid < 64) {
                const float4 r = signal_stats(partials + static_cast<long long>(sig) * nblk * kPartFloats, nblk, fpp, n, K, tid);
                if (tid == 0) st_sh = r;
            }
            __syncthreads();
            st = st_sh;
            __syncthreads();                             // st_sh is rewritten for the block's next signal
        } else {
            st = stats[sig];
        }
        const float m_re = st.x, i_re = st.y, m_im = st.z, i_im = st.w;
        float* base = out + static_cast<long long>(sig) * total;
        if ((total & 3) == 0) {
            // linear float4 sweep of the signal's block (its start is 16-byte aligned because total % 4 == 0); the
            // column of a chunk is tracked incrementally (no integer division in the loop).  When 2K is not a
            // multiple of 4 a float4 can wrap from the end of one row into the next: per-element wrap test.
            float4* b4 = reinterpret_cast<float4*>(base);
            const int tot4 = total >> 2;
            const int i0 = static_cast<int>(static_cast<long long>(tot4) * sl / slices);
            const int i1 = static_cast<int>(static_cast<long long>(tot4) * (sl + 1) / slices);
            int c = static_cast<int>((static_cast<unsigned>(i0 + tid) * 4u) % static_cast<unsigned>(C));
            const int dc = static_cast<int>(1024u % static_cast<unsigned>(C));
            const bool rowwrap = (C & 3) != 0;
#ifndef HSS_NORM_UNROLL
#define HSS_NORM_UNROLL 4
#endif
#pragma unroll HSS_NORM_UNROLL
            for (int i = i0 + tid; i < i1; i += 256) {
                float4 v = b4[i];                        // (streaming load / store hints measured here: 0.147 vs 0.119 ms, worse)
                int c1 = c + 1, c2 = c + 2, c3 = c + 3;
                if (rowwrap) {
                    if (c1 >= C) c1 -= C;
                    if (c2 >= C) c2 -= C;
                    if (c3 >= C) c3 -= C;
                }
                v.x = (c < K) ? (v.x - m_re) * i_re : (v.x - m_im) * i_im;
                v.y = (c1 < K) ? (v.y - m_re) * i_re : (v.y - m_im) * i_im;
                v.z = (c2 < K) ? (v.z - m_re) * i_re : (v.z - m_im) * i_im;
                v.w = (c3 < K) ? (v.w - m_re) * i_re : (v.w - m_im) * i_im;
                b4[i] = v;
                c += dc;
                if (c >= C) c -= C;
            }
        } else {
            const int i0 = static_cast<int>(static_cast<long long>(total) * sl / slices);
            const int i1 = static_cast<int>(static_cast<long long>(total) * (sl + 1) / slices);
            for (int i = i0 + tid; i < i1; i += 256) {
                const int c = i % C;
                const float v = base[i];
                base[i] = (c < K) ? (v - m_re) * i_re : (v - m_im) * i_im;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Device counterpart of hss.moments (hss/moments/__init__.py:1-36) for feature chunks: per signal,
// merge the running {count, mean, M2} of the real and imag blocks with a new un-normalised chunk
// (Chan's pairwise merge == repeated application of update_mean/update_variance, in exact
// arithmetic).  grid = batch, block = 256.
// ------------------------------------------------------------------------------------------------
// Sums {sum re, sum re^2, sum im, sum im^2} (float64) of one signal's [n][2K] features in a FIXED order that a transform kernel can
// form on the fly (fsst_core128_kernel<STREAM>): the signal is cut into PIECES of 16 frames (16 x 2K contiguous floats, the
// last one shorter) -- the 16-frame groups of the transform kernels; within a piece lane l of ONE wave takes the elements
// 4 (l + 64 i) + {0, 1, 2, 3}, i = 0, 1, ... in that order (the lane-linear float4s of the wide-store epilogue), the 64 lanes are
// added by the xor butterfly wave_sum; the pieces' sums P_q are added as  wave_sum over l of (P_l + P_{l+64} + ...).
constexpr int kMomThreads = 1024;
constexpr int kMomPieceFrames = 16;
struct PlainLoad4 { __device__ float4 operator()(const float4* q) const { return *q; } };
// Agent-scope loads (sc1): data that blocks on other XCDs wrote with agent-scope stores during this launch
struct AgentLoad4 {
    __device__ float4 operator()(const float4* q) const
    {
        const unsigned long long* u = reinterpret_cast<const unsigned long long*>(q);
        const unsigned long long a = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float(static_cast<unsigned>(a)), __uint_as_float(static_cast<unsigned>(a >> 32)),
                           __uint_as_float(static_cast<unsigned>(b)), __uint_as_float(static_cast<unsigned>(b >> 32)));
    }
};
// one element into a lane's four accumulators (explicit fma: every site rounds alike)
__device__ __forceinline__ void mom_acc(double (&a)[4], float f, bool imag)
{
    const double v = static_cast<double>(f);
    if (imag) { a[2] += v; a[3] = fma(v, v, a[3]); } else { a[0] += v; a[1] = fma(v, v, a[1]); }
}
// the four elements of float4 number f of a piece (C = 2K floats per frame)
__device__ __forceinline__ void mom_acc4(double (&a)[4], float4 v, int f, int C, int K)
{
    int c = static_cast<int>((4u * static_cast<unsigned>(f)) % static_cast<unsigned>(C));
    int c1 = c + 1, c2 = c + 2, c3 = c + 3;
    if (c1 >= C) c1 -= C;
    if (c2 >= C) c2 -= C;
    if (c3 >= C) c3 -= C;
    mom_acc(a, v.x, c >= K); mom_acc(a, v.y, c1 >= K); mom_acc(a, v.z, c2 >= K); mom_acc(a, v.w, c3 >= K);
}
// sums of one piece (L floats at pb) by one whole wave; every lane gets them
__device__ inline void piece_moments(const float* pb, int L, int C, int K, int lane, double (&t)[4])
{
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const bool wide = (reinterpret_cast<uintptr_t>(pb) & 15) == 0;
    for (int f = lane; 4 * f < L; f += 64) {
        if (wide && 4 * f + 3 < L) mom_acc4(a, *reinterpret_cast<const float4*>(pb + 4 * f), f, C, K);
        else
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * f + u;
                if (e < L) mom_acc(a, pb[e], (e % C) >= K);
            }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = wave_sum(a[e]);
}
// the pieces' sums -> the signal's: lane l of one wave adds P_l, P_{l+64}, ... (piece(q, e) = quantity e of piece q), then the butterfly
template <class Piece>
__device__ inline void moments_from_pieces(int npieces, int lane, Piece piece, double (&out)[4])
{
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = lane; q < npieces; q += 64)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += piece(q, e);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = wave_sum(a[e]);
}
// A 1024-thread block: wave w takes the pieces w, w + 16, ... and adds piece q's sums into its lane q % 64 (lane l's pieces
// l, l + 64, ... all belong to wave l % 16, in increasing order: the order of moments_from_pieces); sh: [64][4] doubles.
// Every thread gets the totals.
__device__ inline void chunk_moments(const float* base, int n, int C, int K, int tid, double (*sh)[4], double out[4])
{
    const int lane = tid & 63, w = tid >> 6;
    const int npieces = (n + kMomPieceFrames - 1) / kMomPieceFrames;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = w; q < npieces; q += kMomThreads / 64) {
        double t[4];
        piece_moments(base + static_cast<long long>(q) * kMomPieceFrames * C, min(kMomPieceFrames, n - q * kMomPieceFrames) * C, C, K, lane, t);
        if (lane == (q & 63))
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += t[e];
    }
    if ((lane & 15) == w)
#pragma unroll
        for (int e = 0; e < 4; ++e) sh[lane][e] = a[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = wave_sum(sh[lane][e]);
}

// Chan merge of one block {sum, sum of squares over nb elements} into the running {count, mean, M2} at st[0..2];
// returns the float32 {mean, 1 / unbiased std} of the merged state
__device__ inline float2 merge_state(double* st, double s, double q, double nb)
{
    const double mean_b = s / nb;
    const double m2_b = q - s * mean_b;
    const double na = st[0], mean_a = st[1], m2_a = st[2];
    const double nn = na + nb;
    const double delta = mean_b - mean_a;
    const double mean_n = mean_a + delta * (nb / nn);
    const double m2_n = m2_a + m2_b + delta * delta * (na * nb / nn);
    st[0] = nn; st[1] = mean_n; st[2] = m2_n;
    return make_float2(static_cast<float>(mean_n), 1.0f / static_cast<float>(sqrt(m2_n / (nn - 1.0))));
}

__global__ __launch_bounds__(kMomThreads) void fsst_moments_merge_kernel(const float* feats, double* state, int n, int K)
{
    __shared__ double sh[64][4];
    const long long b = blockIdx.x;
    const int tid = threadIdx.x;
    const int total = n * 2 * K;                         // < 2^31, checked on the host
    double m[4];
    chunk_moments(feats + b * static_cast<long long>(total), n, 2 * K, K, tid, sh, m);
    if (tid < 2) merge_state(state + b * 6 + tid * 3, m[2 * tid], m[2 * tid + 1], static_cast<double>(K) * static_cast<double>(n));
}

// Running-moments normalisation (streaming): stats[b] from the {count, mean, M2} state of
// fsst_moments_merge_kernel, i.e. mean and unbiased std of everything seen so far.
__global__ __launch_bounds__(64) void fsst_stats_from_state_kernel(const double* state, float4* stats, int nsig)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nsig) return;
    const double* st = state + static_cast<long long>(b) * 6;
    const double vr = st[2] / (st[0] - 1.0), vi = st[5] / (st[3] - 1.0);
    stats[b] = make_float4(static_cast<float>(st[1]), 1.0f / static_cast<float>(sqrt(vr)),
                           static_cast<float>(st[4]), 1.0f / static_cast<float>(sqrt(vi)));
}

// One streaming step's tail in ONE launch (hssfsst_stream_step): block b merges the new un-normalised chunk of channel b
// into its running {count, mean, M2} and then normalises that chunk with the updated moments.  Same arithmetic, same
// order as fsst_moments_merge_kernel -> fsst_stats_from_state_kernel -> fsst_normalize_kernel: bit-identical results,
// two launches (and the statistics round trip through HBM) fewer per step.  grid = channels, block = 1024.
// The normalisation half of a streaming step's tail: NTH threads of one block, the chunk's [n][2K] floats at base.
// (Load4: how the chunk is read -- the wide path only: a caller that needs AgentLoad4 has a 16-byte addressable chunk.)
// Two halves so that a caller can have the chunk's first NPF float4s per thread in flight while it still forms the moments.
template <int NTH, int NPF, class Load4 = PlainLoad4>
__device__ inline void stream_normalize_load(const float* base, int n, int K, int tid, float4 (&v)[NPF], Load4 ld4 = Load4())
{
    const int total = n * 2 * K;
    if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const float4* b4 = reinterpret_cast<const float4*>(base);
        const int tot4 = total >> 2;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * NTH;
            v[k] = ld4(b4 + (i < tot4 ? i : 0));
        }
    }
}
// (mirror: a second destination for the normalised chunk -- the caller's pinned host buffer -- or null)
template <int NTH, int NPF, class Load4 = PlainLoad4>
__device__ inline void stream_normalize_apply(float* base, int n, int K, int tid, float4 st, float4 (&pre)[NPF], Load4 ld4 = Load4(), float* mirror = nullptr)
{
    const int C = 2 * K;
    const int total = n * C;                             // < 2^31, checked on the host
    const float m_re = st.x, i_re = st.y, m_im = st.z, i_im = st.w;
    if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        float4* b4 = reinterpret_cast<float4*>(base);
        const int tot4 = total >> 2;
        int c = static_cast<int>((static_cast<unsigned>(tid) * 4u) % static_cast<unsigned>(C));
        const int dc = static_cast<int>((static_cast<unsigned>(NTH) * 4u) % static_cast<unsigned>(C));
        auto one = [&](int i, float4 v) {
            int c1 = c + 1, c2 = c + 2, c3 = c + 3;
            if (c1 >= C) c1 -= C;
            if (c2 >= C) c2 -= C;
            if (c3 >= C) c3 -= C;
            v.x = (c < K) ? (v.x - m_re) * i_re : (v.x - m_im) * i_im;
            v.y = (c1 < K) ? (v.y - m_re) * i_re : (v.y - m_im) * i_im;
            v.z = (c2 < K) ? (v.z - m_re) * i_re : (v.z - m_im) * i_im;
            v.w = (c3 < K) ? (v.w - m_re) * i_re : (v.w - m_im) * i_im;
            b4[i] = v;
            if (mirror) reinterpret_cast<float4*>(mirror)[i] = v;
            c += dc;
            if (c >= C) c -= C;
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * NTH;
            if (i < tot4) one(i, pre[k]);
        }
        for (int i = tid + NPF * NTH; i < tot4; i += NTH) one(i, ld4(b4 + i));
    } else {
        for (int i = tid; i < total; i += NTH) {
            const float v = base[i];
            base[i] = ((i % C) < K) ? (v - m_re) * i_re : (v - m_im) * i_im;
        }
    }
}
template <int NTH, class Load4 = PlainLoad4>
__device__ inline void stream_normalize_block(float* base, int n, int K, int tid, float4 st, Load4 ld4 = Load4())
{
    float4 none[1];
    // (no prefetch: every float4 through the loop)
    const int C = 2 * K;
    const int total = n * C;
    if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const int tot4 = total >> 2;
        float4* b4 = reinterpret_cast<float4*>(base);
        const float m_re = st.x, i_re = st.y, m_im = st.z, i_im = st.w;
        int c = static_cast<int>((static_cast<unsigned>(tid) * 4u) % static_cast<unsigned>(C));
        const int dc = static_cast<int>((static_cast<unsigned>(NTH) * 4u) % static_cast<unsigned>(C));
        for (int i = tid; i < tot4; i += NTH) {
            float4 v = ld4(b4 + i);
            int c1 = c + 1, c2 = c + 2, c3 = c + 3;
            if (c1 >= C) c1 -= C;
            if (c2 >= C) c2 -= C;
            if (c3 >= C) c3 -= C;
            v.x = (c < K) ? (v.x - m_re) * i_re : (v.x - m_im) * i_im;
            v.y = (c1 < K) ? (v.y - m_re) * i_re : (v.y - m_im) * i_im;
            v.z = (c2 < K) ? (v.z - m_re) * i_re : (v.z - m_im) * i_im;
            v.w = (c3 < K) ? (v.w - m_re) * i_re : (v.w - m_im) * i_im;
            b4[i] = v;
            c += dc;
            if (c >= C) c -= C;
        }
    } else {
        (void)none;
        stream_normalize_apply<NTH, 1>(base, n, K, tid, st, none, ld4);       // (takes its scalar path)
    }
}
__global__ __launch_bounds__(kMomThreads) void fsst_stream_finish_kernel(float* feats, double* state, int n, int K)
{
    __shared__ double sh[64][4];
    __shared__ float4 st_sh;
    const long long b = blockIdx.x;
    const int tid = threadIdx.x;
    float* base = feats + b * static_cast<long long>(n) * (2 * K);
    double m[4];
    chunk_moments(base, n, 2 * K, K, tid, sh, m);
    if (tid < 2) {
        const float2 r = merge_state(state + b * 6 + tid * 3, m[2 * tid], m[2 * tid + 1], static_cast<double>(K) * static_cast<double>(n));
        if (tid == 0) { st_sh.x = r.x; st_sh.y = r.y; } else { st_sh.z = r.x; st_sh.w = r.y; }
    }
    __syncthreads();
    stream_normalize_block<kMomThreads>(base, n, K, tid, st_sh);
}

}  // namespace hssfsst
