// fsst_canon128.hpp -- the transform for the reference's own configuration class: nwin = 128, time-major [re | im]
// output (FSST(stack=True), /root/reference/hss/transforms/synchrosqueeze.py:61-63,67-89) and a kept band that is a
// COMPILE-TIME constant (KLO, KC) inside rows 0..31 -- the canonical [25, 200] Hz at fs = 1000 of
// /root/reference/main.py:153-158 is (4, 22).  Every other configuration keeps fsst_core128_kernel (fsst_mfma128.hpp); the
// algorithm (oracle/fsst_oracle.c steps 1-7) and the rounding-tie machinery are the same, what changes is the cost:
//
//  * the window fold (window multiply + first radix-8 stage, 32 v_mfma_f32_16x16x4_f32 per 16 frames = 27 % of the
//    round-2 kernel's issue time) runs on the 16-bit matrix pipe with SPLIT operands: every sample and every constant is
//    a pair of halves, x = x1 + x2, c = c1 + c2 (22 significant bits each, scaled by powers of two into the middle of the
//    f16 range), and the four products x1 c1, x1 c2, x2 c1, x2 c2 of the 8 fold terms are exactly the K = 32 of ONE
//    v_mfma_f32_16x16x32_f16 per tap (fp32 accumulation; the products of two halves are exact in it).  Error per term
//    <= 2^-22 |x c| -- the same order as the fp32 FMA chain it replaces -- and every rounding decision that float32
//    cannot make is still made in float64 from the signal's own float32 samples (resolve_bitmap reads them from HBM).  The tile of a group is the ALIGNED
//    64-frame window of its signal whatever kernel or chunking processes it, so the power-of-two scale -- and with it every
//    bit of the result -- does not depend on the path (two-launch, one CU per signal, team);
//  * the band is a constant: the own plane covers rows 4 floor(KLO / 4) .. only (24 instead of 32 columns: that is what
//    pays for the 8-byte sample records and the 16 kB operand table in LDS), sources whose half-stripe holds no kept row
//    are not stored, the statistics and the wide-store pass are straight-line code, and a source that is `need` rows
//    away from the band (or its negative-frequency twin) takes the rare path only when |shift| >= need - 1/2 (less a
//    margin): a source that moves without reaching the band changes nothing, which for the canonical band takes the
//    upper half of the spectrum out of the rare path altogether.
#pragma once
#include "fsst_mfma128.hpp"

namespace hssfsst {

using h8 = _Float16 __attribute__((ext_vector_type(8)));
using u2 = unsigned __attribute__((ext_vector_type(2)));
using u4 = unsigned __attribute__((ext_vector_type(4)));
using lds_u2 = __attribute__((address_space(3))) u2;
using lds_u4 = __attribute__((address_space(3))) u4;

// A rare path that loads from memory (float64 passes, offset tables, register reloads the compiler puts there) ends with an explicit
// "all my loads are back": the compiler's wait-count pass merges what is pending over ALL paths into the hot loop's head, and a reload
// left pending by the rounding-tie path made it wait for every outstanding memory operation -- the next tile's samples and the previous
// group's stores, both issued a moment ago on purpose -- in the middle of the next group's fold.  (s_waitcnt vmcnt(0), expcnt / lgkmcnt
// untouched; as a builtin, not inline assembly: the pass must see it.)
#define HSS_RARE_VMEM_DONE() __builtin_amdgcn_s_waitcnt(0x0f70)
constexpr int kCanonTileFrames = 64;                     // frames per aligned tile (4 groups = one statistics block)
constexpr int kCanonRecs = 192;                          // sample records per tile: 64 + 127, rounded up
constexpr int kCanonOpFloats = 16 * 64 * 4;              // f16 A operand: [16 taps][64 lanes][8 halves] = 16 kB
constexpr int kCanonAtabFloats = kCanonOpFloats + 4 * 128;   // + {cos, sin}(2 pi m / 128) as float64 (rounding-tie path), 2 kB
constexpr int kCanonLdsTabFloats = kCanonAtabFloats + 4 * 33 * 2;   // what the kernels keep in LDS: + the interior frame of the offset table ("Offsets"), 1 kB
constexpr int kCanonErrMul = 4;                          // tau^2 of the rounding-tie bound: 4 kTieErr2 (tau = 2e-6 (1 + |shift|) R / |V|)
// (rounding ties: the bitmap of fsst_mfma128.hpp, "Rounding ties"; the float64 path reads the signal's own samples)
constexpr int kCanonFlagWords = 8;                       // [1] the tie bitmap has a bit (the others: spare)
constexpr int kCanonTieWords = 32;                       // [0..31] bitmap (flag[1] = "some bit is set")

template <int KLO, int KC>
struct CanonCfg {
    static_assert(KC % 2 == 0 && KC >= 2 && KC <= 24 && KLO >= 0 && KLO + KC <= 32, "even band of <= 24 rows inside rows 0..31");
    static constexpr int NWIN = 128;
    static constexpr int KHI = KLO + KC - 1;
    static constexpr int H0 = KLO / 4, H1 = KHI / 4;                 // half-stripes (4 rows) that hold kept rows
    static constexpr int COV0 = 4 * H0, COVN = 4 * (H1 - H0 + 1);    // the own plane's columns: rows COV0 .. COV0 + COVN - 1
    static constexpr int LD = odd_up(COVN + 1), LDF = plane_ldf(KC), KOFF = KLO - COV0;
    // lane group g holds classes rA = g and rB = (g ? 8 - g : 4): the A source of stripe s sits in half-stripe 2 s (rows
    // 8 s .. 8 s + 3), the B source in half-stripe 2 s + 1
    static constexpr bool stored(int s, int half) { return 2 * s + half >= H0 && 2 * s + half <= H1; }
    static constexpr int need_row(int row)               // rows between `row` and the nearest kept row or twin of one (cyclic)
    {
        int best = NWIN;
        for (int k = KLO; k <= KHI; ++k)
            for (int tw = 0; tw < 2; ++tw) {
                const int r = tw ? (NWIN - k) % NWIN : k;
                int d = row > r ? row - r : r - row;
                if (NWIN - d < d) d = NWIN - d;
                if (d < best) best = d;
            }
        return best;
    }
    static constexpr int need(int s, int half)
    {
        int best = NWIN;
        for (int r = 0; r < 4; ++r) {
            const int d = need_row(8 * s + 4 * half + r);
            if (d < best) best = d;
        }
        return best;
    }
    // |shift| from which the source may change a kept row; below it the source is left alone (exact: it cannot reach the
    // band).  The margin hands coordinates within (1 + need) / 64 of the deciding half-integer to the rare path, whose
    // error bound then decides between float32 and float64.
    static constexpr float thr(int s, int half)
    {
#ifdef HSS_NO_TIES
        return need(s, half) <= 1 ? 0.5f : static_cast<float>(need(s, half)) - 0.5f;
#else
        const int nd = need(s, half);
        return (nd <= 1 ? 0.5f : static_cast<float>(nd) - 0.5f) - static_cast<float>(1 + (nd <= 1 ? 0 : nd)) * kTieMargin;
#endif
    }
    static_assert(KLO >= 1, "row 0 is its own twin: acc_source / canon_put want the band above it");
    static constexpr int wave_floats(int planes = 1) { return 2 * kCanonRecs + planes * 2 * 16 * LD + kCanonFlagWords + kCanonTieWords; }
};

// Host side of the f16 operand table: entry (tap n, lane l, half h): lane l = (kk, row i); h -> fold term q = kk + 4 (h >> 2),
// product h & 3 = {x1 c1, x1 c2, x2 c1, x2 c2} -> the constant's half c1 (h even) or c2 (h odd).  Row i -> (class, re / im) as
// in the fp32 table of fsst_core128_kernel.  `cs` = 2^sc scales the constants into [2^13, 2^14).
// (built in hssfsst.hip: canon_build_atab)

#ifndef HSS_OFFERR
#define HSS_OFFERR 0.0625f
#endif
constexpr float kOffsetErr2 = HSS_OFFERR;                   // (5e-7 / 2e-6)^2: V = V' + mean x Yc is good to 4e-7 R' + 1.2e-7 |mean Yc|, and |mean Yc| <= |V| + R':
                                                         // the tie bound of such a tile is tau^2 = 4e-12 (1 + |shift|)^2 (R'^2 + kOffsetErr2 |V|^2) / |V|^2
// One tile's scale, as the kernels hand it around (wave-uniform).
struct CanonTile {
    float R2s;            // error-bound scale of the tile in SCALED units (see "Rounding ties" in fsst_mfma128.hpp)
    float inv;            // 1 / (sample scale x constant scale): features = plane values x inv (a power of two)
    float r2s;            // the plan's r2scale in scaled units: R^2 = r2s x (sum of squares of scaled samples)
    float mean_s;         // != 0: the tile's mean was taken out of the records; this is it, in scaled sample units ("Offsets")
    // != 0 with mean_s: the added offset term's own rounding, relative to |V|^2, in units of the tie bound (made where the rare path wants
    // it: as a field it was one more value held across the transform, and the first one the register allocator sent to scratch)
    __device__ __forceinline__ float eoff() const { return mean_s != 0.0f ? kOffsetErr2 : 0.0f; }
};

// Offsets.  float32 resolves a feature to ~4e-7 of its frame's spectrum norm.  A recording that rides on an offset (an ADC bias,
// 1 + t, pcg + 3) has a spectrum norm that is all offset while the kept band holds its far leakage plus the content: round 3 sent
// every such group to the float64 path (2.59 vs 0.208 ms per 1024 windows).  The transform is linear, so a tile whose mean
// carries at least half of its energy is staged WITHOUT it (records and scale of x - mean over the samples inside the signal:
// the fold then works at the resolution of the content) and the mean's own spectrum is added where the sources are formed:
// Z of a lane's two spectra += mean x Z of the all-ones frame, float64 on the host (hssfsst.hip): ONE table row per bin
// for interior frames, a table of the 64 + 63 frames whose window reaches over the start / the end of the signal (both at once
// for signals shorter than a window: ones = left + right - interior).  32 packed multiply-adds per lane and group, for such
// tiles only, in one block between the spectra and the source stage.  (The mean's FOLD as the matrix instructions' C operand was tried first: its taps are as large as the offset,
// they cancel only in the 16-point spectra -- in float32, at the offset's scale: 2.5e-4 on pcg + 100.)
constexpr int kCanonYcGroup = 33;                         // float2 per lane group: za[0..15] | zb[0..15] | pad (the four lane groups of a wave read 264 bytes apart: different LDS banks)
constexpr int kCanonYcFrame = 4 * kCanonYcGroup * 2;      // floats per frame: [lane group][za[0..15] | zb[0..15] | -] as float2 (the lane's two 16-point spectra)
constexpr int kCanonYcRight = 80;                         // right-edge frames tabulated: 0..62 samples to the end, then 17 copies of the interior (a group of 16 frames that
                                                          // reaches into the last 63 columns reads one table whatever its first frame)
// interior [lane group][33] | left edge [lane group][entry][output column 0..63] | right edge [lane group][entry][samples to the end 0..79], float2 each:
// the 16 lanes of a lane group (consecutive frames) read consecutive words of the edge tables
constexpr int kCanonZcFloats = kCanonYcFrame + 4 * 32 * 64 * 2 + 4 * 32 * kCanonYcRight * 2;
constexpr float kMeanTheta = 0.5f;                       // the mean is taken out when S1^2 >= kMeanTheta n E

// Stores the tile's 191 samples (three per lane, sreg[k] = sample lane + 64 k of the aligned tile that starts at output column
// t0, zero outside the signal) as records {x1 | x1 << 16, x2 | x2 << 16} and returns the scales.  The scale exponent comes
// from the tile's energy (max |x| <= sqrt(sum x^2) < 2^hb => |x| 2^(14 - hb) < 2^14 < 65504), which the error bound needs anyway.
// OFFS = false (fsst_team16_kernel, which has no register to spare for the offset term): such a tile is only REPORTED -- mean_s is a
// NaN, nothing is staged -- and the caller hands the whole exec to the kernels that have it.
template <bool OFFS = true>
__device__ __forceinline__ CanonTile canon_land(const float (&sreg)[3], u2* xrec, float r2scale_s, float inv_c, int lane, int t0, int n)
{
    // (samples outside the signal are zeros in sreg: canon_fetch; how many of the tile's 192 slots lie inside it is wave-uniform
    //  arithmetic -- the per-lane count that used to ride through the reduction cost 12 instructions per tile)
    float e2 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { e2 = fmaf(sreg[k], sreg[k], e2); s1 += sreg[k]; }
    const int inside = min(t0 + 128, n) - max(t0 - 64, 0);
    TileEnergy te = tile_energy(e2, s1, 0.0f);           // (E and S1: the reduction of every canonical-band kernel; the count does not ride along)
    te.C = static_cast<float>(max(inside, 0));
    te.dcdom = te.E > 0.0f && te.S1 * te.S1 >= kDcTheta * te.C * te.E;
    // scale, records: one body for both arms below (an offset tile stages x - mean; merged into one path the plain tile paid for the
    // other's copies and selects: ten vector instructions per group of the team kernel)
    auto stage = [&](const float (&x)[3], float Es, float E, float mean) -> CanonTile {
        const int eb = static_cast<int>((__float_as_uint(Es) >> 23) & 0xffu);         // biased exponent (0: zero / denormal tile)
        const int hb = (eb - 127 + 2) >> 1;                                             // sqrt(E) < 2^hb
        const int se = (eb == 0 || eb == 255) ? 127 : 127 + 14 - hb;                    // biased exponent of the sample scale
        const float sx = __uint_as_float(static_cast<unsigned>(se) << 23);
        CanonTile t;
        t.inv = __uint_as_float(static_cast<unsigned>(254 - se) << 23) * inv_c;
        t.R2s = r2scale_s * (E * sx) * sx;
        t.r2s = r2scale_s;
        t.mean_s = mean * sx;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = x[k] * sx;
            const _Float16 x1 = static_cast<_Float16>(v);
            const _Float16 x2 = static_cast<_Float16>(v - static_cast<float>(x1));
            const unsigned b1 = __builtin_bit_cast(unsigned short, x1), b2 = __builtin_bit_cast(unsigned short, x2);
            xrec[lane + 64 * k] = u2{b1 | (b1 << 16), b2 | (b2 << 16)};      // (lane + 128 < kCanonRecs: no predicate)
        }
        wave_sync();
        return t;
    };
    if (__builtin_expect(te.E > 0.0f && te.S1 * te.S1 >= kMeanTheta * te.C * te.E, 0)) {
        if constexpr (!OFFS) { CanonTile r{}; r.mean_s = __builtin_nanf(""); return r; }
        const float mean = te.S1 / te.C;
        float x[3], ea = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gi = t0 + lane + 64 * k - 64;
            x[k] = (gi >= 0 && gi < n) ? sreg[k] - mean : 0.0f;
            ea = fmaf(x[k], x[k], ea);
        }
        const float E = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(piece_sums(ea, 0.0f, 0.0f, 0.0f))));     // (recomputed: E - S1^2 / n cancels)
        // (a tile that is nothing but its mean has E = 0: the scale then comes from the mean, whose spectrum is all there is)
        return stage(x, (E > 0.0f) ? E : te.S1 * mean, E, mean);
    }
    return stage(sreg, te.E, te.E, 0.0f);
}

// One plane.  Rounds 1-4 kept two planes per wave: the own plane, into which every source stores its V, and a displaced plane that
// collected the sources that move (atomic adds), folded into the own plane and cleared after the source stage -- 47 kB of LDS that is
// zero between groups.  The additions now go into the own plane ITSELF.  What the second plane was for is order: a cell must have its
// owner's store -- unconditional, in stripe order -- before anything is added to it, and must not be cleared after.  So, per stripe,
// the rare path (i) decides both classes, (ii) clears the own cells of the sources that leave -- movers and undecided ones alike: an
// undecided source is added by the float64 path, from the float64 V, wherever it belongs --, (iii) adds: at once into rows of stripes
// that are stored already (its own and lower ones: every downward and most upward moves), while a destination in a HIGHER stored
// stripe -- a source in the last row of its stripe moving up -- only sets the lane's bit, and those few sources are redone
// behind the last stored stripe from the spectra (still in registers).  Every cell receives its contributions in program order,
// as before; a cell with two or more of them may differ in the last bit from the two-plane kernels' own + (d1 + d2).
// What the second plane's LDS buys: fsst_team16.hpp.
//
// Decision of a displaced source in float32 (oracle/fsst_oracle.c steps 4-5), as displaced_source of fsst_mfma128.hpp.
struct CanonMove { int row; bool tie; };                 // destination row 0 .. nwin - 1; undecided in float32
__device__ __forceinline__ CanonMove canon_decide(int kpi, float num, float den, float R2, float eoff)
{
    constexpr int NWIN = 128;
    float shift = num * __builtin_amdgcn_rcpf(den);
    if (!(fabsf(shift) <= 1.0e6f)) shift = 0.0f;        // NaN / inf / absurd -> 0 (fsst.m: ~isfinite)
    const float a = static_cast<float>(kpi) + shift;
    float fr = a - floorf(a) - 0.5f;
    const float s1 = 1.0f + fabsf(shift);
    asm volatile("" : "+v"(fr));                        // (see displaced_source: keeps the two product chains unpacked)
    CanonMove m;
#ifndef HSS_NO_TIES
    m.tie = fr * fr * den < (kTieErr2 * kCanonErrMul) * s1 * s1 * fmaf(eoff, den, R2) && den > kTieFloor2 * R2;     // too close to call in float32
#else
    m.tie = false;
#endif
    const float r = truncf(a + copysignf(0.5f, a));     // MATLAB round: half away from zero
    m.row = static_cast<int>(r) & (NWIN - 1);
    return m;
}
// The addition of a source of stripe S that moves to `row` (oracle step 6: the row, or the twin's for a source that wraps around row 0).
// Returns true when the destination lies in a stored stripe ABOVE S -- not stored yet: the caller redoes the source later ("One plane").
// `row0` = this lane's frame row of the plane, indexed by spectrum row.
template <int KLO, int KC, int S>
__device__ __forceinline__ bool canon_put(f2* row0, int kpi, int row, f2 V)
{
    constexpr int NWIN = 128;
    int t = row;
    float vy = V.y;
    if (!(static_cast<unsigned>(row - KLO) < static_cast<unsigned>(KC))) {
        if (!(row > NWIN / 2 && kpi != 0)) return false;
        t = NWIN - row; vy = -V.y;                       // negative-frequency twin: row -> nwin - row, value conj
        if (!(static_cast<unsigned>(t - KLO) < static_cast<unsigned>(KC))) return false;
    }
    if (S < 3 && (t >> 3) > S) return true;              // (the band lies in stripes 0..3)
    float* q = reinterpret_cast<float*>(row0 + t);
    __hip_atomic_fetch_add(q, V.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q + 1, vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return false;
}

// The group's transform up to and including the scatter: on return the own plane [16][LD] holds the group's synchrosqueezed
// rows COV0 .. (scaled by the tile's power of two), displaced cells added, the tie bitmap resolved and cleared.
// xrec = the group's first frame in the tile's records; atab = the shared operand table in LDS; (xsig, n, tg) = the signal and
// the group's first output column, for the float64 tie path.
// (xsig: the signal's samples for the float64 rounding-tie path -- a pointer, or a callable that makes it: a kernel whose signal base is
//  a 64-bit product per group hands over the recipe and pays for it in the rare path only)
template <int KLO, int KC, int TAPB = 4, bool OFFS = true, class XSig = const float*>
__device__ __forceinline__ void canon_group(const u2* xrec, const float* atab, f2* own_base, int* flag, int* tq,
                                            const double* wtab, const double* twtab, const CanonTile& tile, f2 tiny, int lane_o,
                                            XSig xsig, int n, int tg, const float* zc, unsigned long long* cp = nullptr)
{
    using C = CanonCfg<KLO, KC>;
    constexpr int NT = 16, RQ = 8, NWIN = 128;
#ifdef HSS_CANON_PROBE                                   // development: issue-time stamps at the phase boundaries of a group
    unsigned long long cp_last = __builtin_readcyclecounter();
#define CPROBE(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); \
                       if (cp) cp[k] += now_ - cp_last; cp_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPROBE(k) do { } while (0)
#endif
    const int g = (lane_o >> 4) & 3, j = lane_o & 15;    // (& 3: see canon_stats)
    const double* tw_lds = reinterpret_cast<const double*>(atab + kCanonOpFloats);   // the twiddles' copy in LDS (twtab: in HBM)
    (void)twtab;
    // lane (kk = g, f = j) is row-block kk of the B operand for frame f: records f + tap + 16 kk and + 64 (fold terms kk, kk + 4:
    // the 32 lanes of a half-wave then read 32 consecutive records -- every LDS bank once; with terms 2 kk, 2 kk + 1 the two
    // lane groups of a half-wave were 256 bytes apart, on the same banks)
    unsigned xaddr = static_cast<unsigned>(reinterpret_cast<size_t>((lds_u2*)(xrec + j + 16 * g)));
    unsigned aaddr = static_cast<unsigned>(reinterpret_cast<size_t>((lds_u4*)(reinterpret_cast<const u4*>(atab) + lane_o)));
    asm volatile("" : "+v"(xaddr), "+v"(aaddr));
    const lds_u4* ab = (const lds_u4*)static_cast<size_t>(aaddr);
    int pair = g;
    asm volatile("" : "+v"(pair));
    const bool isg0 = (pair == 0);
    const int rAi = pair, rBi = isg0 ? RQ / 2 : RQ - pair;

    f2 za[NT], zb[NT];
    // (TAPB taps at a time: all 16 are independent, and left alone the scheduler loads every operand first -- 128 registers.
    //  A double-buffered variant with every LDS operation issued from inline assembly -- batch k + 1 requested before batch k
    //  is waited for -- was built and measured: 0.2187 vs 0.2157 ms at 16 waves per CU, no difference at 8: LDS latency is
    //  not what this phase waits for.  Round 5 built it again for the team kernel, s_waitcnt lgkmcnt(4) between the batches:
    //  0.1733 vs 0.1713 / 0.1768 ms in one run, 2898 vs 2918 wave quad-cycles per group -- inside the noise, not kept.)
    static_for<NT / TAPB>([&](auto GG) {
        constexpr int g0 = decltype(GG)::value * TAPB;
        u4 a[TAPB], b[TAPB];
        static_for<TAPB>([&](auto I) {
            constexpr int n = g0 + decltype(I)::value;
            // records f + n + 16 kk and + 64 in ONE instruction and one register quad (left to itself the compiler pairs
            // record n with n + 1 -- adjacent addresses -- and then shuffles six registers per two taps)
            asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(b[n - g0]) : "v"(xaddr), "n"(n), "n"(n + 64) : "memory");
            a[n - g0] = ab[n * 64];
        });
        // (the compiler does not count LDS operations issued from inline assembly: wait for them here; its own counts for
        //  the A operands only become more conservative, LDS returns in order)
        if constexpr (TAPB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
        else if constexpr (TAPB == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]) :: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]) :: "memory");
        static_for<TAPB>([&](auto I) {
            constexpr int n = g0 + decltype(I)::value;
            const f4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[n - g0]), __builtin_bit_cast(h8, b[n - g0]),
                                                                  f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
            za[bitrev_n<NT>(n)] = f2{acc.x, acc.y};
            zb[bitrev_n<NT>(n)] = f2{acc.z, acc.w};
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    CPROBE(0);
#if !defined(HSS_CANON_ABLATE) || HSS_CANON_ABLATE < 5
    fft_n<NT>(za);
    fft_n<NT>(zb);
#endif
    CPROBE(1);
    // ("Offsets" above) a tile staged without its mean: + mean x the spectrum of the all-ones frame, ONE block between the spectra and
    // the source stage under one wave-uniform branch (round 4 added it to the mixed pairs of every stripe under eight branches, which
    // cost the plain path of the team kernel 4 %).  Interior groups: 32 constants per lane group from the first frame's copy in LDS; a group
    // at an end of the signal: left + right - interior of the per-frame tables in global memory (8 of a 2000-sample signal's 125 groups).
    if constexpr (OFFS) {
        // (two sibling blocks, not one with two arms: the edge block's loads in flight made the register allocator spill two dozen of the
        //  spectra's registers AROUND THE WHOLE offset block -- every interior group of an offset tile, 117 of a signal's 125, paid 48
        //  scratch operations, 3 GB of scratch traffic per 1024 windows, most of what such a tile cost)
        const bool off_interior = tile.mean_s != 0.0f && tg >= 64 && tg + 15 + 63 <= n - 1;      // (wave-uniform)
        const bool off_edge = tile.mean_s != 0.0f && !off_interior;
        if (__builtin_expect(off_interior, 0)) {         // interior: the constants' copy in LDS (broadcast reads)
            int lane_f = lane_o;
            asm volatile("" : "+v"(lane_f));
            const int gq = (lane_f >> 4) & 3;
            const f2 mm = {tile.mean_s, tile.mean_s};
            const f2* zl = reinterpret_cast<const f2*>(atab + kCanonAtabFloats) + gq * kCanonYcGroup;
            static_for<NT>([&](auto I) {
                constexpr int i = decltype(I)::value;
                za[i] = pk_fma(zl[i], mm, za[i]);
                zb[i] = pk_fma(zl[NT + i], mm, zb[i]);
            });
        }
        if (__builtin_expect(off_edge, 0)) {
            int lane_f = lane_o;
            asm volatile("" : "+v"(lane_f));
            const int gq = (lane_f >> 4) & 3;
            const f2 mm = {tile.mean_s, tile.mean_s};
            const f2* zg = reinterpret_cast<const f2*>(zc) + gq * kCanonYcGroup;
            {
                // a group at an end of the signal: the frame's own constants -- left-edge table by output column, right-edge table by samples
                // to the end (both, less the interior, for a signal shorter than a window and a half).  Eight entries at a time: left alone
                // the scheduler requests all of them first and spills the spectra.
                const int tf = tg + (lane_f & 15);
                const int rr = min(max(n - 1 - tf, 0), kCanonYcRight - 1);       // (a frame behind the signal's end belongs to no output column)
                const bool left = tg < 64, both = left && tg + 15 + 63 > n - 1;      // (wave-uniform)
                // scalar table base + one lane offset per table: every load is "saddr + voffset" (per-lane 64-bit pointers beside the 64
                // registers of the spectra were what spilled)
                const char* lbase = reinterpret_cast<const char*>(zc + kCanonYcFrame);
                const char* rbase = lbase + 4 * 32 * 64 * 8;
                const unsigned lofs = static_cast<unsigned>(gq * (32 * 64) + min(tf, 63)) * 8u, rofs = static_cast<unsigned>(gq * (32 * kCanonYcRight) + rr) * 8u;
                const char* b1 = left ? lbase : rbase;
                const unsigned o1 = left ? lofs : rofs, st1 = (left ? 64u : static_cast<unsigned>(kCanonYcRight)) * 8u;
                auto ent1 = [&](int e) -> f2 { return *reinterpret_cast<const f2*>(b1 + static_cast<size_t>(e) * st1 + o1); };
                auto ent2 = [&](int e) -> f2 { return *reinterpret_cast<const f2*>(rbase + static_cast<size_t>(e) * (kCanonYcRight * 8) + rofs); };
                if (!both) {
                    static_for<8>([&](auto Q) {
                        constexpr int q0 = 2 * decltype(Q)::value;
                        const f2 c0 = ent1(q0), c1 = ent1(q0 + 1), c2 = ent1(NT + q0), c3 = ent1(NT + q0 + 1);
                        za[q0] = pk_fma(c0, mm, za[q0]); za[q0 + 1] = pk_fma(c1, mm, za[q0 + 1]);
                        zb[q0] = pk_fma(c2, mm, zb[q0]); zb[q0 + 1] = pk_fma(c3, mm, zb[q0 + 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else {
                    static_for<NT>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        const f2 ca = ent1(i) + ent2(i) - zg[i], cb = ent1(NT + i) + ent2(NT + i) - zg[NT + i];
                        za[i] = pk_fma(ca, mm, za[i]); zb[i] = pk_fma(cb, mm, zb[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            }
            // (this block's loads and register RELOADS are back before the paths merge -- the allocator spills the spectra around this block
            //  and reloads them at its very end; left pending, they make the wait-count pass put an s_waitcnt vmcnt(0) behind the merge:
            //  on the plain path too, in the middle of every transform, where it waits for whatever the caller has in flight -- the next
            //  tile's samples, the previous group's stores.  The empty statements make the reloads happen in front of the wait.)
            asm volatile("" : "+v"(za[0]), "+v"(za[1]), "+v"(za[2]), "+v"(za[3]), "+v"(za[4]), "+v"(za[5]), "+v"(za[6]), "+v"(za[7]),
                              "+v"(za[8]), "+v"(za[9]), "+v"(za[10]), "+v"(za[11]), "+v"(za[12]), "+v"(za[13]), "+v"(za[14]), "+v"(za[15]));
            asm volatile("" : "+v"(zb[0]), "+v"(zb[1]), "+v"(zb[2]), "+v"(zb[3]), "+v"(zb[4]), "+v"(zb[5]), "+v"(zb[6]), "+v"(zb[7]),
                              "+v"(zb[8]), "+v"(zb[9]), "+v"(zb[10]), "+v"(zb[11]), "+v"(zb[12]), "+v"(zb[13]), "+v"(zb[14]), "+v"(zb[15]));
            HSS_RARE_VMEM_DONE();
        }
    }

    // Conjugate partners.  Source s of class r pairs with bin nwin - (r + 8 s): class 8 - r, index 15 - s -- the OTHER array of the lane
    // -- except in lane group 0, whose two classes are their own partners: class 0 pairs with za[(16 - s) & 15], class 4 with zb[15 - s].
    // Selecting per stripe cost four v_cndmask each (32 per group); instead lane group 0 alone (exec = lanes 0..15) rotates the upper
    // halves of its arrays once, zb[m] <- za[m + 1], za[m] <- zb[m] (m = 8..15; the indices 8..15 are partners only, never sources), and
    // every lane reads PA = zb[15 - s], PB = za[15 - s]: 17 moves of a register pair.
    {
        f2 t0;
        asm volatile("s_mov_b64 exec, 0xffff\n\t"
                     "v_mov_b64 %[t], %[b8]\n\t"
                     "v_mov_b64 %[b8], %[a9]\n\tv_mov_b64 %[a9], %[b9]\n\t"
                     "v_mov_b64 %[b9], %[a10]\n\tv_mov_b64 %[a10], %[b10]\n\t"
                     "v_mov_b64 %[b10], %[a11]\n\tv_mov_b64 %[a11], %[b11]\n\t"
                     "v_mov_b64 %[b11], %[a12]\n\tv_mov_b64 %[a12], %[b12]\n\t"
                     "v_mov_b64 %[b12], %[a13]\n\tv_mov_b64 %[a13], %[b13]\n\t"
                     "v_mov_b64 %[b13], %[a14]\n\tv_mov_b64 %[a14], %[b14]\n\t"
                     "v_mov_b64 %[b14], %[a15]\n\tv_mov_b64 %[a15], %[b15]\n\t"
                     "v_mov_b64 %[b15], %[a0]\n\t"
                     "v_mov_b64 %[a8], %[t]\n\t"
                     "s_mov_b64 exec, -1"
                     : [t] "=&v"(t0), [a8] "+v"(za[8]), [a9] "+v"(za[9]), [a10] "+v"(za[10]), [a11] "+v"(za[11]), [a12] "+v"(za[12]),
                       [a13] "+v"(za[13]), [a14] "+v"(za[14]), [a15] "+v"(za[15]), [b8] "+v"(zb[8]), [b9] "+v"(zb[9]), [b10] "+v"(zb[10]),
                       [b11] "+v"(zb[11]), [b12] "+v"(zb[12]), [b13] "+v"(zb[13]), [b14] "+v"(zb[14]), [b15] "+v"(zb[15])
                     : [a0] "v"(za[0]));
    }

    // own-plane columns of this lane's two classes as ONE opaque byte address each: stripe s is then the immediate + 64 s
    unsigned oa = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)reinterpret_cast<float*>(own_base + j * C::LD + rAi - C::COV0)));
    unsigned ob = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)reinterpret_cast<float*>(own_base + j * C::LD + rBi - C::COV0)));
    asm volatile("" : "+v"(oa), "+v"(ob));
    lds_float* ownA = (lds_float*)static_cast<size_t>(oa);
    lds_float* ownB = (lds_float*)static_cast<size_t>(ob);
    f2* row0 = own_base + j * C::LD - C::COV0;           // this lane's frame row of the plane, indexed by spectrum row (rare path)
    unsigned defer = 0u;                                 // bit 2 s + class: a source of stripe s <= 2 whose destination was not stored yet ("One plane")
    float mx = 0.0f;                                     // largest |V|^2 among this lane's stored cells ("Exact groups")
    bool visited = false;                                // (wave-uniform) some stripe took the rare path: only then are the flags in LDS worth a look
#if defined(HSS_CANON_ABLATE) && HSS_CANON_ABLATE >= 4
    {   f2 accz = {0.0f, 0.0f};
        static_for<NT>([&](auto I) { accz += za[decltype(I)::value] + zb[decltype(I)::value]; });
        ownA[0] = accz.x; ownA[1] = accz.y; mx = 1.0e30f; }
    static_for<0>([&](auto SS) {
#else
    static_for<NT / 2>([&](auto SS) {
#endif
        constexpr int s = decltype(SS)::value;
        constexpr bool STA = C::stored(s, 0), STB = C::stored(s, 1);
        constexpr float TA = C::thr(s, 0), TB = C::thr(s, 1);
        const f2 PA = zb[NT - 1 - s], PB = za[NT - 1 - s];
        f2 a1 = mix_re(za[s], PA), a2 = mix_im(za[s], PA);
        f2 b1 = mix_re(zb[s], PB), b2 = mix_im(zb[s], PB);
        const f2 dna = dn_second(a2, dn_first(a1, tiny)), dnb = dn_second(b2, dn_first(b1, tiny));
        if constexpr (STA) { ownA[16 * s] = a1.x; ownA[16 * s + 1] = a2.x; mx = fmaxf(mx, dna.x); }
        if constexpr (STB) { ownB[16 * s] = b1.x; ownB[16 * s + 1] = b2.x; mx = fmaxf(mx, dnb.x); }
        const bool ma = fabsf(dna.y) >= TA * dna.x, mb = fabsf(dnb.y) >= TB * dnb.x;
#if defined(HSS_CANON_ABLATE) && HSS_CANON_ABLATE >= 3      // development (tools/canon_ablate.sh): results invalid
        if ((ma | mb) && tile.R2s == 123.0f) {
#else
        // (unlikely: the rare path is laid out behind the hot code, no taken branch over it; the test is the wave's, not the lane's, so that
        //  `visited` is a scalar)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(ma | mb) != 0ull, 0)) {
            visited = true;
#endif
            f2* cellA = reinterpret_cast<f2*>((float*)(ownA + 16 * s));
            f2* cellB = reinterpret_cast<f2*>((float*)(ownB + 16 * s));
            const int kA = rAi + RQ * s, kB = rBi + RQ * s;
            unsigned* tb = reinterpret_cast<unsigned*>(tq);
            // ("One plane") (i) decide both classes ...
            CanonMove da{kA, false}, db{kB, false};
            if (ma) da = canon_decide(kA, dna.y, dna.x, tile.R2s, tile.eoff());
            if (mb) db = canon_decide(kB, dnb.y, dnb.x, tile.R2s, tile.eoff());
            const bool la = ma && (da.tie || da.row != kA), lb = mb && (db.tie || db.row != kB);      // leaves its own cell
            // ... (ii) undecided sources set their bits, every source that leaves clears its cell ...
            if (ma && da.tie) { __hip_atomic_fetch_or(tb + (kA >> 1), 1u << (((kA & 1) << 4) + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); flag[1] = 1; }
            if (mb && db.tie) { __hip_atomic_fetch_or(tb + (kB >> 1), 1u << (((kB & 1) << 4) + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); flag[1] = 1; }
            if constexpr (STA) { if (la) *cellA = f2{0.0f, 0.0f}; }
            if constexpr (STB) { if (lb) *cellB = f2{0.0f, 0.0f}; }
            // ... (iii) and the movers add
            if (la && !da.tie) { if (canon_put<KLO, KC, s>(row0, kA, da.row, f2{a1.x, a2.x})) defer |= 1u << (2 * s); }
            if (lb && !db.tie) { if (canon_put<KLO, KC, s>(row0, kB, db.row, f2{b1.x, b2.x})) defer |= 1u << (2 * s + 1); }
        }
        if constexpr (s == 3) {
            // ("One plane") the sources of stripes 0..2 whose destination lay in a stripe that had not been stored yet: redone from the
            // spectra, decision and all (the same arithmetic on the same numbers), now that every stored stripe is there
            if (__builtin_expect(visited && __builtin_amdgcn_ballot_w64(defer != 0u) != 0ull, 0)) {
                static_for<3>([&](auto TT) {
                    constexpr int t = decltype(TT)::value;
                    if (__builtin_amdgcn_ballot_w64((defer >> (2 * t)) & 3u) != 0ull) {
                        const f2 QA = zb[NT - 1 - t], QB = za[NT - 1 - t];
                        const f2 c1 = mix_re(za[t], QA), c2 = mix_im(za[t], QA), e1 = mix_re(zb[t], QB), e2 = mix_im(zb[t], QB);
                        const f2 dc = dn_second(c2, dn_first(c1, tiny)), de = dn_second(e2, dn_first(e1, tiny));
                        if ((defer >> (2 * t)) & 1u) {
                            const CanonMove d = canon_decide(rAi + RQ * t, dc.y, dc.x, tile.R2s, tile.eoff());
                            (void)canon_put<KLO, KC, 3>(row0, rAi + RQ * t, d.row, f2{c1.x, c2.x});
                        }
                        if ((defer >> (2 * t + 1)) & 1u) {
                            const CanonMove d = canon_decide(rBi + RQ * t, de.y, de.x, tile.R2s, tile.eoff());
                            (void)canon_put<KLO, KC, 3>(row0, rBi + RQ * t, d.row, f2{e1.x, e2.x});
                        }
                    }
                });
            }
        }
    });
    CPROBE(2);
    wave_sync();
    // the tie flag is set by the rare path only: a group that never went there skips the trip to LDS and its wait
    int f_ties = 0;
    if (visited) f_ties = flag[1];
    asm volatile("" : "+v"(f_ties));
    auto signal_sample = [&](int i) -> double {
        const int gi = tg + i - NWIN / 2;
        const float* xs;
        if constexpr (__is_pointer(XSig)) xs = xsig; else xs = xsig();
        return (gi >= 0 && gi < n) ? static_cast<double>(xs[gi]) : 0.0;
    };
    bool exact = false;
#ifndef HSS_NO_EXACT
    // ---- "Exact groups" (fsst_mfma128.hpp): no stored cell reaches kExactTheta R of the tile -> look again with the R of
    //      the group's own 143 samples (a quiet group beside a loud burst) -> still none: float64 for the whole group
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx > kExactTheta2 * tile.R2s) == 0ull && tile.R2s > 0.0f, 0)) {
        float e2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = lane_o + 64 * k;
            const u2 r = xrec[min(i, 142)];
            const float v = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(r.x & 0xffffu))) +
                            static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(r.y & 0xffffu)));
            e2 = fmaf(i < 143 ? v : 0.0f, v, e2);
        }
        const float R2g = tile.r2s * __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(piece_sums(e2, 0.0f, 0.0f, 0.0f))));
        exact = __builtin_amdgcn_ballot_w64(mx > kExactTheta2 * R2g) == 0ull && R2g > 0.0f;
    }
#endif
    if (__builtin_expect(exact, 0)) {
        // (the float32 contributions are dropped: every cell of the group comes from the float64 pass)
        for (int i = lane_o; i < 16 * C::LD; i += 64) own_base[i] = f2{0.0f, 0.0f};
        if (lane_o == 0) flag[1] = 0;
        wave_sync();
        resolve_bitmap<NWIN, true, true, true>(reinterpret_cast<unsigned*>(tq), signal_sample, own_base, C::LD, flag, KLO, KC, own_base, C::LD,
                                               C::COV0, C::COV0 + C::COVN, wtab, tw_lds, 1.0 / static_cast<double>(tile.inv), lane_o);
        wave_sync();
        HSS_RARE_VMEM_DONE();
    } else
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(f_ties) != 0, 0)) {       // (rare) cells whose rounding float32 cannot decide
        // the float64 DFT reads the signal itself (HBM / L2): the records hold 22 bits of a sample, and a coordinate that is
        // 1e-5 bins from a half-integer needs all 24
        resolve_bitmap<NWIN, false, true, true>(reinterpret_cast<unsigned*>(tq), signal_sample, own_base, C::LD, flag, KLO, KC, own_base, C::LD, C::COV0, C::COV0 + C::COVN, wtab, tw_lds, 1.0 / static_cast<double>(tile.inv), lane_o);
        wave_sync();
        HSS_RARE_VMEM_DONE();
    }
    CPROBE(3);
#undef CPROBE
}
// Statistics partial of the group in the own plane (see "Statistics" in fsst_kernels.hpp): pivoted sums over the kept cells
// of the nvalid valid frames, computed on the SCALED plane values and scaled back afterwards -- a power of two, so the
// result is the one the unscaled cells would give.  Returns piece_sums' w (row q of the wave: S1re / S2re / S1im / S2im)
// and the pivot, both in feature units.
template <int KLO, int KC>
__device__ __forceinline__ float canon_stats(const f2* own_base, int nvalid, float inv, int lane_o, f2& piv_out)
{
    using C = CanonCfg<KLO, KC>;
    // (& 3: the compiler cannot see through the opaque lane copy that g < 4 -- without it every "g + 4 u < KC" below is a
    //  compare and two selects, with it only the one row group that is partial across lanes)
    const int g = (lane_o >> 4) & 3, j = lane_o & 15;
    const f2* src = own_base + j * C::LD + C::KOFF + g;
    // (pivot: median of frame 0's first, middle and last kept row -- pivot_med3, fsst_kernels.hpp; three broadcast reads)
    const f2 pv0 = own_base[C::KOFF], pv1 = own_base[C::KOFF + KC / 2], pv2 = own_base[C::KOFF + KC - 1];
    const f2 piv = f2{pivot_med3(pv0.x, pv1.x, pv2.x), pivot_med3(pv0.y, pv1.y, pv2.y)};
    constexpr int NU = (KC + 3) / 4, UF = KC / 4;        // rows g + 4 u: u < UF valid in every lane group, u == UF for g < KC - 4 UF
    f2 v[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) v[u] = src[4 * u];
    f2 st_s = {0.0f, 0.0f}, st_q = {0.0f, 0.0f};
    if (__builtin_expect(nvalid == 16, 1)) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f2 d = v[u] - piv;
            if (u >= UF) { const bool ok = g + 4 * u < KC; d = f2{ok ? d.x : 0.0f, ok ? d.y : 0.0f}; }
            st_s += d; st_q = pk_fma(d, d, st_q);
        }
    } else {
        asm volatile("");                                // (keeps the two arms apart: merged, the full group pays for the ragged one's selects)
        const bool jv = j < nvalid;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f2 d = v[u] - piv;
            const bool ok = jv && (g + 4 * u < KC);
            d = f2{ok ? d.x : 0.0f, ok ? d.y : 0.0f};
            st_s += d; st_q = pk_fma(d, d, st_q);
        }
    }
    float w = piece_sums(st_s.x, st_q.x, st_s.y, st_q.y);
    w *= inv;                                            // (the rows of squares: twice -- inv x inv itself leaves float32 for tiny signals)
    if ((lane_o >> 4) & 1) w *= inv;
    piv_out = piv * f2{inv, inv};
    return w;
}

// The group's image [16][2 KC] as this lane's three lane-linear float4s (float4 number lane + 64 i), in feature units.
// ppk = the wide-store offset table in LDS ([3][64] words, two 16-bit byte offsets each: canon_store_offsets).
template <int KLO, int KC, class Sink>
__device__ __forceinline__ void canon_image_to(const f2* own_base, const unsigned* ppk, float inv, int lane_o, Sink sink)
{
    unsigned obase = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)reinterpret_cast<const float*>(own_base)));
    asm volatile("" : "+s"(obase));                      // ONE scalar base: each cell is then "offset + base", no second add
    const f2 sc = {inv, inv};
    static_for<3>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const unsigned pk = ppk[i * 64 + lane_o];
        const lds_float* q0 = (const lds_float*)static_cast<size_t>(obase + (pk & 0xffffu));
        const lds_float* q1 = (const lds_float*)static_cast<size_t>(obase + (pk >> 16));
        const f2 lo = f2{q0[0], q0[2]} * sc, hi = f2{q1[0], q1[2]} * sc;
        sink(i, f4{lo.x, lo.y, hi.x, hi.y});
    });
}
// (fsst_team16_kernel gathers the image itself, in one batch with the words of the group that leaves: fsst_team16.hpp)
template <int KLO, int KC>
__device__ __forceinline__ void canon_image(const f2* own_base, const unsigned* ppk, float inv, int lane_o, f4 (&o)[3])
{
    canon_image_to<KLO, KC>(own_base, ppk, inv, lane_o, [&](int i, f4 v) { o[i] = v; });
}

// Byte offsets, inside the own plane, of the two (re, re) / (im, im) pairs of float4 number f = lane + 64 i of a group's
// contiguous [16][2 KC] image; packed lo | hi << 16.  (A pair's second element is the next cell: + 8 bytes.)
template <int KLO, int KC>
__device__ __forceinline__ unsigned canon_store_offsets(int f)
{
    using C = CanonCfg<KLO, KC>;
    constexpr int Q = KC / 2;
    const int jj = min(f / Q, 15), c = 4 * (f - (f / Q) * Q);
    const int rowb = jj * C::LD + C::KOFF;
    const unsigned p0 = (c < KC) ? (rowb + c) * 8 : (rowb + c - KC) * 8 + 4;
    const unsigned p1 = (c + 2 < KC) ? (rowb + c + 2) * 8 : (rowb + c + 2 - KC) * 8 + 4;
    return p0 | (p1 << 16);
}

struct CanonParams {
    const float* x;       // [nsig][xstride]
    float* out;           // [nsig][ncols][2 KC]
    float* partials;      // two-launch path: [nsig][groups][kPartFloats]
    const float* atab;    // f16 operand table (kCanonAtabFloats floats), then the offset table (kCanonZcFloats floats)
    const double* wtab;   // float64 {w, dw'}[128]          } rounding-tie path
    const double* twtab;  // float64 {cos, sin}(2 pi m / 128) }
    float r2scale_s;      // r2scale of the plan x (constant scale)^2
    float inv_c;          // 1 / constant scale
    int n, mode, nsig, col0, ncols;
    long long xstride;
    Core128Regions reg;
    unsigned* status;     // FUSED: device status word
    const unsigned* gate; // non-null: this launch is the fallback of a team-kernel exec: it runs only if *gate == gate_val
    unsigned gate_val;
};

constexpr int kCanonCtlFloats = 16 + 192;                            // [0] work counter, [16..207] wide-store offsets
constexpr int kCanonCtlFusedFloats = kCtlFusedFloats;                // the layout of fsst_core128_kernel<FUSED>

// The three samples per lane of the aligned tile that starts at output column t0 (xpad index t0 + i <-> sample t0 + i - 64),
// zero outside the signal.  Buffer loads: the signal is a raw buffer of n floats, an offset outside it (negative ones are huge
// as unsigned) returns 0 -- the zero padding of oracle/fsst_oracle.c step 1 done by the address unit, no compare / exec-mask /
// 64-bit address per sample.
__device__ __forceinline__ void canon_fetch(const float* xsig, int n, int t0, int lane_o, float (&sreg)[3])
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsig), 0, n * 4, 0x00020000);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        sreg[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (t0 + lane_o + 64 * k - 64) * 4, 0, 0));
}

// ------------------------------------------------------------------------------------------------
// fsst_canon_kernel: work distribution, tickets and the FUSED z-score exactly as fsst_core128_kernel (see there:
// "Work distribution", "Fused z-score"); the body of a transform chunk is canon_group + canon_stats + canon_image.
// ------------------------------------------------------------------------------------------------
#ifdef HSS_FUSE_PROBE      // development (tools/fuse_probe2.py): shader-clock totals per ticket kind over all waves of the FUSED kernel
__device__ unsigned long long g_fuse_probe[8];      // [0] A tickets [1] cycles in A [2] B tickets [3] B: wait for statistics
                                                    // [4] B: loads issued -> data there [5] B: arithmetic + stores issued [6] resolver
#endif
template <int KLO, int KC, bool FUSED>
__global__ __launch_bounds__(64 * 16, HSS_MW128) void fsst_canon_kernel(CanonParams p)
{
    using C = CanonCfg<KLO, KC>;
    constexpr int WPB = 16, K = KC, GPCF = kCanonTileFrames / 16;
    constexpr int ATAB = kCanonLdsTabFloats;
    constexpr int CTL = FUSED ? kCanonCtlFusedFloats : kCanonCtlFloats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (p.gate != nullptr && *p.gate != p.gate_val) return;          // (uniform: the team kernel this launch backs up did not give up)
    const int n = p.n;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* atab = smem;
    int* next_q = reinterpret_cast<int*>(smem + ATAB);
    unsigned* done_a = reinterpret_cast<unsigned*>(smem + ATAB) + 1;
    unsigned* dead = done_a + 2;
    unsigned* ready = dead + 1;
    float4* fin_stats = reinterpret_cast<float4*>(smem + ATAB + 272);
    unsigned* cls_lds = reinterpret_cast<unsigned*>(smem + ATAB + 16);
    unsigned* ppk_lds = reinterpret_cast<unsigned*>(smem + ATAB + (FUSED ? 80 : 16));
    float* part_lds = smem + ATAB + 288;
    float* wbase = smem + ATAB + CTL + wv * C::wave_floats();
    u2* xrec = reinterpret_cast<u2*>(wbase);
    f2* own_base = reinterpret_cast<f2*>(wbase + 2 * kCanonRecs);
    int* flag = reinterpret_cast<int*>(own_base + 16 * C::LD);
    int* tq = flag + kCanonFlagWords;

    for (int i = threadIdx.x; i < ATAB; i += 64 * WPB) atab[i] = p.atab[i];
    const int ncols = p.ncols, cend = p.col0 + p.ncols;
    if (lane < kCanonFlagWords) flag[lane] = 0;
    if (lane < kCanonTieWords) tq[lane] = 0;
    if (threadIdx.x < (FUSED ? 8 : 1)) next_q[threadIdx.x] = 0;
    if (wv == 0) {
        unsigned cls = 0u;                               // bit 2i / 2i+1: the first / second pair of float4 i is imaginary
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned c = (4u * static_cast<unsigned>(lane + 64 * i)) % static_cast<unsigned>(2 * K);
            cls |= (c >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i);
            cls |= (c + 2 >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i + 1);
            ppk_lds[i * 64 + lane] = canon_store_offsets<KLO, KC>(lane + 64 * i);
        }
        if constexpr (FUSED) cls_lds[lane] = cls;
    }
    __syncthreads();

    const int nc0 = p.nsig * p.reg.npc[0];
    const int nc1 = nc0 + p.nsig * p.reg.npc[1];
    const int nchunks = nc1 + p.nsig * p.reg.npc[2];
    const int ngroups = (ncols + 15) >> 4;
    const int nk = (FUSED && p.nsig > static_cast<int>(blockIdx.x)) ? (p.nsig - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x) : 0;
    const int NC = (ngroups + GPCF - 1) / GPCF;
    const int lead = min(8, NC);
    const int nwork = FUSED ? 2 * NC * nk : nchunks;
    auto draw = [&]() -> int {
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = __builtin_amdgcn_readfirstlane(q);
        if constexpr (FUSED) return q < nwork ? q : nwork;
        const long long c = static_cast<long long>(blockIdx.x) + static_cast<long long>(q) * gridDim.x;
        return c < nwork ? static_cast<int>(c) : nwork;
    };
    int chunk = draw();
    f2 tiny = {1.0e-37f, 0.0f};
    asm volatile("" : "+s"(tiny));
#ifdef HSS_FUSE_PROBE
    unsigned long long fp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FPROBE_NOW() __builtin_readcyclecounter()
#endif
    while (chunk < nwork) {
#ifdef HSS_FUSE_PROBE
    const unsigned long long fp_t0 = FPROBE_NOW();
#endif
    long long b;
    int grp0, ngrp;
    long long ksig = 0;
    bool zpass = false;
    if constexpr (FUSED) {
        int c;
        if (chunk < NC) {
            ksig = 0; c = chunk;
        } else {
            const int u = chunk - NC;
            const int kk = 1 + u / (2 * NC), v = u - (kk - 1) * (2 * NC);
            if (kk < nk) {
                const int npairs = NC - lead;
                if (v < lead) { ksig = kk; c = v; }
                else if (v - lead < 2 * npairs) {
                    const int w = v - lead;
                    if (w & 1) { ksig = kk; c = lead + (w >> 1); }
                    else { ksig = kk - 1; c = w >> 1; zpass = true; }
                } else { ksig = kk - 1; c = npairs + (v - lead - 2 * npairs); zpass = true; }
            } else { ksig = nk - 1; c = v; zpass = true; }
        }
        b = static_cast<long long>(blockIdx.x) + ksig * gridDim.x;
        grp0 = c * GPCF;
        ngrp = min(GPCF, ngroups - grp0);
    } else {
        const int rg = (chunk < nc0) ? 0 : (chunk < nc1) ? 1 : 2;
        const int local = chunk - ((rg == 0) ? 0 : (rg == 1) ? nc0 : nc1);
        const int npc = p.reg.npc[rg], gpc = p.reg.gpc[rg];
        b = local / npc;
        const int cidx = local - static_cast<int>(b) * npc;
        grp0 = p.reg.g0[rg] + cidx * gpc;
        ngrp = min(gpc, ngroups - grp0);
    }
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    if (FUSED && zpass) {
        // ---- B(ksig, c): z-score of ngrp groups of a signal whose statistics are (about to be) in LDS
        const int sl = static_cast<int>(ksig) & 3;
        const unsigned epoch = static_cast<unsigned>(ksig) + 1u;
        constexpr int CC = 2 * K;
        float4* d4 = reinterpret_cast<float4*>(p.out + (b * static_cast<long long>(ncols) + grp0 * 16) * CC) + lane_o;
        constexpr int per = 8 * K;
        for (unsigned spins = 0;; ++spins) {
            unsigned have = 0;
            if (lane == 0) have = __hip_atomic_load(ready + sl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(have) == epoch) break;
            if (spins >= kSpinLimit || __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                if (lane == 0) {
                    __hip_atomic_store((gu32*)(p.status), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef HSS_FUSE_PROBE
        const unsigned long long fp_t1 = FPROBE_NOW();
#endif
        const float4 st = fin_stats[sl];
        const unsigned cls = cls_lds[lane_o];
        const float4* base0 = reinterpret_cast<const float4*>(p.out + (b * static_cast<long long>(ncols) + grp0 * 16) * CC);
        auto glim = [&](int q) { return (q < ngrp) ? min(16, ncols - (grp0 + q) * 16) * (K >> 1) : 0; };
        {
            f4 o[GPCF][3];
#pragma unroll
            for (int q = 0; q < GPCF; ++q) {
                const int lim = glim(q);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float4* src = (lane_o + 64 * i < lim) ? (d4 + q * per + 64 * i) : base0;
                    o[q][i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src));
                }
            }
            f2 mu[3][2], rs[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bool im0 = (cls >> (2 * i)) & 1u, im1 = (cls >> (2 * i + 1)) & 1u;
                const float m0 = im0 ? st.z : st.x, r0 = im0 ? st.w : st.y;
                const float m1 = im1 ? st.z : st.x, r1 = im1 ? st.w : st.y;
                mu[i][0] = f2{m0, m0}; rs[i][0] = f2{r0, r0};
                mu[i][1] = f2{m1, m1}; rs[i][1] = f2{r1, r1};
            }
#ifdef HSS_FUSE_PROBE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long fp_t2 = FPROBE_NOW();
#endif
#pragma unroll
            for (int q = 0; q < GPCF; ++q) {
                const int lim = glim(q);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (lane_o + 64 * i < lim) {
                        const f2 lo = (f2{o[q][i].x, o[q][i].y} - mu[i][0]) * rs[i][0];
                        const f2 hi = (f2{o[q][i].z, o[q][i].w} - mu[i][1]) * rs[i][1];
                        __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(d4 + q * per + 64 * i));
                    }
                }
            }
#ifdef HSS_FUSE_PROBE
            const unsigned long long fp_t3 = FPROBE_NOW();
            fp[2] += 1; fp[3] += fp_t1 - fp_t0; fp[4] += fp_t2 - fp_t1; fp[5] += fp_t3 - fp_t2;
#endif
        }
    } else {
    const float* xsig = p.x + b * p.xstride;
    // per-signal bases once per ticket (uniform): the group loop then adds 32-bit offsets (was: a 64-bit multiply per group)
    float* out_sig = p.out + b * static_cast<long long>(ncols) * (2 * K);
    float* part_sig = FUSED ? nullptr : p.partials + b * static_cast<long long>(ngroups) * kPartFloats;
    const int cg0 = p.col0 >> 4;                         // (the host sends only column ranges that start on a group boundary)
    for (int gcur = grp0; gcur < grp0 + ngrp;) {
        // the ALIGNED tile of the signal that holds group gcur: aligned in absolute columns, so that a column-range exec
        // stages the very tiles -- and scales -- of the whole-signal transform
        const int tbase = (gcur + cg0) & ~(GPCF - 1);    // absolute group index of the tile's first group
        const int gstop = min(tbase + GPCF - cg0, grp0 + ngrp);
        float sreg[3];
        int lane_t = lane;                               // opaque per tile / per group: nothing derived from the lane id is
        asm volatile("" : "+v"(lane_t));                 // hoisted out of these loops, held across the transform and spilled
        canon_fetch(xsig, n, tbase * 16, lane_t, sreg);
        const CanonTile tile = canon_land(sreg, xrec, p.r2scale_s, p.inv_c, lane_t, tbase * 16, n);
        for (int gidx = gcur; gidx < gstop; ++gidx) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int tg = p.col0 + gidx * 16;
            canon_group<KLO, KC>(xrec + (gidx + cg0 - tbase) * 16, atab, own_base, flag, tq, p.wtab, p.twtab, tile, tiny, lane_o, xsig, n, tg, p.atab + kCanonAtabFloats);
            const int nvalid = min(16, cend - tg);
#if defined(HSS_CANON_ABLATE) && HSS_CANON_ABLATE >= 2
            if (p.mode == 77) {
#else
            if (p.mode == kModeStack) {
#endif
                f2 piv;
                const float w = canon_stats<KLO, KC>(own_base, nvalid, tile.inv, lane_o, piv);
                if constexpr (FUSED) store_partial(part_lds + ((static_cast<int>(ksig) & 1) * kFusedMaxGroups + gidx) * kPartFloats, w, piv.x, piv.y);
                else store_partial(part_sig + gidx * kPartFloats, w, piv.x, piv.y);
            }
#if defined(HSS_CANON_ABLATE) && HSS_CANON_ABLATE >= 1
            if (tg == 123456789) {
#else
            {
#endif
            f4 o[3];
            canon_image<KLO, KC>(own_base, ppk_lds, tile.inv, lane_o, o);
            float4* dst4 = reinterpret_cast<float4*>(out_sig + (tg - p.col0) * (2 * K)) + lane_o;
            const int lim = nvalid * (K >> 1);
            asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
            // FUSED: the un-normalised tile is stored with AGENT scope (sc1: written through the L2 instead of staying there as
            // dirty lines until something evicts them) -- a wave of this CU reads it back a signal later through a streaming load, by
            // which time the L2 has long dropped it either way (profiles/r02_fused_team_variant.txt), and the final stores overwrite
            // it.  Measured against the ordinary store, interleaved A/B, three runs: 0.2114 -> 0.2043, 0.2171 -> 0.2093, 0.2078 ->
            // 0.2000 ms per launch (-3.5 %); sc0 sc1 the same; sc0 alone nothing; nt 16 % slower (profiles/r03_lead_ab.txt).
            // The instruction comes from inline assembly (no builtin carries the scope bits of a 16-byte store), so the two things
            // the compiler does for its own stores are done by hand: the wait state of the store-data hazard (s_nop), and the
            // s_waitcnt vmcnt(0) in front of the release on the delivery counter below.
#ifndef HSS_A_STORE_POLICY
#define HSS_A_STORE_POLICY "sc1"
#endif
            auto put = [&](int i) {
                if constexpr (FUSED) { asm volatile("global_store_dwordx4 %0, %1, off " HSS_A_STORE_POLICY "\n\ts_nop 1" :: "v"(dst4 + 64 * i), "v"(o[i]) : "memory"); }
                else __builtin_nontemporal_store(o[i], reinterpret_cast<f4*>(dst4 + 64 * i));
            };
            if (__builtin_expect(nvalid == 16, 1)) {
                // a full group is 8 K float4s: every lane has the first (8 K / 64) of its three, one predicate for the rest
                // (three compare + exec-mask + branch sequences otherwise)
                static_for<3>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (64 * (i + 1) <= 8 * K) put(i);
                    else if constexpr (64 * i < 8 * K) { if (lane_o + 64 * i < 8 * K) put(i); }
                });
            } else {
                asm volatile("");
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (lane_o + 64 * i < lim) put(i);
            }
            }
            wave_sync();
        }
        gcur = gstop;
    }
#ifdef HSS_FUSE_PROBE
    const unsigned long long fp_ta = FPROBE_NOW();
    fp[0] += 1; fp[1] += fp_ta - fp_t0;
#endif
    if constexpr (FUSED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler does not count the stores issued from inline assembly)
        const int sl = static_cast<int>(ksig) & 1;
        unsigned before = 0;
        if (lane == 0) before = __hip_atomic_fetch_add(done_a + sl, static_cast<unsigned>(ngrp), __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        before = __builtin_amdgcn_readfirstlane(before);
        if (before + static_cast<unsigned>(ngrp) == static_cast<unsigned>(ngroups) * (static_cast<unsigned>(ksig >> 1) + 1u)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int ncols_o = ncols, K_o = K, ng_o = ngroups;
            asm volatile("" : "+s"(ncols_o), "+s"(K_o), "+s"(ng_o));
            const float4 st = signal_stats(part_lds + sl * kFusedMaxGroups * kPartFloats, ng_o, 16, ncols_o, K_o, lane_o);
            if (lane == 0) {
                fin_stats[static_cast<int>(ksig) & 3] = st;
                __hip_atomic_store(ready + (static_cast<int>(ksig) & 3), static_cast<unsigned>(ksig) + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_sync();
#ifdef HSS_FUSE_PROBE
            fp[6] += FPROBE_NOW() - fp_ta;
#endif
        }
    }
    }
    chunk = draw();
    }
#ifdef HSS_FUSE_PROBE
    if (FUSED && lane == 0) for (int k = 0; k < 7; ++k) atomicAdd(g_fuse_probe + k, fp[k]);
#endif
}

}  // namespace hssfsst
