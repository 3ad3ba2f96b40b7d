// fsst_mfma128.hpp -- second-generation synchrosqueeze core for the canonical window length
// nwin = 128 (the reference's Kaiser(128) configuration, /root/reference/main.py:153-158); the same kernel with a
// radix-16 first stage in two passes also serves nwin = 256 and, with 32 taps, nwin = 512 (template parameters NT = taps,
// RQ = first-stage radix, nwin = NT RQ; the description below is written for (16, 8)).
//
// Why a second kernel: measured on MI355X (profiles/r01_valu_ubench.txt) a 3-operand fp32 FMA
// costs ~4 cycles per wave64 instruction, an add ~2.8, a PACKED v_pk_{add,mul,fma}_f32 ~4.3 for two
// results, one wave per SIMD issues a VALU op only every ~5.4 cycles, and wave-uniform table
// constants cost SALU issue slots.  So this kernel
//   * runs the one dense contraction of the path -- the window multiply fused with the first
//     radix-8 decimation-in-frequency stage,  y_r[n] = sum_{q<8} x[t+n+16q] * C_r[n,q]  (a constant
//     16 x 8 real matrix applied at every tap n, for 16 frames at a time) -- as 32
//     v_mfma_f32_16x16x4_f32 per 16 frames (bit-exact fp32 FMA chain).  Measured
//     (profiles/r01_mfma_valu_overlap_ubench.txt): on this chip MFMA time does NOT hide behind VALU
//     time of the same SIMD, so the gain is not concurrency but issue economy: 32 instructions
//     instead of 256 v_pk_fma_f32 + their constant-operand traffic for the same ALU time; the
//     constants come from a shared 8 kB LDS table (one ds_read_b64 per tap), the frame operand is
//     one contiguous ds_read2_b32 per tap (hop-1 frames are shifted copies);
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
#include <cstdlib>

#include "fsst_kernels.hpp"

#ifndef HSS_MW128
#define HSS_MW128 4
#endif

namespace hssfsst {

using f2 = float __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));
using lds_float = __attribute__((address_space(3))) float;

// Work distribution: one persistent block of 16 waves per CU.  The launch is cut into CHUNKS of consecutive 16-frame
// groups of one signal; the chunk list is ordered by region -- all region-0 chunks of all signals (8 groups each),
// then region 1 (4 groups), then region 2 (2 groups).  Block B owns chunks B, B + grid, B + 2 grid, ... (the same
// mix of big and small chunks for every CU) and its waves draw them in that order from a counter in LDS, so the 16
// waves of a CU finish within one small chunk of each other.  Why: the SIMD arbiter favours the oldest wave, so with
// equal static work per wave the four waves of a SIMD finished at 114 / 125 / 140 / 162 us, and with one fixed chunk
// per wave and several rounds of 4-wave blocks the last 20 % of the kernel ran at falling occupancy
// (profiles/r01_block_timeline.txt).  A ticket counter in HBM instead of LDS costs ~4 ns per draw, serialised
// chip-wide: 23 552 draws made the kernel 0.29 ms.  The chunk pattern of a signal depends only on the number of
// columns, not on the batch, and every chunk writes its own statistics partial: results are independent of the
// batch composition and run-to-run deterministic whichever wave processes a chunk.
struct Core128Regions {
    int g0[3];            // first 16-frame group of the region (per signal)
    int gpc[3];           // groups per chunk
    int npc[3];           // chunks per signal
};

struct Core128Params {
    const float* x;       // [batch][n]
    float* out;
    float* partials;      // [batch][groups][kPartFloats], one statistics partial per 16-frame group of the signal
    const float* atab;    // MFMA A-operand constants [16 taps][2 k-halves][64 lanes], then the FAST store offsets
    const double* wtab;   // float64 {w, dw' (bin units)}[nwin]        } the rounding-tie path (resolve_ties)
    const double* twtab;  // float64 {cos, sin}(2 pi m / nwin)[nwin]   }
    float r2scale;        // 4 nwin max_n |(w + i dw') / 2|^2[n]: R^2 = r2scale * sum x^2 over a tile bounds |V|^2 sums
    int n;
    int klo;
    int K;
    int mode;
    int nsig;             // signals in this launch
    int col0;             // first output column (frame centre) of every signal
    int ncols;            // number of output columns (== n for a whole-signal transform)
    long long xstride;    // samples between the starts of consecutive signals (n for a dense batch)
    Core128Regions reg;
    unsigned* status;         // FUSED kernel only: device status word, 0 = ok, else the code of a wait that gave up
    // STREAM kernel only (one rolling step, hssfsst_stream_step): x = the tape at the oldest sample the step needs, n = hist + chunk
    const float* xnew;        // the step's new samples [nsig][xnew_stride] (samples hist .. n - 1 of every signal), or null: already in the tape
    long long xnew_stride;
    int hist;                 // nwin - 1
    int bpc;                  // blocks per channel
    double* state;            // running moments [nsig][6], or null (no normalisation)
    unsigned* arrive;         // [nsig] blocks of the channel that have delivered (the last one merges and normalises, and clears it)
    double* pieces;           // [nsig][groups][4] the groups' float64 sums (chunk_moments' pieces, fsst_kernels.hpp)
    float* mirror;            // the caller's pinned host buffer for the step's features (device view), or null
};

// Chunk pattern for `ngroups` 16-frame groups per signal: 8-group chunks, then 4-group chunks over the last
// quarter or so, then 2-group chunks at the very end (each chunk costs a counter draw, a tile staging and a
// statistics reduction, so the small ones are kept to the tail).  Measured (tools/tail_sweep.sh): (16, 6) 0.1691 ms,
// (24, 2) 0.1683, (32, 6) 0.1701, (8, 4) 0.1738, 8-group chunks only 0.1749.
inline Core128Regions core128_regions(int ngroups, long long nsig = -1)
{
    Core128Regions r{};
#ifdef HSS_TAIL_ENV                                      // development only (tools/tail_sweep.sh)
    static const int env2 = std::getenv("HSSFSST_TAIL2") ? std::atoi(std::getenv("HSSFSST_TAIL2")) : 6;
    static const int env4 = std::getenv("HSSFSST_TAIL4") ? std::atoi(std::getenv("HSSFSST_TAIL4")) : 16;
    const int tail2 = ngroups >= 32 ? env2 : 0, tail4 = ngroups >= 32 ? env4 : 0;
#else
    const int tail2 = ngroups >= 32 ? 6 : 0;             // groups wanted as 2-group chunks
    const int tail4 = ngroups >= 32 ? 16 : 0;            // groups wanted as 4-group chunks
#endif
    if (ngroups < 32) {
        // short signals / streaming steps (a rolling transform adds 8 groups per step): parallelism matters more than
        // the per-chunk overhead -- 2-group chunks up to 8 groups, 4-group chunks up to 31; and single groups when
        // the whole launch is smaller than the chip (one streaming step of 64 channels = 512 groups for 1024 SIMDs:
        // the step's latency is then one group, not two)
        const bool tiny = nsig >= 0 && ngroups <= 8 && nsig * ngroups <= 1024;
        const int gpc = tiny ? 1 : ngroups <= 8 ? 2 : 4;
        r.g0[0] = 0; r.gpc[0] = 8; r.npc[0] = 0;
        r.g0[1] = 0; r.gpc[1] = 4; r.npc[1] = gpc == 4 ? (ngroups + 3) / 4 : 0;
        r.g0[2] = 0; r.gpc[2] = gpc == 1 ? 1 : 2; r.npc[2] = gpc == 1 ? ngroups : gpc == 2 ? (ngroups + 1) / 2 : 0;
        return r;
    }
    int big = ngroups - tail2 - tail4;
    big -= big % 8;                                      // whole 8-group chunks only
    if (big < 0) big = 0;
    int mid = ngroups - big - tail2;
    if (tail2 > 0) mid -= mid % 4;                       // whole 4-group chunks; the remainder joins the 2-group tail
    const int rest = ngroups - big - mid;
    r.g0[0] = 0;          r.gpc[0] = 8; r.npc[0] = big / 8;
    r.g0[1] = big;        r.gpc[1] = 4; r.npc[1] = (mid + 3) / 4;
    r.g0[2] = big + mid;  r.gpc[2] = 2; r.npc[2] = (rest + 1) / 2;
    return r;
}
__host__ __device__ inline int core128_chunks_per_signal(const Core128Regions& r) { return r.npc[0] + r.npc[1] + r.npc[2]; }


// cos / sin of 2*pi*j/16, j = 0..7
__device__ constexpr float kCos16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508978f,
                                        0.0f, -0.38268343236508973f, -0.70710678118654746f, -0.92387953251128674f};
__device__ constexpr float kSin16[8] = {0.0f, 0.38268343236508978f, 0.70710678118654746f, 0.92387953251128674f,
                                        1.0f, 0.92387953251128674f, 0.70710678118654757f, 0.38268343236508984f};

// (cos / sin of 2*pi*j/32 for the 32-tap variant, nwin = 512: kCos32 / kSin32 of fsst_kernels.hpp)
constexpr __host__ __device__ int bitrev4(int n) { return ((n & 1) << 3) | ((n & 2) << 1) | ((n & 4) >> 1) | ((n & 8) >> 3); }
template <int N>
constexpr __host__ __device__ int bitrev_n(int n)
{
    return N == 16 ? bitrev4(n) : ((bitrev4(n & 15) << 1) | (n >> 4));        // N == 32: 5 bits
}

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Packed complex helpers with VOP3P operand swizzles (op_sel) and sign modifiers (neg_lo/neg_hi):
// hipcc materialises (y, -x)-style operands with v_xor + v_mov instead of using the modifiers.
// Inputs of these statements are always results of ordinary VALU instructions (never MFMA
// destinations), so no manual hazard padding is needed inside the strings.
// Two-for-one unpack of a source bin X = Z[k'] with conjugate partner P = Z[128 - k'], emitted directly in
// the operand layout of the packed den/num product below:
//   V   = X + conj(P)       = (X.x + P.x, X.y - P.y)
//   Vd' = (X - conj(P)) / i = (X.y + P.y, P.x - X.x)
__device__ __forceinline__ f2 mix_re(f2 x, f2 p)         // (V.re, Vd'.im) = (p.x + x.x, p.x - x.x)
{
    f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(x)); return d;
}
__device__ __forceinline__ f2 mix_im(f2 x, f2 p)         // (V.im, Vd'.re) = (x.y - p.y, x.y + p.y)
{
    f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1]" : "=v"(d) : "v"(x), "v"(p)); return d;
}
// (den, num) = (|V|^2, Vd'.re V.im - Vd'.im V.re) in two packed FMAs: t1 = mix_re, t2 = mix_im
__device__ __forceinline__ f2 dn_first(f2 t1, f2 c)      // (t1.lo^2, -t1.lo t1.hi) + c
{
    f2 d; asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(t1), "s"(c)); return d;
}
__device__ __forceinline__ f2 dn_second(f2 t2, f2 m)     // (t2.lo^2, t2.lo t2.hi) + m
{
    f2 d; asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(t2), "v"(m)); return d;
}
__device__ __forceinline__ f2 add_mi(f2 e, f2 o)         // e + (-i) o
{
    f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(e), "v"(o)); return d;
}
__device__ __forceinline__ f2 sub_mi(f2 e, f2 o)         // e - (-i) o
{
    f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(e), "v"(o)); return d;
}
// Radix-2 DIT butterfly on packed complex values, twiddle W = exp(-2*pi*i*TW/N), N = 16 or 32.
template <int N, int TW>
__device__ __forceinline__ void bfly_n(f2& e, f2& o)
{
    f2 a, b;
    if constexpr (TW == 0) {
        a = e + o; b = e - o;
    } else if constexpr (TW == N / 4) {                 // W = -i: W o = (o.im, -o.re)
        a = add_mi(e, o); b = sub_mi(e, o);
    } else {                                            // W = wr + i wi, wr = cos, wi = -sin
        constexpr float wr = (N == 16) ? kCos16[TW & 7] : kCos32[TW & 15];
        constexpr float wi = (N == 16) ? -kSin16[TW & 7] : -kSin32[TW & 15];
        const f2 t1 = pk_fma(f2{o.x, o.x}, f2{wr, wi}, e);
        a = pk_fma(f2{o.y, o.y}, f2{-wi, wr}, t1);      // e + W o
        b = pk_fma(e, f2{2.0f, 2.0f}, -a);              // e - W o = 2e - (e + W o)
    }
    e = a; o = b;
}

// In-place N-point complex FFT (forward), N = 16 or 32; input in bit-reversed order, output natural order.
template <int N>
__device__ __forceinline__ void fft_n(f2 (&z)[N])
{
    constexpr int LOG2N = (N == 16) ? 4 : 5;
    static_for<LOG2N>([&](auto S) {
        constexpr int L = 2 << decltype(S)::value;
        constexpr int H = L / 2;
        constexpr int STEP = N / L;
        static_for<N / 2>([&](auto B) {
            constexpr int b = decltype(B)::value;
            constexpr int j = b % H;
            constexpr int a = (b / H) * L + j;
            bfly_n<N, j * STEP>(z[a], z[a + H]);
        });
    });
}

// ---- LDS planes of one 16-frame group (per wave), packed complex (re, im) per cell:
//   own  [16 frames][own_ld]  columns for rows 8*s0 .. 8*s1+7, the 8-aligned cover of the kept band:
//                             source k' stores (-1)^k' V[k'] into its column unconditionally
//                             (address = per-lane base + compile-time offset); sources whose
//                             8-row stripe lies outside the cover skip the store (wave-uniform);
//   disp [16 frames][LDF(K)]  kept rows only, zero-initialised: corrections from displaced sources.
__host__ __device__ constexpr int odd_up(int v) { return (v & 1) ? v : v + 1; }      // odd => b64 conflict-free
__host__ __device__ constexpr int plane_ldf(int K) { return odd_up(K); }
// rq = first-stage radix = rows per stripe (8 for nwin = 128, 16 for nwin = 256): source k' = rq * s + r sits in stripe s
__host__ __device__ constexpr int own_s0(int klo, int rq = 8) { return klo / rq; }
__host__ __device__ constexpr int own_s1(int klo, int K, int rq = 8) { return (klo + K - 1) / rq; }   // inclusive stripe
__host__ __device__ constexpr int own_ld(int klo, int K, int rq = 8)
{
    return odd_up(rq * (own_s1(klo, K, rq) - own_s0(klo, rq) + 1) + 1);
}
// MFMA A-operand constants: [pass][nt taps][k-step][64 lanes] floats, rq / 8 passes of rq / 4 k-steps
__host__ __device__ constexpr int core128_atab_floats(int rq = 8, int nt = 16) { return (rq / 8) * nt * (rq / 4) * 64; }
constexpr int kMaxWavesPerBlock = 16;        // 16 = one block owns a whole CU (4 waves per SIMD); fewer when LDS is short
constexpr int kCtlFloats = 16 + 192;         // block control words in LDS: [0] work counter, [16..207] the wide-store offset
                                             // table (3 words per lane: held in registers it costs the 16-wave kernels a spill)
// FUSED kernel: [0] ticket counter, [1..2] groups delivered per signal slot (monotone), [3] a wait gave up, [4..7] epoch
// of the resolved statistics (4 slots), [8..15] (unused), [16..79] per-lane column classes of the z-score
// pass, [80..271] the wide-store offset table (3 words per lane; the fused kernel has no register to spare for it),
// [272..287] four float4 statistics, [288 ..] the statistics partials of two signals [2][kFusedMaxGroups][kPartFloats]
constexpr int kFusedMaxGroups = 128;         // signals of at most 2048 frames
constexpr int kFusedMinChunks = 16;          // and of at least 16 chunks: see "Slots" in the kernel
constexpr int kCtlFusedFloats = 288 + 2 * kFusedMaxGroups * kPartFloats;

// FAST epilogue: byte offsets, inside a wave's own plane, of the two (re,re) / (im,im) pairs that make up
// float4 number f = lane + 64 i of a 16-frame group's contiguous [16][2K] output image (K even, K <= 24).
// tab[i][lane] = first pair, tab[3 + i][lane] = second pair; a pair's second element is 8 bytes further.
inline void core128_store_offsets(int klo, int K, int* tab /* [6][64] */, int rq = 8)
{
    const int Q = (K >> 1) > 0 ? (K >> 1) : 1;
    const int old = own_ld(klo, K, rq), koff0 = klo - rq * own_s0(klo, rq);
    for (int i = 0; i < 3; ++i)
        for (int lane = 0; lane < 64; ++lane) {
            const int f = lane + 64 * i;
            const int jj = (f / Q < 15) ? f / Q : 15, c = 4 * (f - (f / Q) * Q);
            const int rowb = jj * old + koff0;
            tab[i * 64 + lane] = (c < K) ? (rowb + c) * 8 : (rowb + c - K) * 8 + 4;
            tab[(3 + i) * 64 + lane] = (c + 2 < K) ? (rowb + c + 2) * 8 : (rowb + c + 2 - K) * 8 + 4;
        }
}
// Rounding ties.  A displaced source lands in row round(k' + shift).  The float32 estimate of the shift carries an
// error of ~1e-7 (1 + |shift|) max|Z| / |V|: for a cell that is small against its frame's spectrum and moves tens of
// bins (Hann-, Blackman-, Kaiser(beta >> 1)-class windows) that is up to ~1e-2 bins, enough to put the cell into the
// neighbouring row of the float64 reference.  The kernel therefore carries an error bound with every displaced cell:
// with R^2 = 4 nwin max|c|^2 sum x^2 over the staged tile (Parseval: an upper bound of sum |Z|^2 of every frame of the tile,
// c = (w + i dw') / 2) the coordinate is good to about  tau = 1e-6 (1 + |shift|) R / |V|.  A displaced cell whose float32
// coordinate lies within tau of a half-integer is NOT moved on the spot: it goes to a per-wave queue in LDS (frame, bin,
// value) and, once the group's spectra are done and their registers free, the whole wave recomputes that one bin of V
// and Vd' by a float64 DFT of the frame (two taps per lane for nwin = 128, window and twiddle tables in float64 from
// HBM, a float64 butterfly sum) and rounds the float64 coordinate.  Cells below 1e-6 R are left to float32: they
// cannot change a feature by 1e-4 of the largest feature wherever they land.  Large cells have tau ~ 1e-6 and are
// practically never queued; the queue holds the small far-moving cells that float32 cannot place.
// No queues (rounds 1-2 queued such cells, 240 + 24 per group, and fell back to float32 beyond that: tonal and offset-
// dominated inputs under low-sidelobe windows overflowed them -- profiles/r02_adversarial_parity.txt class iii): an undecided
// cell sets ONE BIT of a per-group bitmap in LDS, bit 16 (k' & 1) + frame of word k' >> 1, k' = 0 .. nwin/2 - 1, so their number
// is not limited by anything and the bitmap is a sixth of the queues' size.  Resolution (resolve_bitmap): from kTieGroup64
// cells per 64 sources on, the whole group at once by the float64 fold + DFT factorisation (resolve_group_f64); below that, per
// set of 64 sources, up to kTieCoop cells one by one with the whole wave on one float64 DFT, more than that lane l takes source
// k' = l of the set and the wave walks the frames that have a bit set (one frame per round: its windowed samples are handed
// round by v_readlane).  The float32 V of a cell inside the stored cover of the own plane is read back from -- and cleared
// in -- its own column; for a cell outside it V is the float64 DFT's own result, rounded once.
constexpr int kTieCoop = 6;                  // up to this many undecided cells of a 64-source set the wave resolves one by one
__host__ __device__ constexpr int tie_words(int nwin) { return nwin / 4; }     // the bitmap (flag[1] = "some bit is set")
constexpr float kTieMargin = 1.0f / 64.0f;   // the stay-in-row test hands |shift| > 1/2 - this to the rare path
constexpr float kTieErr2 = 1.0e-12f;         // (1e-6)^2: tau^2 = kTieErr2 (1 + |shift|)^2 R^2 / |V|^2  (4e-7 left 2 of 1000
                                             // random configurations 1.4-1.8x over the gate: tools/fuzz_parity.py 1000 3)
constexpr float kTieFloor2 = 1.0e-12f;       // (1e-6)^2: cells with |V|^2 below this times R^2 stay with float32.  R over-
                                             // estimates the frame's spectrum norm by up to ~5x and that norm is at most
                                             // sqrt(nwin) times the largest bin, so 1e-6 R is < 1e-4 of the largest feature
                                             // (1e-5 R was not: 3 of 2000 random narrow-band configurations failed the gate)

__host__ __device__ constexpr int wave_lds_floats(int fpw, int klo, int K, int rq = 8, int nt = 16)
{
    return ((fpw + nt * rq - 1 + 3) / 4) * 4 + 2 * 16 * (own_ld(klo, K, rq) + plane_ldf(K)) + 4   // + dirty flag
           + tie_words(nt * rq);                                                                     // + tie queues
}

__device__ __forceinline__ void wave_sync()
{
    // LDS traffic of one wave is executed in program order: only the compiler must not reorder.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Moves a source k' -> row in a frame's displaced plane: the own plane holds V in the source's own column
// (unconditional store), so V is taken out again there, added at `row` and -- conjugated -- at the negative-frequency
// twin's row nwin - row (oracle/fsst_oracle.c step 6: two-sided cyclic scatter, one-sided rows kept).
// The odd wave of a PAIR (fsst_core128_kernel) does not add into the displaced plane while the even wave does: it LISTS its
// additions -- (cell, re, im) in program order, lanes in order (ballot prefix: no atomics) -- and the even wave replays the list
// behind its own additions: the plane receives every addition in the order of the one-wave kernel, whichever wave runs faster.
constexpr int kPairListCap = 256;
struct PairList {
    int* cell;            // [kPairListCap] float index into the displaced plane (re component; im = + 1)
    f2* val;              // [kPairListCap]
    int* cnt;             // (in LDS) additions listed; > kPairListCap: entries were dropped, the group is redone without the list
    int rowoff;           // this lane's frame row of the displaced plane, in cells
};
// COLS (K <= 24): flag[2..7] collect WHICH columns of the displaced plane were added to (a byte per column), so that the fold looks at
// those only; flag[0] says that there is something to fold, as without COLS.
template <int NWIN, bool LANE_OWNS = false, bool LIST = false, bool COLS = false>
__device__ __forceinline__ void move_source(f2* row_disp, int* flag, int klo, int K, int kpi, int row, f2 V,
                                            f2* own_cell = nullptr, bool stored = false, PairList* pl = nullptr)
{
    auto add = [&](int idx, float re, float im) {
        if constexpr (LIST) {
            // (the count lives in LDS: a variable updated under a divergent branch would be per-lane, stale in the lanes that
            //  did not take it; the first active lane reserves the slots of all of them)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(true);      // the lanes that add here, in lane order
            const int below = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u)));
            int base = 0;
            if (below == 0) base = __hip_atomic_fetch_add(pl->cnt, static_cast<int>(__builtin_popcountll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int slot = __builtin_amdgcn_readfirstlane(base) + below;
            if (slot < kPairListCap) { pl->cell[slot] = 2 * (pl->rowoff + idx); pl->val[slot] = f2{re, im}; }
        } else {
            float* q = reinterpret_cast<float*>(row_disp + idx);
            __hip_atomic_fetch_add(q, re, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(q + 1, im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    if (row == kpi) return;                             // rounds back into its own row after all
    if constexpr (COLS && LANE_OWNS && !LIST) if (klo >= 1 && klo + K <= NWIN / 2) {
        // a band inside rows 1 .. nwin/2 - 1 (the canonical-band kernels; row 0 is its own twin and keeps the general form below): a
        // destination is in the band, or -- only above nwin/2 -- its twin is; the twin's arithmetic is spent on the few sources that
        // wrap around row 0
        if (stored) *own_cell = f2{0.0f, 0.0f};
        const int idx = row - klo;
        unsigned char* colb = reinterpret_cast<unsigned char*>(flag + 2);     // one byte per column (K <= 24): plain stores, no atomic
        if (static_cast<unsigned>(idx) < static_cast<unsigned>(K)) {
            add(idx, V.x, V.y);
            colb[idx] = 1; *flag = 1;
        } else if (row > NWIN / 2 && kpi != 0) {
            const int idm = (NWIN - row) - klo;
            if (static_cast<unsigned>(idm) < static_cast<unsigned>(K)) {
                add(idm, V.x, -V.y);
                colb[idm] = 1; *flag = 1;
            }
        }
        return;
    }
    const int own = kpi - klo, idx = row - klo;
    const int idm = ((NWIN - row) & (NWIN - 1)) - klo;  // negative-frequency twin: row -> nwin - row, value conj
    const bool in_row = static_cast<unsigned>(idx) < static_cast<unsigned>(K);
    const bool in_twin = kpi != 0 && static_cast<unsigned>(idm) < static_cast<unsigned>(K);     // (k' = nwin/2 never moves)
    bool touched = in_row | in_twin;
    // taking V out of its own column: the lane that stored it a moment ago (LANE_OWNS; own_cell = its cell of the own
    // plane, `stored` = wave-uniform: the source's stripe lies inside the stored cover, outside it there is nothing to
    // take out) simply clears it; the float64 path, which runs later and for any lane's cell, subtracts it in the
    // displaced plane
    if constexpr (LANE_OWNS) {
        if (stored) *own_cell = f2{0.0f, 0.0f};
    } else if (static_cast<unsigned>(own) < static_cast<unsigned>(K)) {
        add(own, -V.x, -V.y);
        touched = true;
    }
    if (in_row) add(idx, V.x, V.y);
    if (in_twin) add(idm, V.x, -V.y);
    if constexpr (COLS) {                               // (bands that take the general form: row 0 inside the band)
        unsigned char* colb = reinterpret_cast<unsigned char*>(flag + 2);
        if (in_row) colb[idx] = 1;
        if (in_twin) colb[idm] = 1;
    }
    if (touched) *flag = 1;                             // this wave's displaced plane is no longer zero: ONE write
}

// The move for a plane that takes the displaced additions ITSELF (fsst_canon128.hpp, "One plane"; bands inside rows 1 .. nwin/2 - 1): the
// source's own cell has been cleared by whoever found it moving, V is added at `row` (its own row included: it "rounds back") or, for a
// source that wraps around row 0, conjugated at the twin's.  `row0` = this frame's row of the plane, indexed by spectrum row.
template <int NWIN>
__device__ __forceinline__ void acc_source(f2* row0, int klo, int K, int kpi, int row, f2 V)
{
    if (static_cast<unsigned>(row - klo) < static_cast<unsigned>(K)) {
        float* q = reinterpret_cast<float*>(row0 + row);
        __hip_atomic_fetch_add(q, V.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(q + 1, V.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (row > NWIN / 2 && kpi != 0) {
        const int idm = NWIN - row;                      // negative-frequency twin: row -> nwin - row, value conj
        if (static_cast<unsigned>(idm - klo) < static_cast<unsigned>(K)) {
            float* q = reinterpret_cast<float*>(row0 + idm);
            __hip_atomic_fetch_add(q, V.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(q + 1, -V.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// Rare path for a displaced source: oracle/fsst_oracle.c steps 4-6 in fp32, except for coordinates too close to a
// rounding tie, which are queued for resolve_ties().  `row_disp` points at this lane's frame row (frame j of the
// group) in the displaced plane.
template <int NWIN, int ERRMUL = 1, bool LIST = false>
__device__ __forceinline__ void displaced_source(f2* row_disp, int* flag, int* tq, int klo, int K, int kpi, int j,
                                                 float num, float den, f2 V, float R2, f2* own_cell, bool stored, PairList* pl = nullptr)
{
    float shift = num * __builtin_amdgcn_rcpf(den);
    if (!(fabsf(shift) <= 1.0e6f)) shift = 0.0f;        // NaN / inf / absurd -> 0 (fsst.m: ~isfinite)
    const float a = static_cast<float>(kpi) + shift;
    float fr = a - floorf(a) - 0.5f;
    const float s1 = 1.0f + fabsf(shift);
    // (opaque: otherwise the two independent product chains below are packed into v_pk_mul_f32 pairs whose operand
    //  shuffles and hazard nops cost more than the five plain multiplies)
    asm volatile("" : "+v"(fr));
#ifdef HSS_NO_TIES                                       // development: cost of the tie path (tools/ab_bench.py)
    if (false) {
#else
    if (fr * fr * den < (kTieErr2 * ERRMUL) * s1 * s1 * R2 && den > kTieFloor2 * R2) {     // too close to call in float32
#endif
        __hip_atomic_fetch_or(reinterpret_cast<unsigned*>(tq) + (kpi >> 1), 1u << (((kpi & 1) << 4) + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        flag[1] = 1;
        return;
    }
    const float r = truncf(a + copysignf(0.5f, a));     // MATLAB round: half away from zero
    move_source<NWIN, true, LIST>(row_disp, flag, klo, K, kpi, static_cast<int>(r) & (NWIN - 1), V, own_cell, stored, pl);
}

// The whole wave, after the group's spectra: every undecided cell's bin of V and Vd' by a float64 DFT of its frame
// (sample(i) = sample i of the group's first frame as a double: the float32 signal itself; wtab[n] = {w, dw'}[n],
// twtab[m] = {cos, sin}(2 pi m / nwin), float64), the float64 coordinate k' - Im(Vd'/V) rounded half away from zero, then the
// move.  [cov0, cov1) = the sources whose float32 V sits in the own plane (leading dimension OLD): it is read back from --
// and cleared in -- its own column; a source outside the cover has no own cell and none in the band: its V is the float64
// DFT's, times the plane's sign (-1)^k' (even nwin: the modified-STFT phase) and `plane_scale` (1 unless the plane holds
// scaled values, fsst_canon128.hpp).
// REFRESH (the exact mode of a group, see "Exact groups" below): the own cell is first rewritten with the float64 value.
// ACC: the own plane takes the additions itself and holds nothing of an undecided source (its cell was cleared when it was found
// undecided: fsst_canon128.hpp, "One plane"): V is the float64 DFT's, rounded once, added where it belongs -- its own row included.
template <int NWIN, bool REFRESH = false, bool ACC = false>
__device__ __forceinline__ void resolve_one(f2* disp_base, int LDF, int* flag, int klo, int K, f2* own_base, int OLD, int cov0, int cov1,
                                            int kpi, int jf, double vr, double vi, double dr, double di, double plane_scale)
{
    const double den = vr * vr + vi * vi;
    double shift = (dr * vi - di * vr) / den;
    if (!(fabs(shift) <= 1.0e6)) shift = 0.0;           // V == 0 or absurd -> 0 (fsst.m: ~isfinite)
    const double a = static_cast<double>(kpi) + shift;
    const double r = (a >= 0.0) ? floor(a + 0.5) : -floor(0.5 - a);
    const int row = static_cast<int>(static_cast<long long>(r)) & (NWIN - 1);
    const double sg = (kpi & 1) ? -plane_scale : plane_scale;
    if constexpr (ACC) {
        acc_source<NWIN>(own_base + jf * OLD - cov0, klo, K, kpi, row, f2{static_cast<float>(vr * sg), static_cast<float>(vi * sg)});
        return;
    }
    if (kpi >= cov0 && kpi < cov1) {
        f2* cell = own_base + jf * OLD + (kpi - cov0);
        f2 V;
        if constexpr (REFRESH) { V = f2{static_cast<float>(vr * sg), static_cast<float>(vi * sg)}; *cell = V; }
        else V = *cell;
        move_source<NWIN, true>(disp_base + jf * LDF, flag, klo, K, kpi, row, V, cell, true);
    } else {
        const f2 V = {static_cast<float>(vr * sg), static_cast<float>(vi * sg)};
        move_source<NWIN, true>(disp_base + jf * LDF, flag, klo, K, kpi, row, V, nullptr, false);
    }
}

// Exact groups.  float32 resolves a feature to ~4e-7 of its FRAME's spectrum norm, whatever the feature's own size: when the kept
// band holds nothing but the far leakage of an out-of-band component (a tone or an offset 1e3 times the in-band content under
// a low-sidelobe window) that is more than 1e-4 of the band's largest feature (profiles/r02_adversarial_parity.txt, classes i
// and ii).  A group none of whose stored cells reaches kExactTheta R (R = the spectrum-norm bound of the group's own samples)
// is therefore redone in float64: REFRESH = true treats EVERY cell as undecided -- lane l takes source k' = l, walks all 16
// frames, rewrites the own cell with the float64 value, rounds the float64 coordinate, moves.  ~50x the cost of the group;
// ordinary signals never get there (white noise: largest cell 0.09 R).
constexpr float kExactTheta2 = 1.0e-4f;      // (1e-2)^2
__device__ __forceinline__ double readlane_f64(double v, int l)      // lane l's value in every lane (l wave-uniform)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(b), l));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(b >> 32), l));
    return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}
// TWLDS: `twtab` points into LDS (the canonical-band kernels keep the 2 kB of twiddles beside their operand table).
using d2v = double __attribute__((ext_vector_type(2)));
using lds_d2v = __attribute__((address_space(3))) d2v;
template <bool TWLDS>
__device__ __forceinline__ d2v tie_twiddle(const double* twtab, int idx)
{
    if constexpr (TWLDS) return ((const lds_d2v*)reinterpret_cast<const d2v*>(twtab))[idx];
    else return reinterpret_cast<const d2v*>(twtab)[idx];
}
// Heavily undecided groups (an on-bin tone under a near-rectangular window, a band that holds only leakage: ~all of the group's
// 16 nwin / 2 (source, frame) cells).  One float64 DFT per cell is nwin taps x 4 FMAs; done for all cells it is the fold +
// NT-point DFT factorisation of the float32 kernel in float64 instead: per tap ONE chain of RQ / 4 v_mfma_f64_16x16x4_f64
// gives lane (g, j) the folded {Za, Zb}[tap] of its two classes for frame j (A operand: the float64 table a64 behind the
// twiddles of `wtab`, made by the host with the row order of the float64 instruction; B operand: the float32 samples, exact
// in float64), and the lane accumulates the NT-point DFT outputs of SPP stripes at a time (registers) -- X = Z[RQ s + r] and
// its conjugate partner, as process_stripe.  NT complex multiply-adds per cell instead of nwin real-complex ones.
// Measured at 128 points on an on-bin tone (every group): 22.8 -> 2.9 ms per 1024 windows, 1.0 of it the displaced cells
// themselves.  The fold is redone in every pass (64 cycles per float64 matrix instruction on this chip: ~40 % of the path at 128
// points); holding its NT x 4 doubles per lane would take 128 / 256 registers, two stripes per pass made the 128-register
// kernels spill (SPP = 1 there; the 256-register kernels of 256 / 512 points take 4 / 2), and making the A operand on the spot
// from the window pair and twiddles in LDS instead of loading it was slower (4.3 ms).
// XREG: the B operand's samples are fetched once into registers (the canonical-band kernels: sample() is a load from HBM / L2);
// otherwise sample() is called per tap (a read of the float32 tile in LDS).  sample(i): sample i of the group's first frame.
__host__ __device__ constexpr int fold64_doubles(int rq, int nt) { return (rq / 8) * nt * (rq / 4) * 64; }
__device__ __forceinline__ d2v uniform_d2v(d2v v)
{
    auto u = [](double x) -> double {
        const long long b = __double_as_longlong(x);
        const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(b)));
        const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(b >> 32)));
        return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
    };
    return d2v{u(v.x), u(v.y)};
}
using d4v = double __attribute__((ext_vector_type(4)));
template <int NT, int RQ, int SPP, bool XREG, bool REFRESH, bool TWLDS, class Sample>
__device__ __forceinline__ void resolve_group_f64(const unsigned* tb, Sample sample, f2* disp_base, int LDF, int* flag, int klo, int K,
                                                  f2* own_base, int OLD, int cov0, int cov1, const double* a64, const double* twtab,
                                                  double plane_scale, int lane)
{
    constexpr int NWIN = NT * RQ, NPASS = RQ / 8, KST = RQ / 4;
    static_assert((NT / 2) % SPP == 0, "whole passes");
    const int g = lane >> 4, j = lane & 15;
    float xs[XREG ? KST : 1][XREG ? NT : 1];             // B operand rows kk = g + 4 ks of every tap: x[j + n + NT kk]
    if constexpr (XREG) {
#pragma unroll
        for (int h = 0; h < KST; ++h)
#pragma unroll
            for (int n = 0; n < NT; ++n) xs[h][n] = static_cast<float>(sample(j + n + NT * (g + 4 * h)));
    }
    // this lane's flagged frames: bit (k' & 1) * 16 + j of word k' >> 1
    auto flagged = [&](int kp) -> bool { return REFRESH || ((tb[kp >> 1] >> (((kp & 1) << 4) + j)) & 1u) != 0u; };
    auto cmac = [](double& ar, double& ai, double zr, double zi, d2v cs) {      // a += z (cos - i sin)
        ar = fma(zr, cs.x, ar); ar = fma(zi, cs.y, ar);
        ai = fma(zi, cs.x, ai); ai = fma(-zr, cs.y, ai);
    };
#pragma unroll 1
    for (int pz = 0; pz < NPASS; ++pz) {
        const int pair = 4 * pz + g;
        const bool isg0 = pair == 0;                         // the self-conjugate classes {0, RQ / 2}
        const int rA = pair, rB = isg0 ? RQ / 2 : RQ - pair;
        const double* ap = a64 + pz * (NT * KST * 64);       // (uniform base + lane index: no per-lane 64-bit pointer to hold)
#pragma unroll 1
        for (int s0 = 0; s0 < NT / 2; s0 += SPP) {
            double xa[SPP][2], xb[SPP][2], pza[SPP][2], pzb[SPP][2];
#pragma unroll
            for (int u = 0; u < SPP; ++u) { xa[u][0] = xa[u][1] = xb[u][0] = xb[u][1] = pza[u][0] = pza[u][1] = pzb[u][0] = pzb[u][1] = 0.0; }
#pragma unroll 4
            for (int n = 0; n < NT; ++n) {
                d4v z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    const double a = ap[(n * KST + ks) * 64 + lane];
                    double x;
                    if constexpr (XREG) x = static_cast<double>(xs[ks][n]);
                    else x = sample(j + n + NT * (g + 4 * ks));
                    z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, z, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < SPP; ++u) {
                    const int s = s0 + u;
                    const int ia = isg0 ? ((NT - s) & (NT - 1)) : NT - 1 - s;    // partner index in array a (process_stripe)
                    // (t1, t2 are the same in every lane: scalar registers -- eight vector registers the 128-register kernels lack)
                    const d2v t1 = uniform_d2v(tie_twiddle<TWLDS>(twtab, (RQ * s * n) & (NWIN - 1)));
                    const d2v t2 = uniform_d2v(tie_twiddle<TWLDS>(twtab, (RQ * (NT - 1 - s) * n) & (NWIN - 1)));
                    const d2v t3 = tie_twiddle<TWLDS>(twtab, (RQ * ia * n) & (NWIN - 1));
                    cmac(xa[u][0], xa[u][1], z.x, z.y, t1);
                    cmac(xb[u][0], xb[u][1], z.z, z.w, t1);
                    cmac(pzb[u][0], pzb[u][1], z.z, z.w, t2);
                    cmac(pza[u][0], pza[u][1], z.x, z.y, t3);
                }
            }
#pragma unroll
            for (int u = 0; u < SPP; ++u) {
                const int s = s0 + u;
                const double par = isg0 ? pza[u][0] : pzb[u][0], pai = isg0 ? pza[u][1] : pzb[u][1];      // partner of source a
                const double pbr = isg0 ? pzb[u][0] : pza[u][0], pbi = isg0 ? pzb[u][1] : pza[u][1];      // partner of source b
                const int ka = RQ * s + rA, kb = RQ * s + rB;
                // plane values (-1)^k' V = X + conj(P), (-1)^k' Vd' = (X - conj(P)) / i; resolve_one takes V, Vd' themselves
                const double sa = (ka & 1) ? -1.0 : 1.0, sb = (kb & 1) ? -1.0 : 1.0;
                if (flagged(ka))
                    resolve_one<NWIN, REFRESH>(disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, ka, j,
                                               sa * (xa[u][0] + par), sa * (xa[u][1] - pai), sa * (xa[u][1] + pai), sa * (par - xa[u][0]), plane_scale);
                if (flagged(kb))
                    resolve_one<NWIN, REFRESH>(disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, kb, j,
                                               sb * (xb[u][0] + pbr), sb * (xb[u][1] - pbi), sb * (xb[u][1] + pbi), sb * (pbr - xb[u][0]), plane_scale);
            }
        }
    }
}

// nwin = 128 (16 taps, radix 8, one stripe per pass, samples in registers): the same, written out for the 128-register kernels
// -- the general form above needs a few registers more than they have.
template <bool REFRESH, bool TWLDS, bool ACC = false, class Sample>
__device__ __forceinline__ void resolve_group_f64_128(const unsigned* tb, Sample sample, f2* disp_base, int LDF, int* flag, int klo, int K,
                                                  f2* own_base, int OLD, int cov0, int cov1, const double* a64, const double* twtab,
                                                  double plane_scale, int lane)
{
    constexpr int NWIN = 128, NT = 16, RQ = 8;
    const int g = lane >> 4, j = lane & 15;
    const bool isg0 = g == 0;
    const int rA = g, rB = isg0 ? RQ / 2 : RQ - g;
    float xs[2][NT];                                     // B operand rows kk = g and g + 4 of every tap: x[j + n + 16 kk]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < NT; ++n) xs[h][n] = static_cast<float>(sample(j + n + 16 * (g + 4 * h)));
    // this lane's flagged frames: bit (k' & 1) * 16 + j of word k' >> 1
    auto flagged = [&](int kp) -> bool { return REFRESH || ((tb[kp >> 1] >> (((kp & 1) << 4) + j)) & 1u) != 0u; };
    auto cmac = [](double& ar, double& ai, double zr, double zi, d2v cs) {      // a += z (cos - i sin)
        ar = fma(zr, cs.x, ar); ar = fma(zi, cs.y, ar);
        ai = fma(zi, cs.x, ai); ai = fma(-zr, cs.y, ai);
    };
#pragma unroll 1
    for (int s = 0; s < NT / 2; ++s) {                   // (one stripe per pass: 16 accumulator registers; two made the 128-register kernels spill)
        double xa[2] = {0.0, 0.0}, xb[2] = {0.0, 0.0}, pza[2] = {0.0, 0.0}, pzb[2] = {0.0, 0.0};
        const int ia = isg0 ? ((NT - s) & (NT - 1)) : NT - 1 - s;                // partner index in array a (process_stripe)
#pragma unroll 2
        for (int n = 0; n < NT; ++n) {
            const double a0 = a64[(2 * n) * 64 + lane], a1 = a64[(2 * n + 1) * 64 + lane];
            d4v z = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, static_cast<double>(xs[0][n]), d4v{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, static_cast<double>(xs[1][n]), z, 0, 0, 0);
            // (t1, t2 are the same in every lane: scalar registers -- eight vector registers the 128-register kernels lack)
            const d2v t1 = uniform_d2v(tie_twiddle<TWLDS>(twtab, (RQ * s * n) & (NWIN - 1)));
            const d2v t2 = uniform_d2v(tie_twiddle<TWLDS>(twtab, (RQ * (NT - 1 - s) * n) & (NWIN - 1)));
            const d2v t3 = tie_twiddle<TWLDS>(twtab, (RQ * ia * n) & (NWIN - 1));
            cmac(xa[0], xa[1], z.x, z.y, t1);
            cmac(xb[0], xb[1], z.z, z.w, t1);
            cmac(pzb[0], pzb[1], z.z, z.w, t2);
            cmac(pza[0], pza[1], z.x, z.y, t3);
        }
        const double par = isg0 ? pza[0] : pzb[0], pai = isg0 ? pza[1] : pzb[1];      // partner of source a
        const double pbr = isg0 ? pzb[0] : pza[0], pbi = isg0 ? pzb[1] : pza[1];      // partner of source b
        const int ka = RQ * s + rA, kb = RQ * s + rB;
        // plane values (-1)^k' V = X + conj(P), (-1)^k' Vd' = (X - conj(P)) / i; resolve_one takes V, Vd' themselves
        const double sa = (ka & 1) ? -1.0 : 1.0, sb = (kb & 1) ? -1.0 : 1.0;
        if (flagged(ka))
            resolve_one<NWIN, REFRESH, ACC>(disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, ka, j,
                                       sa * (xa[0] + par), sa * (xa[1] - pai), sa * (xa[1] + pai), sa * (par - xa[0]), plane_scale);
        if (flagged(kb))
            resolve_one<NWIN, REFRESH, ACC>(disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, kb, j,
                                       sb * (xb[0] + pbr), sb * (xb[1] - pbi), sb * (xb[1] + pbi), sb * (pbr - xb[0]), plane_scale);
    }
}

constexpr int kTieGroup64 = 24;              // nwin = 128: from this many undecided cells on the group is redone by resolve_group_f64
template <int NWIN, bool REFRESH = false, bool TWLDS = false, bool ACC = false, class Sample>
__device__ __forceinline__ void resolve_bitmap(unsigned* tb, Sample sample, f2* disp_base, int LDF, int* flag, int klo, int K,
                                               f2* own_base, int OLD, int cov0, int cov1,
                                               const double* wtab, const double* twtab, double plane_scale, int lane_in)
{
    constexpr int NSETS = NWIN / 128;                   // sets of 32 words = 64 sources
    int lane = lane_in;                                 // opaque HERE, inside the rare branch: nothing this function derives
    asm volatile("" : "+v"(lane));                      // from the lane id is computed per chunk, held across the transform and spilled
    if constexpr (NWIN == 128 || NWIN == 256 || NWIN == 512) {
        // from kTieGroup64 undecided cells per 64 sources on (and for every exact group): the whole group in float64 at once
        int total = 0;
        if constexpr (!REFRESH) {
            int cnt = 0;
#pragma unroll
            for (int set = 0; set < NSETS; ++set) cnt += (lane < 32) ? __popc(tb[32 * set + lane]) : 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) total += __builtin_popcountll(__builtin_amdgcn_ballot_w64((cnt >> b) & 1)) << b;
        }
        if (REFRESH || total >= kTieGroup64 * NSETS) {
            constexpr int NT = NWIN == 512 ? 32 : 16, RQ = NWIN / NT;
            constexpr int SPP = NWIN == 128 ? 1 : 4;
            if constexpr (NWIN == 128)
                resolve_group_f64_128<REFRESH, TWLDS, ACC>(tb, sample, disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, wtab + 4 * NWIN, twtab,
                                                      plane_scale, lane);
            else
                resolve_group_f64<NT, RQ, SPP, false, REFRESH, TWLDS>(tb, sample, disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1,
                                                                     wtab + 4 * NWIN, twtab, plane_scale, lane);
            for (int i = lane; i < 32 * NSETS; i += 64) tb[i] = 0u;
            if constexpr (REFRESH) {
                if (cov1 > NWIN / 2 && lane < 16) {          // the Nyquist row (see below)
                    double vr = 0.0;
#pragma unroll 1
                    for (int n = 0; n < NWIN; ++n) {
                        const double xw = sample(lane + n) * wtab[2 * n];
                        vr = (n & 1) ? vr - xw : vr + xw;
                    }
                    own_base[lane * OLD + (NWIN / 2 - cov0)] = f2{static_cast<float>(vr * plane_scale), 0.0f};
                }
            }
            if (lane == 0) flag[1] = 0;
            return;
        }
    }
#pragma unroll 1
    for (int set = 0; set < NSETS; ++set) {
        unsigned* tbs = tb + 32 * set;
        const int k0 = 64 * set;
        unsigned w = (lane < 32) ? tbs[lane] : 0u;
        const int cnt = __popc(w);                       // <= 32 per lane: the wave's total from six ballots (scalar unit only)
        int total = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) total += __builtin_popcountll(__builtin_amdgcn_ballot_w64((cnt >> b) & 1)) << b;
        if constexpr (REFRESH) total = 1024;
        if (total == 0) continue;
        if (total <= kTieCoop) {
            // few cells: one by one, all lanes on one cell (NWIN / 64 taps per lane, float64 butterfly sum)
            for (int it = 0; it < total; ++it) {
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(w != 0u);
                const int l0 = __builtin_ctzll(mask);
                const unsigned ww = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(w), l0));
                const int bit = __builtin_ctz(ww);
                if (lane == l0) w &= w - 1u;
                const int kpi = k0 + 2 * l0 + (bit >> 4), jf = bit & 15;
                double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
#pragma unroll 1
                for (int n = lane; n < NWIN; n += 64) {
                    const double x = sample(jf + n);
                    const double2 wd = reinterpret_cast<const double2*>(wtab)[n];
                    const d2v cs = tie_twiddle<TWLDS>(twtab, (kpi * n) & (NWIN - 1));
                    const double xw = x * wd.x, xd = x * wd.y;
                    vr = fma(xw, cs.x, vr); vi = fma(-xw, cs.y, vi);
                    dr = fma(xd, cs.x, dr); di = fma(-xd, cs.y, di);
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    vr += shfl_xor_f64(vr, off, lane); vi += shfl_xor_f64(vi, off, lane);
                    dr += shfl_xor_f64(dr, off, lane); di += shfl_xor_f64(di, off, lane);
                }
                if (lane == 0) resolve_one<NWIN, false, ACC>(disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, kpi, jf, vr, vi, dr, di, plane_scale);
            }
        } else {
            // many cells (tonal or offset-dominated signals under low-sidelobe windows): lane l owns source k' = k0 + l, and
            // the wave walks the FRAMES that have a bit set in some lane.  One frame per round: its windowed samples x w,
            // x dw' are the same for every source, so each lane computes NWIN / 64 of them (coalesced loads) and the round
            // hands them round through v_readlane; per tap a lane then loads only its twiddle (the 16 nwin byte table, L1 /
            // L2).  Same products, same order of additions as the one-by-one branch above.  (The first version let every
            // lane walk its own frames -- three dependent-latency loads per tap and lane: 730 cycles per tap, 23 ms per
            // 1024 windows of an on-bin tone; profiles/r03_input_cost.txt.)
            const unsigned hw = REFRESH ? 0xffffu : (tbs[lane >> 1] >> ((lane & 1) << 4)) & 0xffffu;
            const int kpi = k0 + lane;
            unsigned fm = 0u;                                    // frames with work (wave-uniform)
#pragma unroll
            for (int b = 0; b < 16; ++b) fm |= (__builtin_amdgcn_ballot_w64((hw >> b) & 1u) != 0ull ? 1u : 0u) << b;
            while (fm != 0u) {
                const int jf = __builtin_ctz(fm);
                fm &= fm - 1u;
                constexpr int TPL = NWIN / 64;                   // taps per lane
                double xw[TPL], xd[TPL];
#pragma unroll
                for (int t = 0; t < TPL; ++t) {
                    const int n = lane + 64 * t;
                    const double x = sample(jf + n);
                    const double2 wd = reinterpret_cast<const double2*>(wtab)[n];
                    xw[t] = x * wd.x; xd[t] = x * wd.y;
                }
                double vr = 0.0, vi = 0.0, dr = 0.0, di = 0.0;
#pragma unroll
                for (int t = 0; t < TPL; ++t) {
#pragma unroll 8
                    for (int l = 0; l < 64; ++l) {
                        const int n = 64 * t + l;
                        const double a = readlane_f64(xw[t], l), b = readlane_f64(xd[t], l);
                        const d2v cs = tie_twiddle<TWLDS>(twtab, (kpi * n) & (NWIN - 1));
                        vr = fma(a, cs.x, vr); vi = fma(-a, cs.y, vi);
                        dr = fma(b, cs.x, dr); di = fma(-b, cs.y, di);
                    }
                }
                if ((hw >> jf) & 1u) resolve_one<NWIN, REFRESH, ACC>(disp_base, LDF, flag, klo, K, own_base, OLD, cov0, cov1, kpi, jf, vr, vi, dr, di, plane_scale);
            }
        }
        if (lane < 32) tbs[lane] = 0u;
    }
    if constexpr (REFRESH) {
        // the Nyquist row (its own partner, never moves: V = sum x w (-1)^n is real) when the stored cover holds it
        if (cov1 > NWIN / 2 && lane < 16) {
            double vr = 0.0;
#pragma unroll 1
            for (int n = 0; n < NWIN; ++n) {
                const double xw = sample(lane + n) * wtab[2 * n];
                vr = (n & 1) ? vr - xw : vr + xw;
            }
            own_base[lane * OLD + (NWIN / 2 - cov0)] = f2{static_cast<float>(vr * plane_scale), 0.0f};
        }
    }
    if (lane == 0) flag[1] = 0;
}
template <int NWIN>
__device__ __forceinline__ void resolve_ties(int* tq, const float* xg, f2* disp_base, int LDF, int* flag, int klo, int K,
                                             f2* own_base, int OLD, int cov0, int cov1,
                                             const double* wtab, const double* twtab, int lane)
{
    resolve_bitmap<NWIN>(reinterpret_cast<unsigned*>(tq), [xg](int i) -> double { return static_cast<double>(xg[i]); }, disp_base, LDF, flag,
                         klo, K, own_base, OLD, cov0, cov1, wtab, twtab, 1.0, lane);
}

// One one-sided source bin k' held as packed spectrum value X = Z[k'] with conjugate partner
// P = Z[128 - k'] (Z = FFT of x (w + i dw') / 2 with the sign (-1)^k' folded into the constants):
//   V = X + conj(P) = (-1)^k' V[k'],   Vd' = (X - conj(P)) / i,   shift = -Im(Vd'/V) = num / den.
// The source stays in its own row iff |shift| < 1/2 iff |num| < den/2 (no division); den carries +1e-37
// so that V == 0 (which contributes nothing wherever it lands) never takes the rare path.  V is stored
// unconditionally into the source's own column; only when some lane of the wave has a displaced cell does
// the wave run the exact rounding path for it.  (Folding the 1/2 into the constants by doubling dw' saves
// one more multiply per source, 1 % of the kernel, but doubles the cancellation error of V for far-moving
// cells: measured 5 instead of 3 rounding flips per 918 k robust Hann columns, so it is not done.)
// The two sources of one stripe (classes a and b of this lane) with ONE rare-path branch for both (a branch per
// source -- v_cmp + s_and_saveexec + s_cbranch + s_or each -- measured 1.8 % slower; one branch per two stripes: no
// further gain).
template <int S, int RQ, int NWIN, bool LIST = false>
__device__ __forceinline__ void process_stripe(f2 XA, f2 PA, f2 XB, f2 PB, f2 tiny, f2* ownA, f2* ownB, bool store,
                                               f2* row_disp, int* flag, int* tq, int j, int klo, int K, int rA, int rB, float R2, float& mx,
                                               PairList* pl = nullptr)
{
    const f2 a1 = mix_re(XA, PA), a2 = mix_im(XA, PA);
    const f2 b1 = mix_re(XB, PB), b2 = mix_im(XB, PB);
    const f2 dna = dn_second(a2, dn_first(a1, tiny)), dnb = dn_second(b2, dn_first(b1, tiny));
    if (store) {                                        // wave-uniform predicate
        float* qa = reinterpret_cast<float*>(ownA);
        float* qb = reinterpret_cast<float*>(ownB);
        qa[0] = a1.x; qa[1] = a2.x;
        qb[0] = b1.x; qb[1] = b2.x;
        // largest |V|^2 among the stored cells ("Exact groups"); in the cover's FIRST stripe only rows of the band count:
        // an offset (row 0 and its main lobe) sits there, below the band, and is exactly what must not pass for content
        if (RQ * S <= klo) mx = fmaxf(mx, fmaxf(rA + RQ * S >= klo ? dna.x : 0.0f, rB + RQ * S >= klo ? dnb.x : 0.0f));
        else mx = fmaxf(mx, fmaxf(dna.x, dnb.x));
    }
    // (the threshold sits kTieMargin below 1/2 so that a cell whose |shift| is within the margin of 1/2 -- a rounding
    //  tie as well -- reaches the rare path and its float64 decision)
#ifdef HSS_NO_TIES
    constexpr float kStay = 0.5f;
#else
    constexpr float kStay = 0.5f - kTieMargin;
#endif
    const bool ma = fabsf(dna.y) >= kStay * dna.x, mb = fabsf(dnb.y) >= kStay * dnb.x;
    if (ma | mb) {                                      // skipped when no lane moved (execz)
        if (ma) displaced_source<NWIN, 1, LIST>(row_disp, flag, tq, klo, K, rA + RQ * S, j, dna.y, dna.x, f2{a1.x, a2.x}, R2, ownA, store, pl);
        if (mb) displaced_source<NWIN, 1, LIST>(row_disp, flag, tq, klo, K, rB + RQ * S, j, dnb.y, dnb.x, f2{b1.x, b2.x}, R2, ownB, store, pl);
    }
}
// ------------------------------------------------------------------------------------------------
// grid = persistent blocks of WPB waves (see "Work distribution" above); each WAVE draws chunks of one signal from
// the block's LDS counter, stages FPW + 127 samples per FPW frames and walks them in groups of 16 frames,
// independently of its sibling waves (no block barrier after the prologue).
// LDS: atab[16 taps][64 lanes][2] (shared, 8 KB) | control words + FAST store-offset table | per wave: xs[FPW+127] |
//      own plane | displaced plane | flag.
// ------------------------------------------------------------------------------------------------
// FAST: the time-major [re | im] epilogue with 16-byte stores (mode STACK / STACK_UNNORM, K even, K <= 24 -- the
// canonical configuration); otherwise the general epilogue (raw / abs / any K).  The host picks.
// WPB: waves per block -- 16 (a whole CU) whenever 16 wave regions fit the 160 KB of LDS, else 8 / 4 / 2 / 1.
// NT, RQ: taps (= size of the per-lane FFT) and radix of the first (matrix-pipe) stage, nwin = NT RQ.  (16, 8) is the
// canonical nwin = 128; (16, 16) = nwin 256 runs the same 8-class structure twice per group of frames ("passes":
// classes {0,8,1,15,2,14,3,13}, then {4,12,...,7,9}), each tap as a chain of RQ / 4 MFMA k-steps; (32, 16) = nwin 512
// does the same with 32 taps and 32-point FFTs per lane (two waves per SIMD, up to 256 VGPRs).
// S1C >= 0: the host guarantees that the kept band starts in stripe 0 and ends in stripe S1C (the canonical band
// [25, 200] Hz at fs = 1000 is stripes 0..3 for every RQ); the per-source "does this stripe have a column in the own
// plane" tests and their branches are then compile-time (measured 3.3-3.6 % of the kernel; making the whole band
// (klo, K) a compile-time constant gave nothing more).  S1C = -1: any band.
// FUSED (z-score inside the core launch; STACK, FAST epilogue, signals of at most kFusedMaxGroups groups): see the section
// "Fused z-score" inside the kernel.
using gu32 = __attribute__((address_space(1))) unsigned;
constexpr unsigned kSpinLimit = 1u << 18;          // polls before a wait gives up (a poll takes ~0.1 us: ~30 ms; a healthy
                                                   // wait is microseconds).  After the first give-up of a block its other
                                                   // waits give up at once (LDS word `dead`), so a broken launch ends fast

// STREAM (one step of the rolling transform, BASELINE config 5; FAST epilogue, un-normalised): ONE launch per step instead of
// copy + transform + merge-and-normalise.  Blocks are bound to channels (bpc blocks per channel, block `part` takes the groups
// part, part + bpc, ... one group per ticket, staged exactly as the one-group chunks of the plain kernel: same tiles, same bits);
// a group's samples come from the tape or, from index hist on, straight from the step's new samples, which the group's wave also
// appends to the tape (nobody reads the tape there during the step); the block that delivers last for its channel (one counter
// per channel in HBM) runs the arithmetic of fsst_stream_finish_kernel on the channel's chunk.
#ifdef HSS_STREAM_PROBE      // development (tools/stream_probe.py): 100 MHz ticks from a wave's start to its phase boundaries, kept per wave, written at the end
constexpr int kStreamProbeWaves = 2048;
__device__ unsigned long long g_stream_probe[kStreamProbeWaves * 8];      // [wave of the last launch][stamp]
#define SPROBE(k) do { if constexpr (STREAM) sprobe[k] = wall_clock64() - sprobe_t0; } while (0)
#else
#define SPROBE(k) do { } while (0)
#endif
// PAIR (nwin 256 / 512, whose transform is two passes over the same 16 frames): TWO waves share one wave region -- wave 2 r does
// pass 0 of region r's group, wave 2 r + 1 pass 1, at the same time; the passes write disjoint columns of the own plane, and the
// odd wave LISTS its additions into the displaced plane instead of making them (PairList): the even wave replays the list behind
// its own additions, so the plane receives every addition in the order of the one-wave kernel -- the same bits whoever runs
// faster.  The even wave then runs everything that follows the passes.  The pair
// meets through two phase words in LDS (pair_sync: LDS operations of a wave are executed in order, so a phase word written
// after a wave's data is seen after it): twice the waves on the same LDS -- 6 instead of 3 per CU for nwin 512 with 90 kept
// rows -- and half the latency of a lone group (one streaming step).
constexpr unsigned kPairSpinLimit = 1u << 24;       // looks at the partner's phase word before a pair's wait ends on its own (~1 s)
constexpr int kPairFloats = 4 + 64 + 3 * 256;  // [0..1] phase words, [2] the pair's ticket, [3] the odd wave's list count, [4..67] its per-lane
                                             // max |V|^2, then its list of additions (PairList: 256 cells, 256 values)
template <int NT, int RQ, int FPW, bool FAST, int WPB, int S1C, bool FUSED = false, bool STREAM = false, bool PAIR = false>
__global__ __launch_bounds__(64 * WPB, (NT == 32 ? 2 : WPB == 16 ? HSS_MW128 : WPB == 12 ? 3 : 2)) void fsst_core128_kernel(Core128Params p)
{
    static_assert(!STREAM || (FAST && !FUSED), "the streaming step: wide-store epilogue, no per-signal z-score");
    static_assert(!PAIR || (RQ == 16 && !FUSED && WPB % 2 == 0), "wave pairs: two passes, whole pairs");
    // (FUSED && !FAST: the general epilogue's STACK mode -- any K, nwin 256 / 512 -- with the linear z-score sweep of
    //  fsst_normalize_kernel as the B ticket; the host sends only STACK execs whose signal blocks are 16-byte aligned)
    constexpr int NWIN = NT * RQ, NPASS = RQ / 8, KST = RQ / 4;
    constexpr int ATAB = core128_atab_floats(RQ, NT);
    // (FUSED && !FAST: the plain control block -- the statistics partials stay in HBM as on the two-launch path, the wave
    //  regions of nwin 256 / 512 leave no LDS for them --, resolved statistics at [16..31])
    constexpr int CTL = (FUSED && FAST) ? kCtlFusedFloats : kCtlFloats;
    constexpr int XS = ((FPW + NWIN - 1 + 3) / 4) * 4;
    using avec = float __attribute__((ext_vector_type(KST)));          // one tap's A operand, all k-steps
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.K, klo = p.klo, n = p.n;
    const int LDF = plane_ldf(K);
    const int OLD = own_ld(klo, K, RQ);
    const int s0 = (S1C >= 0) ? 0 : own_s0(klo, RQ), s1 = (S1C >= 0) ? S1C : own_s1(klo, K, RQ);

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
#ifdef HSS_STREAM_PROBE
    const unsigned long long sprobe_t0 = wall_clock64();
    unsigned long long sprobe[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#ifdef HSS_CLOCKPROBE
    const unsigned long long probe_c0 = __builtin_readcyclecounter(), probe_r0 = wall_clock64();
#endif
    float* atab = smem;                                                      // [pass][NT taps][64 lanes][KST]
    int* next_q = reinterpret_cast<int*>(smem + ATAB);                       // block's work counter
    unsigned* done_a = reinterpret_cast<unsigned*>(smem + ATAB) + 1;         // FUSED: [2] groups delivered (monotone)
    unsigned* dead = done_a + 2;                                             // FUSED: a wait of this block gave up
    unsigned* ready = dead + 1;                                              // FUSED: [4] epoch of fin_stats[]
    float4* fin_stats = reinterpret_cast<float4*>(smem + ATAB + (FAST ? 272 : 16));   // FUSED: [4] resolved statistics
    unsigned* cls_lds = reinterpret_cast<unsigned*>(smem + ATAB + 16);       // FUSED: [64] see "cls" below
    unsigned* ppk_lds = reinterpret_cast<unsigned*>(smem + ATAB + (FUSED ? 80 : 16));   // [3][64] wide-store offsets (FAST)
    float* part_lds = smem + ATAB + 288;                                     // FUSED: [2][kFusedMaxGroups][kPartFloats]
    const int role = PAIR ? (wv & 1) : 0;                                    // PAIR: the pass this wave does
    float* wbase = smem + ATAB + CTL + (PAIR ? (wv >> 1) : wv) * (wave_lds_floats(FPW, klo, K, RQ, NT) + (PAIR ? kPairFloats : 0));
    float* xs = wbase;
    f2* own_base = reinterpret_cast<f2*>(wbase + XS);
    f2* disp_base = own_base + 16 * OLD;
    int* flag = reinterpret_cast<int*>(disp_base + 16 * LDF);
    int* tq = flag + 4;                                                       // rounding-tie bitmap (tie_words(NWIN))
    int* pw = tq + tie_words(NWIN);                                           // PAIR: phase words, ticket
    float* pmx = reinterpret_cast<float*>(pw + 4);                            // PAIR: the odd wave's per-lane max
    PairList plist{pw + 4 + 64, reinterpret_cast<f2*>(pw + 4 + 64 + kPairListCap), pw + 3, 0};      // PAIR: the odd wave's additions
    bool pordered = false;                               // PAIR: a list overflowed: this pair forms the odd wave's sources behind the even wave's
    if constexpr (PAIR) { if (lane < 4) pw[lane] = 0; }
    int pphase = 0;
    auto pair_sync = [&]() {
        if constexpr (PAIR) {
            wave_sync();
            ++pphase;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(pw + role, pphase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // (bounded like every wait of this library: the partner is a wave of the same workgroup and cannot stay away, but a
            //  kernel that could spin for ever is a kernel that can hang a GPU)
            bool met = false;
            for (unsigned spins = 0; spins < kPairSpinLimit; ++spins) {
                int v = 0;
                if (lane == 0) v = __hip_atomic_load(pw + (role ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(v) - pphase >= 0) { met = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            // (a wait that ran out: the wave goes on -- on planes its partner may still be writing -- but not silently: the status word
            //  makes hssfsst_plan_check / the next exec / the next streaming step fail, as the bounded waits of the other kernels do)
            if (!met && lane == 0 && p.status) __hip_atomic_store((gu32*)(p.status), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            wave_sync();
        }
    };

    // STREAM: tickets are dealt statically (wave region r takes the block's tickets r, r + regions, ...), so the first group's
    // samples can be on their way while the operand table is staged
    constexpr int NREG = PAIR ? WPB / 2 : WPB;
    constexpr int NPRE = (FPW + NWIN - 1 + 63) / 64;
    float pre[NPRE];
    int st_q = PAIR ? (wv >> 1) : wv;
    auto stream_sample = [&](long long ch, int gi) -> float {     // sample gi of the step's hist + chunk samples of channel ch
        const float* src = (p.xnew != nullptr && gi >= p.hist) ? p.xnew + ch * p.xnew_stride + (gi - p.hist) : p.x + ch * p.xstride + gi;
        return *src;
    };
    if constexpr (STREAM) {
        const int ch0 = static_cast<int>(blockIdx.x) / p.bpc, g0 = static_cast<int>(blockIdx.x) - ch0 * p.bpc + st_q * p.bpc;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int gi = g0 * 16 + lane + 64 * k;          // (t0 - NWIN / 2 = 16 g0: the step's frames start at col0 = NWIN / 2)
            pre[k] = (g0 * 16 < p.ncols && lane + 64 * k < FPW + NWIN - 1 && gi < p.n) ? stream_sample(ch0, gi) : 0.0f;
        }
    }
    // shared MFMA A operand, [pass * NT + tap][lane][k-step]: a lane reads all k-steps of a tap with one LDS instruction
    // (the host lays the table out like this: a straight 16-byte copy, 64 KiB of it for nwin 512)
    if constexpr ((ATAB / 4) % (64 * WPB) == 0 && ATAB / 4 / (64 * WPB) <= 16) {
        constexpr int NA = ATAB / 4 / (64 * WPB);            // all of a thread's loads in flight at once (one memory round trip)
        float4 t[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) t[k] = reinterpret_cast<const float4*>(p.atab)[threadIdx.x + k * 64 * WPB];
#pragma unroll
        for (int k = 0; k < NA; ++k) reinterpret_cast<float4*>(atab)[threadIdx.x + k * 64 * WPB] = t[k];
    } else {
        for (int i = threadIdx.x; i < ATAB / 4; i += 64 * WPB)
            reinterpret_cast<float4*>(atab)[i] = reinterpret_cast<const float4*>(p.atab)[i];
    }
    const int ncols = p.ncols, cend = p.col0 + p.ncols;   // output rows are relative to col0
    for (int i = lane; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
    if (lane < 4) flag[lane] = 0;
    for (int i = lane; i < tie_words(NWIN); i += 64) tq[i] = 0;
    if (threadIdx.x < (FUSED ? 8 : 1)) next_q[threadIdx.x] = 0;
    if constexpr (FUSED && FAST) {
        if (wv == 0) {
            // bit 2i / 2i+1 of cls: the first / second pair of this lane's float4 i (of a group's [16][2K] image) is an
            // imaginary column
            unsigned cls = 0u;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned c = (4u * static_cast<unsigned>(lane + 64 * i)) % static_cast<unsigned>(2 * K);
                cls |= (c >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i);
                cls |= (c + 2 >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i + 1);
            }
            cls_lds[lane] = cls;
        }
    }
    // wide-store epilogue (time-major [re | im] rows, K even, <= 3 float4 per lane and group): every lane's six LDS byte
    // offsets of the store pass (host-made table, core128_store_offsets), two 16-bit offsets per word
    if constexpr (FAST) {
        if (wv == 0) {
            const int* ptab = reinterpret_cast<const int*>(p.atab + ATAB);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                ppk_lds[i * 64 + lane] = static_cast<unsigned>(ptab[i * 64 + lane]) | (static_cast<unsigned>(ptab[(3 + i) * 64 + lane]) << 16);
        }
    }
    __syncthreads();
    SPROBE(0);

    // chunk bookkeeping (wave-uniform)
    const int nc0 = p.nsig * p.reg.npc[0];               // (the host keeps nsig * chunks per signal below 2^31)
    const int nc1 = nc0 + p.nsig * p.reg.npc[1];
    const int nchunks = nc1 + p.nsig * p.reg.npc[2];
    const int ngroups = (ncols + 15) >> 4;
    // FUSED work list of this block (wave-uniform): signals blockIdx, blockIdx + grid, ... (nk of them), each cut into
    // NC chunks of FPW / 16 groups; see "Fused z-score" below for the order of the 2 NC nk tickets
    constexpr int GPCF = FPW / 16;
    const int nk = (FUSED && p.nsig > static_cast<int>(blockIdx.x)) ? (p.nsig - static_cast<int>(blockIdx.x) + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x) : 0;
    const int NC = (ngroups + GPCF - 1) / GPCF;
    const int lead = min(8, NC);
    const int st_ch = STREAM ? static_cast<int>(blockIdx.x) / p.bpc : 0;              // STREAM: this block's channel and
    const int st_part = STREAM ? static_cast<int>(blockIdx.x) - st_ch * p.bpc : 0;    // its first group
    const int nwork = STREAM ? (st_part < ngroups ? (ngroups - st_part + p.bpc - 1) / p.bpc : 0) : FUSED ? 2 * NC * nk : nchunks;
    auto draw = [&]() -> int {                           // next work item of this block, or nwork when none is left
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = __builtin_amdgcn_readfirstlane(q);
        if constexpr (FUSED || STREAM) return q < nwork ? q : nwork;
        const long long c = static_cast<long long>(blockIdx.x) + static_cast<long long>(q) * gridDim.x;
        return c < nwork ? static_cast<int>(c) : nwork;
    };
    auto draw_pair = [&]() -> int {                      // PAIR: the even wave draws, the pair takes the ticket together
        if constexpr (STREAM) { const int q = st_q; st_q += NREG; return q < nwork ? q : nwork; }
        if constexpr (!PAIR) return draw();
        if (role == 0) { const int c = draw(); if (lane == 0) pw[2] = c; }
        pair_sync();
        return __builtin_amdgcn_readfirstlane(pw[2]);
    };
    int chunk = draw_pair();
    bool first_tile = STREAM;                            // STREAM: the first ticket's samples are in `pre`
    const float* myA = atab + lane * KST;
    f2 tiny = {1.0e-37f, 0.0f};
    asm volatile("" : "+s"(tiny));                       // keep it in an SGPR pair (VOP3P takes no literal)
    while (chunk < nwork) {
    // decode: signal, first group, number of groups
    long long b;
    int grp0, ngrp;
    long long ksig = 0;                                  // FUSED: index of the signal in this block's list
    bool zpass = false;                                  // FUSED: this ticket is a z-score chunk
    if constexpr (FUSED) {
        // ------------------------------------------------------------------------------------------------
        // Fused z-score (FSST._stack_real_imag, synchrosqueeze.py:78-85) inside the core launch.
        // One CU owns whole signals; nothing crosses CUs (no flags in HBM, no placement assumption, no way to hang on
        // another block).  Its 16 waves draw TICKETS from one LDS counter.  A ticket is either a transform chunk A(k, c)
        // -- FPW / 16 groups of the block's k-th signal: transform, un-normalised features to HBM with ordinary stores,
        // one statistics partial per group into LDS -- or a z-score chunk B(k, c): read the same chunk back, z-score it
        // with the arithmetic of fsst_normalize_kernel, store it for good (streaming).  Ticket order: all A(0, .); then
        // for k = 1 .. nk-1 the first `lead` chunks of A(k, .), then B(k-1, 0), A(k, lead), B(k-1, 1), A(k, lead+1), ...
        // and what is left of B(k-1, .); finally B(nk-1, .).  The wave that delivers the last group of a signal turns
        // the signal's partials into {mean, 1/std} with signal_stats() -- the very function, order and data of the
        // two-kernel path, hence bit-identical results -- while its siblings already transform the next signal; a B
        // ticket waits (bounded) for that, which in the steady state it never has to: the B tickets of a signal start
        // `lead` chunks after its last A ticket was handed out.
        // Visibility: the un-normalised tile is written and read back by waves of ONE workgroup; the chain writer
        // (release on the delivery counter) -> resolver (acquire / release on `ready`) -> reader (acquire) is
        // workgroup scope, the waves share the CU's L1, and a line of `out` is read by this kernel exactly once.
        // Slots: the partials of signal k live in LDS slot k & 1, written by the A(k, .) tickets and read once by the
        // resolver of k.  A(k+2, .) tickets are handed out only after ALL NC tickets B(k, .) were handed out, and a B(k, .)
        // ticket is held (waiting) until the resolver of k is done; the 15 waves other than the resolver cannot hold NC
        // >= 16 of them, so slot reuse is safe by construction -- which is why the host sends signals of fewer than
        // kFusedMinChunks chunks down the two-kernel path.  {mean, 1/std} are copied to registers at the start of a B
        // ticket and live in 4 slots.
        // (An earlier variant -- teams of 8 CUs per signal with the tile meant to stay in the XCD's L2 -- was built,
        // bit-identical and slower; PMC showed that the L2 does not retain the written lines:
        // profiles/r02_fused_team_variant.txt.)
        // ------------------------------------------------------------------------------------------------
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
    } else if constexpr (STREAM) {
        b = st_ch; grp0 = st_part + chunk * p.bpc; ngrp = 1;
    } else {
        const int rg = (chunk < nc0) ? 0 : (chunk < nc1) ? 1 : 2;
        const int local = chunk - ((rg == 0) ? 0 : (rg == 1) ? nc0 : nc1);
        const int npc = p.reg.npc[rg], gpc = p.reg.gpc[rg];
        b = local / npc;
        const int cidx = local - static_cast<int>(b) * npc;
        grp0 = p.reg.g0[rg] + cidx * gpc;
        ngrp = min(gpc, ngroups - grp0);
    }
    // opaque copies of the lane coordinates for the HBM addressing below: otherwise the per-lane 64-bit address
    // parts are hoisted out of the chunk loop, held in registers across it and spilled to scratch (a kernel with
    // scratch costs isolated launches ~200 us on this runtime)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    int g_o = lane_o >> 4, j_o = lane_o & 15;
    asm volatile("" : "+v"(g_o), "+v"(j_o));
    const int g = lane_o >> 4, j = lane_o & 15;          // (re-derived per work item: two registers fewer across the loop)
    f2* row_disp = disp_base + j * LDF;
    if (FUSED && zpass) {
        // ---- B(ksig, c): z-score of ngrp groups of a signal whose statistics are (about to be) in LDS
        const int sl = static_cast<int>(ksig) & 3;
        const unsigned epoch = static_cast<unsigned>(ksig) + 1u;
        const int C = 2 * K;
        float4* d4 = reinterpret_cast<float4*>(p.out + (b * static_cast<long long>(ncols) + grp0 * 16) * C) + lane_o;
        const int per = 8 * K;                           // float4 per full group = 16 * 2K / 4
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
        const float4 st = fin_stats[sl];
        if constexpr (!FAST) {
            // general K: the chunk's frames x 2K floats are one contiguous, 16-byte aligned run of `out`; float4 sweep with
            // the column tracked incrementally and a per-element wrap test, the arithmetic of fsst_normalize_kernel
            // (kZB float4 per lane in flight at a time: one memory round trip per 64 kZB float4)
            constexpr int kZB = 12;
            const int frames = min(ngrp * 16, ncols - grp0 * 16);
            const int nfl = frames * C;                      // even
            const int nf4 = nfl >> 2;
            f4* b4 = reinterpret_cast<f4*>(p.out + (b * static_cast<long long>(ncols) + grp0 * 16) * C);
            int c = static_cast<int>((4u * static_cast<unsigned>(lane_o)) % static_cast<unsigned>(C));
            const int dc = static_cast<int>(256u % static_cast<unsigned>(C));
            auto zs1 = [&](float v, int col) -> float { return (col < K) ? (v - st.x) * st.y : (v - st.z) * st.w; };
            for (int i0 = 0; i0 < nf4; i0 += 64 * kZB) {
                f4 o[kZB];
#pragma unroll
                for (int u = 0; u < kZB; ++u) {
                    const int idx = i0 + 64 * u + lane_o;
                    o[u] = __builtin_nontemporal_load(b4 + (idx < nf4 ? idx : 0));
                }
#pragma unroll
                for (int u = 0; u < kZB; ++u) {
                    const int idx = i0 + 64 * u + lane_o;
                    int c1 = c + 1, c2 = c + 2, c3 = c + 3;
                    if (c1 >= C) c1 -= C;
                    if (c2 >= C) c2 -= C;
                    if (c3 >= C) c3 -= C;
                    const f4 r = {zs1(o[u].x, c), zs1(o[u].y, c1), zs1(o[u].z, c2), zs1(o[u].w, c3)};
                    if (idx < nf4) __builtin_nontemporal_store(r, b4 + idx);
                    c += dc;
                    if (c >= C) c -= C;
                }
            }
            if ((nfl & 2) && lane_o == 0) {                  // frames x 2K = 2 mod 4: the last two floats of the chunk
                float* tl = p.out + (b * static_cast<long long>(ncols) + grp0 * 16) * C + (nfl - 2);
                const int ct = C - 2;                        // (they end a row)
                tl[0] = zs1(tl[0], ct);
                tl[1] = zs1(tl[1], ct + 1);
            }
        } else {
        const unsigned cls = cls_lds[lane_o];
        // One memory round trip per ticket: all 12 pieces (4 groups x 3 float4) of the chunk in flight at once.  The loads
        // are unconditional (a lane without a piece re-reads the chunk's first float4): conditionally defined registers
        // made the allocator spill here although 100 registers are free.
        const float4* base0 = reinterpret_cast<const float4*>(p.out + (b * static_cast<long long>(ncols) + grp0 * 16) * C);
        auto glim = [&](int q) { return (q < ngrp) ? min(16, ncols - (grp0 + q) * 16) * (K >> 1) : 0; };
        {
            f4 o[GPCF][3];
#pragma unroll
            for (int q = 0; q < GPCF; ++q) {
                const int lim = glim(q);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float4* src = (lane_o + 64 * i < lim) ? (d4 + q * per + 64 * i) : base0;
                    // streaming loads: read once, must not displace anything (measured 0.247 vs 0.261 ms per launch
                    // with ordinary loads; streaming stores in the transform chunks on top of that give the gain back)
                    o[q][i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src));
                }
            }
            // this lane's {mean, 1/std} per float4 and pair, once per ticket (the column classes do not depend on the group)
            f2 mu[3][2], rs[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bool im0 = (cls >> (2 * i)) & 1u, im1 = (cls >> (2 * i + 1)) & 1u;
                const float m0 = im0 ? st.z : st.x, r0 = im0 ? st.w : st.y;
                const float m1 = im1 ? st.z : st.x, r1 = im1 ? st.w : st.y;
                mu[i][0] = f2{m0, m0}; rs[i][0] = f2{r0, r0};
                mu[i][1] = f2{m1, m1}; rs[i][1] = f2{r1, r1};
            }
#pragma unroll
            for (int q = 0; q < GPCF; ++q) {
                const int lim = glim(q);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (lane_o + 64 * i < lim) {
                        const f2 lo = (f2{o[q][i].x, o[q][i].y} - mu[i][0]) * rs[i][0];     // (v - mean) * (1 / std): two roundings,
                        const f2 hi = (f2{o[q][i].z, o[q][i].w} - mu[i][1]) * rs[i][1];     // exactly as fsst_normalize_kernel
                        __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(d4 + q * per + 64 * i));
                    }
                }
            }
        }
        }
    } else {
    const float* xsig = p.x + b * p.xstride;
    auto stage_tile = [&](int t0) -> TileEnergy {        // xs[i] = xpad[t0 + i] = x[t0 + i - 64]; the tile's energy / offset test
        float e = 0.0f, s1 = 0.0f, cnt = 0.0f;
        int lane_t = lane;                               // (opaque per tile: the tile's lane addresses are not hoisted out of
        asm volatile("" : "+v"(lane_t));                 //  the chunk loop, held across the transform and spilled)
        if constexpr (STREAM) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {                 // (the same elements in the same order as the loop below)
                const int i = lane_t + 64 * k;
                if (i < FPW + NWIN - 1) {
                    const int gi = t0 + i - NWIN / 2;
                    const bool in = (gi >= 0 && gi < n);
                    const float v = first_tile ? pre[k] : (in ? stream_sample(b, gi) : 0.0f);
                    xs[i] = v;
                    e = fmaf(v, v, e); s1 += v; cnt += in ? 1.0f : 0.0f;
                }
            }
            first_tile = false;
            return tile_energy(e, s1, cnt);
        }
        for (int i = lane_t; i < FPW + NWIN - 1; i += 64) {
            const int gi = t0 + i - NWIN / 2;
            const bool in = (gi >= 0 && gi < n);
            const float v = in ? xsig[gi] : 0.0f;
            xs[i] = v;
            e = fmaf(v, v, e); s1 += v; cnt += in ? 1.0f : 0.0f;
        }
        return tile_energy(e, s1, cnt);
    };
    for (int sub = 0; sub < ngrp; sub += FPW / 16) {
    const int t0 = p.col0 + (grp0 + sub) * 16;
    // R^2 of the tile's frames for the error bound of displaced cells (see "Rounding ties")
    const TileEnergy te = stage_tile(t0);
    const float R2 = p.r2scale * te.E;
    wave_sync();
    SPROBE(5);
    if constexpr (STREAM) {
        // the group's 16 new samples (hist + 16 grp0 + lane: the last sample of its frame `lane`) go to the tape; the groups of a
        // channel cover the chunk once
        if (p.xnew != nullptr && role == 0 && lane < 16 && grp0 * 16 + lane < ncols)
            const_cast<float*>(xsig)[p.hist + grp0 * 16 + lane] = p.xnew[b * p.xnew_stride + grp0 * 16 + lane];
    }
    const int gend = min(FPW / 16, ngrp - sub);
    for (int grp = 0; grp < gend; ++grp) {
    const int tg = t0 + grp * 16;
        const int tr = tg - p.col0;
        // the tile's LDS byte address as ONE opaque register: every tap is then an immediate offset of it
        // (otherwise each merged ds_read2 gets its own "base + 0x2000 + tap" v_add).  Lane (kk = lane >> 4, f = lane & 15)
        // is row kk of the B operand for frame f: sample f + tap + NT (kk + 4 ks); for NT = 16 that is lane + tap + 64 ks.
        unsigned xaddr = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)(xs + grp * 16 + j + NT * g)));
        asm volatile("" : "+v"(xaddr));
        const lds_float* xb = (const lds_float*)static_cast<size_t>(xaddr);
        float mx = 0.0f;
        // ---- NPASS passes over the same 16 frames: pass pz handles class pairs 4 pz + g (pair 0 = the two
        //      self-conjugate classes {0, RQ/2}, pair m = {m, RQ - m})
        auto one_pass = [&](auto PZ) {
        constexpr int pz = decltype(PZ)::value;
        // keep the per-lane class ids opaque inside the loop: otherwise LICM hoists every
        // "rA + RQ s" of the rare path out of the loop and pins ~30 VGPRs for the whole kernel
        int pair = 4 * pz + g;
        asm volatile("" : "+v"(pair));
        const bool isg0 = (pz == 0) && (pair == 0);          // lane holds the self-conjugate pair
        const int rAi = pair, rBi = isg0 ? RQ / 2 : RQ - pair;
        f2* ownA = own_base + j * OLD + rAi - RQ * s0;       // column of k' = RQ s + rA at + RQ s
        f2* ownB = own_base + j * OLD + rBi - RQ * s0;
        const float* myAp = myA + pz * NT * 64 * KST;

        // ---- folded window + radix-RQ stage on the matrix pipe: NT taps x KST k-steps
        f2 za[NT], zb[NT];
        // four taps at a time, k-step by k-step (the dependent MFMA of a tap is issued three MFMAs after the
        // previous k-step of the same tap; measured 1 % faster than tap by tap)
        static_for<NT / 4>([&](auto GG) {
            constexpr int g0 = decltype(GG)::value * 4;
            f4 acc[4];
            avec a2[4];
            static_for<4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                a2[i] = *reinterpret_cast<const avec*>(myAp + (g0 + i) * 64 * KST);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i][0], xb[g0 + i], f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
            });
            static_for<KST - 1>([&](auto KS) {
                constexpr int ks = decltype(KS)::value + 1;
                __builtin_amdgcn_sched_barrier(0);
                static_for<4>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i][ks], xb[g0 + i + 4 * NT * ks], acc[i], 0, 0, 0);
                });
            });
            static_for<4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                za[bitrev_n<NT>(g0 + i)] = f2{acc[i].x, acc[i].y};
                zb[bitrev_n<NT>(g0 + i)] = f2{acc[i].z, acc[i].w};
            });
        });
#if !defined(HSS_ABLATE) || HSS_ABLATE < 4
        fft_n<NT>(za);
        fft_n<NT>(zb);
#endif
#if defined(HSS_ABLATE) && HSS_ABLATE >= 3
        {   // development only: keep the spectra alive without the source stage
            f2 acc = {0.0f, 0.0f};
            static_for<NT>([&](auto I) { acc += za[decltype(I)::value] + zb[decltype(I)::value]; });
            if (s1 >= 0 && isg0) own_base[j * OLD] = acc;
        }
#else
        // ---- one-sided sources of this lane: classes rA (array a) and rB (array b)
        bool listed = false;
        if constexpr (PAIR && pz == 1) {
            if (!pordered) {
                // the odd wave's sources at the same time as the even wave's: its additions into the displaced plane go to a list
                // (PairList) that the even wave replays behind its own -- the order of the one-wave kernel, the same bits
                listed = true;
                plist.rowoff = j * LDF;
                if (lane == 0) pw[3] = 0;
                wave_sync();
        static_for<NT / 2>([&](auto SS) {
                constexpr int s = decltype(SS)::value;
                // partner of a[s]: class 0 -> a[(NT-s) mod NT];  else b[NT-1-s].  partner of b[s]: class RQ/2 -> b[NT-1-s]; else a[NT-1-s]
                const f2 pa0 = za[(NT - s) & (NT - 1)], pb = zb[NT - 1 - s], pa = za[NT - 1 - s];
                f2 PA, PB;
                if constexpr (pz == 0) {
                    PA = f2{isg0 ? pa0.x : pb.x, isg0 ? pa0.y : pb.y};
                    PB = f2{isg0 ? pb.x : pa.x, isg0 ? pb.y : pa.y};
                } else {                                         // no self-conjugate class in the later passes
                    PA = pb; PB = pa;
                }
                const bool st = (s >= s0) && (s <= s1);
                process_stripe<s, RQ, NWIN, true>(za[s], PA, zb[s], PB, tiny, ownA + RQ * s, ownB + RQ * s, st, row_disp, flag, tq, j, klo, K, rAi, rBi, R2, mx, &plist);
            });
            } else {
                // (a list overflowed earlier: the odd wave forms its sources once the even wave is through its pass -- it has arrived
                //  at the sync behind it --, adding into the plane itself: the same order)
                bool met = false;
                for (unsigned spins = 0; spins < kPairSpinLimit; ++spins) {
                    int v = 0;
                    if (lane == 0) v = __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__builtin_amdgcn_readfirstlane(v) - (pphase + 1) >= 0) { met = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!met && lane == 0 && p.status) __hip_atomic_store((gu32*)(p.status), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                wave_sync();
                if (lane == 0) pw[3] = 0;
            }
        }
        if (!listed)
        static_for<NT / 2>([&](auto SS) {
            constexpr int s = decltype(SS)::value;
            // partner of a[s]: class 0 -> a[(NT-s) mod NT];  else b[NT-1-s].  partner of b[s]: class RQ/2 -> b[NT-1-s]; else a[NT-1-s]
            const f2 pa0 = za[(NT - s) & (NT - 1)], pb = zb[NT - 1 - s], pa = za[NT - 1 - s];
            f2 PA, PB;
            if constexpr (pz == 0) {
                PA = f2{isg0 ? pa0.x : pb.x, isg0 ? pa0.y : pb.y};
                PB = f2{isg0 ? pb.x : pa.x, isg0 ? pb.y : pa.y};
            } else {                                         // no self-conjugate class in the later passes
                PA = pb; PB = pa;
            }
            const bool st = (s >= s0) && (s <= s1);
            process_stripe<s, RQ, NWIN>(za[s], PA, zb[s], PB, tiny, ownA + RQ * s, ownB + RQ * s, st, row_disp, flag, tq, j, klo, K, rAi, rBi, R2, mx);
        });
        // k' = nwin/2 (class 0, j = 8) is its own partner: V = 2 Re(Z[nwin/2]) is real, its shift is exactly 0
        if constexpr (pz == 0) {
            if (s1 == NT / 2 && isg0) own_base[j * OLD + NWIN / 2 - RQ * s0] = f2{2.0f * za[NT / 2].x, 0.0f};
        }
#endif
        };
        if constexpr (PAIR) {
            for (int attempt = 0; attempt < 2; ++attempt) {
                mx = 0.0f;
                if (role == 0) one_pass(std::integral_constant<int, 0>{});
                else { one_pass(std::integral_constant<int, 1>{}); pmx[lane] = mx; }
                pair_sync();                                 // both passes are in the planes, the odd wave's additions in its list
                const int cnt = __builtin_amdgcn_readfirstlane(pw[3]);
                if (cnt <= kPairListCap) {
                    if (role == 0 && cnt > 0) {              // replay: list order = the one-wave kernel's order of pass 1
                        float* dpl = reinterpret_cast<float*>(disp_base);
                        for (int base = 0; base < cnt; base += 64) {
                            const int i = base + lane;
                            if (i < cnt) {
                                float* q = dpl + plist.cell[i];
                                const f2 v = plist.val[i];
                                __hip_atomic_fetch_add(q, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(q + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                        wave_sync();
                    }
                    break;
                }
                // (rare) more additions than the list holds: the planes are cleaned and the group is done again with the odd wave's
                // sources behind the even wave's -- for the rest of this pair's launch (a tonal input that moves everything)
                if (role == 0) {
                    for (int i = lane; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
                    if (lane == 0) flag[0] = 0;
                    wave_sync();
                }
                pordered = true;
                pair_sync();
            }
            if (role == 0) mx = fmaxf(mx, pmx[lane]);
        } else {
            static_for<NPASS>(one_pass);
            wave_sync();
        }
        SPROBE(6);
        if (!PAIR || role == 0) {                            // (PAIR: the odd wave waits at the sync that ends the group)
        // one LDS round trip for both per-group flags (dirty displaced plane, queued rounding ties)
        int f_dirty = flag[0];
        const int f_ties = flag[1];
        bool exact = false;
#ifndef HSS_NO_EXACT
        // ---- "Exact groups": no stored cell reaches kExactTheta R of the tile -> the R of the group's own samples -> float64
        if (__builtin_amdgcn_ballot_w64(mx > kExactTheta2 * R2) == 0ull && R2 > 0.0f) {
            float e2 = 0.0f;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            for (int i = lane_e; i < 16 + NWIN - 1; i += 64) { const float v = xs[grp * 16 + i]; e2 = fmaf(v, v, e2); }
            const float R2g = p.r2scale * __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(piece_sums(e2, 0.0f, 0.0f, 0.0f))));
            exact = __builtin_amdgcn_ballot_w64(mx > kExactTheta2 * R2g) == 0ull && R2g > 0.0f;
        }
        if (te.dcdom) exact = true;                      // (an offset with little on top: tile_energy, fsst_kernels.hpp)
#endif
        if (exact) {
            for (int i = lane_o; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
            if (lane_o < 2) flag[lane_o] = 0;
            wave_sync();
            const float* xg = xs + grp * 16;
            resolve_bitmap<NWIN, true>(reinterpret_cast<unsigned*>(tq), [xg](int i) -> double { return static_cast<double>(xg[i]); }, disp_base, LDF,
                                       flag, klo, K, own_base, OLD, RQ * s0, RQ * (s1 + 1), p.wtab, p.twtab, 1.0, lane_o);
            wave_sync();
            f_dirty = flag[0];
        } else if (__builtin_amdgcn_readfirstlane(f_ties) != 0) {       // (rare) cells whose rounding float32 cannot decide
            resolve_ties<NWIN>(tq, xs + grp * 16, disp_base, LDF, flag, klo, K, own_base, OLD, RQ * s0, RQ * (s1 + 1), p.wtab, p.twtab, lane_o);
            wave_sync();
            f_dirty = flag[0];
        }

        // ---- epilogue for these 16 frames: element f -> (frame jj, kept row k)
        const bool wdirty = __builtin_amdgcn_readfirstlane(f_dirty) != 0;
        const int nvalid = min(16, cend - tg);
        const int koff = klo - RQ * s0;
        const int gidx = grp0 + sub + grp;                   // group index in the signal = statistics partial slot
        if constexpr (FAST) {
            // lane (g, j): frame j, kept rows k = g + 4 u (u < 6 covers K <= 24) as packed (re, im) cells
            f2* src = own_base + j * OLD + koff + g;
            if (wdirty) {                                    // (rare) fold the displaced plane into the own plane
                const f2* dsp = disp_base + j * LDF + g;
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (g + 4 * u < K) src[4 * u] += dsp[4 * u];
                wave_sync();
            }
            if (p.mode == kModeStack) {
                // statistics partial of this group (see "Statistics" in fsst_kernels.hpp): sums of (v - pivot) and
                // (v - pivot)^2 about a pivot (pivot_med3).  Six unconditional cell reads (a row past the
                // band still lies inside this wave's LDS), then only the one row group that is partial across lanes
                // pays for a select
                // (pivot: median of frame 0's first, middle and last kept row -- pivot_med3, fsst_kernels.hpp; broadcast reads)
                const f2 pv0 = own_base[koff], pv1 = own_base[koff + (K >> 1)], pv2 = own_base[koff + K - 1];
                const f2 piv = f2{pivot_med3(pv0.x, pv1.x, pv2.x), pivot_med3(pv0.y, pv1.y, pv2.y)};
                f2 v[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) v[u] = src[4 * u];
                const int ufull = (nvalid == 16) ? (K >> 2) : 0;     // rows 4u + 3 < K: valid in every lane
                const bool jv = j < nvalid;
                f2 st_s = {0.0f, 0.0f}, st_q = {0.0f, 0.0f};
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const f2 d = v[u] - piv;
                    if (u < ufull) {
                        st_s += d; st_q = pk_fma(d, d, st_q);
                    } else if (4 * u < K) {
                        const bool ok = jv && (g + 4 * u < K);
                        const f2 dm = {ok ? d.x : 0.0f, ok ? d.y : 0.0f};
                        st_s += dm; st_q = pk_fma(dm, dm, st_q);
                    }
                }
                const float w = piece_sums(st_s.x, st_q.x, st_s.y, st_q.y);
                if constexpr (FUSED) store_partial(part_lds + ((static_cast<int>(ksig) & 1) * kFusedMaxGroups + gidx) * kPartFloats, w, piv.x, piv.y);
                else store_partial(p.partials + (b * ngroups + gidx) * kPartFloats, w, piv.x, piv.y);
            }
            // the group's nvalid x 2K floats are contiguous in HBM: 16-byte stores, lane-linear; each float4 =
            // two adjacent (re,re) or (im,im) pairs of one frame row.  All LDS reads first, then the stores.
            const int C = 2 * K;
            const char* ob = reinterpret_cast<const char*>(own_base);
            float4* dst4 = reinterpret_cast<float4*>(p.out + (b * static_cast<long long>(ncols) + tr) * C) + lane_o;
            const int lim = nvalid * (K >> 1);
            f4 o[3];
#if defined(HSS_ABLATE) && HSS_ABLATE >= 1
            if (tg == 123456789)
#endif
            {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned pk = ppk_lds[i * 64 + lane_o];
                const int p0 = static_cast<int>(pk & 0xffffu), p1 = static_cast<int>(pk >> 16);
                o[i].x = *reinterpret_cast<const float*>(ob + p0);
                o[i].y = *reinterpret_cast<const float*>(ob + p0 + 8);
                o[i].z = *reinterpret_cast<const float*>(ob + p1);
                o[i].w = *reinterpret_cast<const float*>(ob + p1 + 8);
            }
            asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));   // keep the reads ahead of the predicated stores
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (lane + 64 * i < lim) {
                    // FUSED: ordinary stores (a wave of this CU reads the tile back for the z-score).  Otherwise
                    // streaming stores: the features are not read again by this kernel, and lines left dirty in L2 by
                    // 256 CUs that all write until the last microsecond cost ~10 us of write-back after the kernel
                    if constexpr (FUSED) *reinterpret_cast<f4*>(dst4 + 64 * i) = o[i];
                    else if constexpr (STREAM) {
                        if (p.state != nullptr) {
                            // read back by the channel's last block, which may sit on another XCD: agent-scope stores (sc1, written
                            // through) and agent-scope loads there -- no L2 write-back / invalidate fences (measured: they made
                            // the step 51 us instead of 33)
                            unsigned long long* q = reinterpret_cast<unsigned long long*>(dst4 + 64 * i);
                            __hip_atomic_store(q, static_cast<unsigned long long>(__float_as_uint(o[i].x)) | (static_cast<unsigned long long>(__float_as_uint(o[i].y)) << 32),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(q + 1, static_cast<unsigned long long>(__float_as_uint(o[i].z)) | (static_cast<unsigned long long>(__float_as_uint(o[i].w)) << 32),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            __builtin_nontemporal_store(o[i], reinterpret_cast<f4*>(dst4 + 64 * i));
                            if (p.mirror != nullptr)         // (no normalisation: the features are final, the host copy is written here)
                                __builtin_nontemporal_store(o[i], reinterpret_cast<f4*>(p.mirror + (b * static_cast<long long>(ncols) + tr) * C) + lane_o + 64 * i);
                        }
                    }
                    else __builtin_nontemporal_store(o[i], reinterpret_cast<f4*>(dst4 + 64 * i));
                }
            }
            if constexpr (STREAM) {
                // the group IS a piece of the running moments' summation order (fsst_kernels.hpp, chunk_moments): its float64
                // sums from the image registers -- the elements and the order piece_moments takes from memory
                if (p.state != nullptr) {
                    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (lane_o + 64 * i < lim) mom_acc4(a, make_float4(o[i].x, o[i].y, o[i].z, o[i].w), lane_o + 64 * i, C, K);
                    unsigned long long* pq = reinterpret_cast<unsigned long long*>(p.pieces + (b * ngroups + gidx) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double t = wave_sum(a[e]);
                        if (lane_o == 0) __hip_atomic_store(pq + e, static_cast<unsigned long long>(__double_as_longlong(t)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            }
        } else if (p.mode == kModeRaw) {
            float2* dst = reinterpret_cast<float2*>(p.out) + (b * K) * static_cast<long long>(ncols) + tr;
            int e0 = lane_o;                                 // opaque per GROUP (see the general epilogue below)
            asm volatile("" : "+v"(e0));
            for (int e = e0; e < K * 16; e += 64) {
                const int k = e >> 4, jj = e & 15;
                if (jj < nvalid) {
                    f2 v = own_base[jj * OLD + koff + k];
                    if (wdirty) v += disp_base[jj * LDF + k];
                    dst[static_cast<long long>(k) * ncols + jj] = make_float2(v.x, v.y);
                }
            }
        } else {
            // lane (g, j): frame j, kept rows k = g, g + 4, g + 8 ... as packed (re, im) pairs
            const bool isabs = (p.mode == kModeAbs);
            f2 pv0 = own_base[koff], pv1 = own_base[koff + (K >> 1)], pv2 = own_base[koff + K - 1];     // statistics pivot: pivot_med3 of frame 0
            if (wdirty) { pv0 += disp_base[0]; pv1 += disp_base[K >> 1]; pv2 += disp_base[K - 1]; }
            const f2 piv = f2{pivot_med3(pv0.x, pv1.x, pv2.x), pivot_med3(pv0.y, pv1.y, pv2.y)};
            f2 st_s = {0.0f, 0.0f}, st_q = {0.0f, 0.0f};
            if (j < nvalid) {
                const int C = isabs ? K : 2 * K;
                const int steps = (K - g + 3) >> 2;          // rows g + 4 i < K
                const f2* src = own_base + j * OLD + koff + g;
                const f2* dsp = disp_base + j * LDF + g;
                // (uniform 64-bit base + a 32-bit lane offset: no per-lane 64-bit address pair to keep or spill)
                float* gbase = p.out + (b * static_cast<long long>(ncols) + tr) * C;
                int g_g = g_o, j_g = j_o;                    // opaque per GROUP: otherwise the two addresses are computed per
                asm volatile("" : "+v"(g_g), "+v"(j_g));     // chunk, held through the transform and spilled
                const int loff = j_g * C + g_g;
                float* dst = gbase + loff;
                float* dsti = gbase + (loff + K);
                for (int i = 0; i < steps; ++i) {
                    f2 v = src[4 * i];
                    if (wdirty) v += dsp[4 * i];
                    if (isabs) {
                        dst[4 * i] = sqrtf(fmaf(v.x, v.x, v.y * v.y));
                    } else {
#if defined(HSS_ABLATE) && HSS_ABLATE >= 1
                        if (tg == 123456789)
#endif
                        {
                        dst[4 * i] = v.x;
                        dsti[4 * i] = v.y;
                        }
                        const f2 d = v - piv;
                        st_s += d;
                        st_q = pk_fma(d, d, st_q);
                    }
                }
            }
            if (p.mode == kModeStack) {
                const float w = piece_sums(st_s.x, st_q.x, st_s.y, st_q.y);
                store_partial(p.partials + (b * ngroups + gidx) * kPartFloats, w, piv.x, piv.y);     // (FUSED too: see CTL)
            }
        }
        wave_sync();
        if (wdirty) {
            for (int i = lane_o; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
            if (lane_o == 0) *flag = 0;
            wave_sync();
        }
        }
        pair_sync();                                         // the planes are free for the next group's passes
    }
    }
    if constexpr (FUSED) {
        // ---- A(ksig, c) delivered: count its groups; whoever completes the signal resolves its statistics
        const int sl = static_cast<int>(ksig) & 1;
        unsigned before = 0;
        if (lane == 0) before = __hip_atomic_fetch_add(done_a + sl, static_cast<unsigned>(ngrp), __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        before = __builtin_amdgcn_readfirstlane(before);
        // (monotone counter: slot sl has seen the signals sl, sl + 2, ..., ksig; no reset, no window for a race)
        if (before + static_cast<unsigned>(ngrp) == static_cast<unsigned>(ngroups) * (static_cast<unsigned>(ksig >> 1) + 1u)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int ncols_o = ncols, K_o = K, ng_o = ngroups;    // opaque: nothing of the float64 arithmetic below may be
            asm volatile("" : "+s"(ncols_o), "+s"(K_o), "+s"(ng_o));   // hoisted out of the work loop (it would be spilled)
            // (!FAST: the partials were written to HBM by waves of this workgroup, each wave's stores complete before its
            //  release on the delivery counter, and no line of them was read by this CU before)
            const float* parts = FAST ? part_lds + sl * kFusedMaxGroups * kPartFloats : p.partials + b * ngroups * kPartFloats;
            const float4 st = signal_stats(parts, ng_o, 16, ncols_o, K_o, lane_o);
            if (lane == 0) {
                fin_stats[static_cast<int>(ksig) & 3] = st;
                __hip_atomic_store(ready + (static_cast<int>(ksig) & 3), static_cast<unsigned>(ksig) + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_sync();
        }
    }
    }
    chunk = draw_pair();
    }
    SPROBE(1);
    if constexpr (STREAM) {
        if (p.state != nullptr) {
            // the channel's last block to get here merges the chunk into the running moments and normalises it
            // this wave's feature stores (agent scope, written through) are out; the counter below and the loads of the last block
            // are agent-scope accesses issued after that
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                          // (the operand table in LDS is dead from here on)
            unsigned* last_sh = reinterpret_cast<unsigned*>(smem);
            if (threadIdx.x == 0) {
                const unsigned before = __hip_atomic_fetch_add(p.arrive + st_ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool last = before + 1u == static_cast<unsigned>(p.bpc);
                if (last) __hip_atomic_store(p.arrive + st_ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (steps are stream-ordered)
                last_sh[0] = last ? 1u : 0u;
            }
            __syncthreads();
            SPROBE(2);
            if (last_sh[0] != 0u) {
                float4* st_sh = reinterpret_cast<float4*>(smem + 4);
                float* cbase = p.out + static_cast<long long>(st_ch) * ncols * (2 * K);
                constexpr int NPF = 6;                       // the chunk's first float4s per thread, in flight during the merge
                float4 cpre[NPF];
                stream_normalize_load<64 * WPB, NPF>(cbase, ncols, K, static_cast<int>(threadIdx.x), cpre, AgentLoad4());
                if (wv == 0) {
                    const unsigned long long* pq = reinterpret_cast<const unsigned long long*>(p.pieces + static_cast<long long>(st_ch) * ngroups * 4);
                    double m[4];
                    moments_from_pieces(ngroups, lane, [&](int q, int e) {
                        return __longlong_as_double(static_cast<long long>(__hip_atomic_load(pq + q * 4 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                    }, m);
                    if (lane < 2) {
                        const float2 r = merge_state(p.state + static_cast<long long>(st_ch) * 6 + lane * 3, m[2 * lane], m[2 * lane + 1],
                                                     static_cast<double>(K) * static_cast<double>(ncols));
                        float* sf = reinterpret_cast<float*>(st_sh);
                        sf[2 * lane] = r.x; sf[2 * lane + 1] = r.y;
                    }
                }
                __syncthreads();
                SPROBE(3);
                stream_normalize_apply<64 * WPB, NPF>(cbase, ncols, K, static_cast<int>(threadIdx.x), *st_sh, cpre, AgentLoad4(),
                                                      p.mirror ? p.mirror + static_cast<long long>(st_ch) * ncols * (2 * K) : nullptr);
                SPROBE(4);
            }
        }
    }
#ifdef HSS_STREAM_PROBE
    if constexpr (STREAM) {
        sprobe[7] = wall_clock64() - sprobe_t0;
        const int wid = static_cast<int>(blockIdx.x) * WPB + wv;
        if (lane == 0 && wid < kStreamProbeWaves)
            for (int k = 0; k < 8; ++k) g_stream_probe[wid * 8 + k] = sprobe[k];
    }
#endif
#ifdef HSS_CLOCKPROBE
    // development only (STACK_UNNORM, tools/clock_probe.py): HSS_CLOCKPROBE=1 -- shader-clock ticks and 100 MHz ticks one
    // wave in the middle of the grid lived; =2 -- start / end time (100 MHz ticks, low 32 bits) of every wave
    if (HSS_CLOCKPROBE == 1 && blockIdx.x == gridDim.x / 2 && wv == 0 && lane == 0) {
        p.out[0] = static_cast<float>(__builtin_readcyclecounter() - probe_c0);
        p.out[1] = static_cast<float>(wall_clock64() - probe_r0);
    }
    if (HSS_CLOCKPROBE == 2 && lane == 0) {
        unsigned* o = reinterpret_cast<unsigned*>(p.out) + 2 * (static_cast<size_t>(blockIdx.x) * WPB + wv);
        o[0] = static_cast<unsigned>(probe_r0); o[1] = static_cast<unsigned>(wall_clock64());
    }
#endif
}

}  // namespace hssfsst
