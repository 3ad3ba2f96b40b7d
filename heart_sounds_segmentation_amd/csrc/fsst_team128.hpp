// fsst_team128.hpp -- the canonical-window transform with the z-score of FSST._stack_real_imag
// (/root/reference/hss/transforms/synchrosqueeze.py:78-85) done IN REGISTERS: every feature is written to HBM exactly
// once, already normalised (algorithmic traffic: 8 000 B in + 352 000 B out per 2000-sample window, nothing else).
//
// Why this shape.  The z-score needs the mean / unbiased std of a whole signal's (n, 2K) feature block before its first
// element can be stored.  Round 2 kept the block in HBM (written, read back by the same CU, overwritten: 2.94x the
// algorithmic traffic, profiles/r02_pmc.json) because neither the L2 (profiles/r02_fused_team_variant.txt) nor the LDS of
// 16 resident waves can hold it.  Measured this round (profiles/r03_occupancy.txt): the transform loses only 6 % at TWO
// waves per SIMD instead of four -- and two waves per SIMD own 256 VGPRs each.  So:
//   * one persistent block of 8 waves per CU; a TEAM of T CUs of one XCD shares each signal: the signal's 64-frame chunks
//     (4 groups of 16 frames = one tile = one block of the statistics' summation order, kStatBlock) are dealt round-robin
//     to the team's CUs, rotated by the signal index so that the short last chunk does not always hit the same CU;
//   * a wave transforms a chunk exactly like fsst_core128_kernel, but the group's (16, 2K) image goes from the LDS plane
//     into 12 VGPRs per group instead of to HBM; the four statistics partials of the chunk are folded, in float64 and in
//     the order of signal_stats(), into ONE block sum {sum re, sum re^2, sum im, sum im^2} that the wave publishes in the
//     team's mailbox (8 tagged 8-byte words, relaxed agent-scope atomics: no fence, no cache-wide write-back);
//   * the wave then transforms its NEXT chunk (second register set) -- that is the time in which the team-mates publish
//     theirs -- and only afterwards collects the signal's block sums (one tagged 16-byte read per lane and 16 blocks),
//     runs stats_finish() on them (the very instructions of the two-kernel path: bit-identical statistics), z-scores
//     the first chunk's registers and streams them out.  A wave never waits while it still has work it may start.
// Only statistics cross CUs (64 B per chunk); nothing relies on a cache keeping anything.
//
// Progress.  Every CU's work list is signal-major and identical in shape, a wave publishes a chunk before it waits for
// anything, and it waits only for signals whose chunks were all handed out earlier than the chunk it would start next;
// with at most kTeamWaves chunks of one signal per CU (host-checked) the oldest unresolved signal of a team can always
// complete.  Mailbox slots (kTeamMailSlots per team, indexed by the signal's ordinal in the team's list) are recycled
// safely because a CU does not START a chunk more than kTeamWindow signals ahead of its own oldest unresolved one
// (2 kTeamWindow + 2 <= kTeamMailSlots; argument in DESIGN.md section 4.1b).  All blocks of the grid must be resident at
// once (grid <= number of CUs, one block per CU by its 256-VGPR waves); every wait is bounded in wall-clock time and
// reports through the plan's status word instead of hanging.
#pragma once
#include "fsst_mfma128.hpp"
#include "fsst_canon128.hpp"

namespace hssfsst {

constexpr int kTeamWaves = 8;                // waves per block: two per SIMD, up to 256 VGPRs each
constexpr int kTeamGpc = 4;                  // groups per chunk = one 64-frame tile = one statistics block
constexpr int kTeamMailSlots = 64;           // mailbox slots per team (signal ordinal mod this)
constexpr int kTeamWindow = 31;              // a CU starts no chunk this many signals ahead of its oldest unresolved signal
constexpr int kTeamMaxChunks = 64;           // chunks per signal the mailbox layout allows (signals up to 4096 frames)
constexpr int kTeamPark = 2;                 // canonical-band instantiation: groups of the held chunk that wait in LDS, not in registers
constexpr int kTeamPf = 2;                   // mailbox blocks per lane requested in one go (16 kTeamPf chunks = 2048 frames)
constexpr int kTeamCtlFloats = 16 + 64 + 192;    // [0] work counter, [1] a wait gave up, [8..15] oldest unresolved signal per
                                                 // wave, [16..79] column classes, [80..271] wide-store offsets
static_assert(kStatBlock == kTeamGpc, "one published block sum per chunk: the chunk is the statistics block");
static_assert(2 * kTeamWindow + 2 <= kTeamMailSlots, "mailbox recycling argument");

struct Team128Params {
    const float* x;       // [nsig][xstride]
    float* out;           // [nsig][ncols][2K]
    const float* atab;    // MFMA A-operand constants, then the wide-store offset table (as Core128Params)
    const double* wtab;   // rounding-tie path tables
    const double* twtab;
    unsigned long long* mail;   // [teams][kTeamMailSlots][nchunks][8] tagged words {tag << 32 | half of a double}
    unsigned* status;     // device status word (0 = ok)
    float r2scale;
    float inv_c;          // canonical-band instantiation: 1 / (scale of the f16 constants)
    int n, klo, K, nsig, col0, ncols;
    long long xstride;
    int team;             // CUs per team (power of two, <= 32, <= nchunks)
    int cpc;              // list positions per CU and signal: ceil(nchunks / team) rounded up to a power of two (<= kTeamWaves)
    int cpc_shift;        // log2(cpc)
    int nchunks;          // chunks per signal
    unsigned seq;         // launch sequence number of the plan (upper half of the mailbox tags)
    unsigned spin_ticks;  // bound of a wait in 100 MHz ticks
    unsigned* arrive;     // arrival counter of the plan (monotone over launches)
    unsigned arrive_base; // its value before this launch: block identity = arrival number - arrive_base
    int static_ids;       // development: block identity = blockIdx (teams inside one XCD, but no progress guarantee)
    unsigned* abort_word; // device word: a wait that ran out of time stores `launch` here; every wave then leaves the kernel
    unsigned* fallbacks;  // pinned host word: the same store, for the host's eyes (hssfsst_plan_fallbacks)
    unsigned launch;      // identity of this launch (never 0)
    unsigned long long* probe;   // development (HSS_TEAM_PROBE): per-phase shader-clock totals over all waves
};

using gu64 = __attribute__((address_space(1))) unsigned long long;

// (KLO, KC) >= 0: the band is that compile-time constant and a chunk's transform is canon_group / canon_stats / canon_image
// of fsst_canon128.hpp (p.atab = the f16 operand table, p.r2scale = its scaled r2scale, p.inv_c its constant scale) -- the
// same arithmetic as fsst_canon_kernel, hence bit-identical features on every z-score path.
template <int S1C, int KLO = -1, int KC = 0>
__global__ __launch_bounds__(64 * kTeamWaves, 2) void fsst_team128_kernel(Team128Params p)
{
    constexpr bool CANON = KLO >= 0;
    using CC = CanonCfg<(CANON ? KLO : 4), (CANON ? KC : 22)>;
    constexpr int NT = 16, RQ = 8, NWIN = 128, KST = 2, FPW = 16 * kTeamGpc, WPB = kTeamWaves;
    constexpr int ATAB = CANON ? kCanonAtabFloats : core128_atab_floats(RQ, NT);
    constexpr int XS = ((FPW + NWIN - 1 + 3) / 4) * 4;
    using avec = float __attribute__((ext_vector_type(KST)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = CANON ? KC : p.K, klo = CANON ? KLO : p.klo, n = p.n;
    const int LDF = CANON ? CC::LDF : plane_ldf(K);
    const int OLD = CANON ? CC::LD : own_ld(klo, K, RQ);
    const int s0 = (S1C >= 0) ? 0 : own_s0(klo, RQ), s1 = (S1C >= 0) ? S1C : own_s1(klo, K, RQ);

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* atab = smem;
    int* next_q = reinterpret_cast<int*>(smem + ATAB);
    unsigned* dead = reinterpret_cast<unsigned*>(smem + ATAB) + 1;
    int* pend = reinterpret_cast<int*>(smem + ATAB) + 8;                     // [kTeamWaves]
    unsigned* cls_lds = reinterpret_cast<unsigned*>(smem + ATAB + 16);       // [64]
    unsigned* ppk_lds = reinterpret_cast<unsigned*>(smem + ATAB + 80);       // [3][64]
    float* wbase = smem + ATAB + kTeamCtlFloats + wv * (CANON ? CC::wave_floats() : wave_lds_floats(FPW, klo, K, RQ, NT));
    float* xs = wbase;
    // CANON: the held chunk's last kTeamPark groups (three float4 per lane and group) wait in LDS instead of registers -- 3 kB
    // per wave and group that this 8-wave kernel has to spare; the registers are what the allocator was short of (it spilled
    // them to scratch)
    f4* park = reinterpret_cast<f4*>(smem + ATAB + kTeamCtlFloats + WPB * (CANON ? CC::wave_floats() : 0)) + wv * 3 * 64 * kTeamPark;
    f2* own_base = reinterpret_cast<f2*>(wbase + (CANON ? 2 * kCanonRecs : XS));
    f2* disp_base = own_base + 16 * OLD;
    int* flag = reinterpret_cast<int*>(disp_base + 16 * LDF);
    int* tq = flag + 4;

    for (int i = threadIdx.x; i < ATAB; i += 64 * WPB) {
        if constexpr (CANON) {
            atab[i] = p.atab[i];
        } else {
            const int ks = i % KST, l = (i / KST) & 63, pt = i / (KST * 64);
            atab[i] = p.atab[(pt * KST + ks) * 64 + l];
        }
    }
    const int ncols = p.ncols, cend = p.col0 + p.ncols;
    for (int i = lane; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
    if constexpr (CANON) {
        if (lane < 4) flag[lane] = 0;
        if (lane < kCanonTieWords) tq[lane] = 0;
    } else {
        if (lane < 4) flag[lane] = 0;
        for (int i = lane; i < tie_words(NWIN); i += 64) tq[i] = 0;
    }
    if (threadIdx.x < 8) next_q[threadIdx.x] = 0;
    if (threadIdx.x >= 8 && threadIdx.x < 16) next_q[threadIdx.x] = 0x7fffffff;      // pend[]: nothing unresolved
    // Block identity = ARRIVAL number, not blockIdx: the blocks that are running always hold the identities 0 .. R - 1, so
    // every team below R / T is complete whatever share of the chip this launch was given (another process's kernels may
    // hold the rest -- DataLoader workers, /root/reference/main.py:202-218), and a complete team depends on nobody else:
    // it finishes, frees its CUs, the next blocks arrive.  With blockIdx as identity two processes could each hold half of
    // every team and wait for the other half until the time limit.
    if (threadIdx.x == 2)
        next_q[2] = p.static_ids ? static_cast<int>(blockIdx.x)
                                 : static_cast<int>(__hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.arrive_base);
    if (wv == 0) {
        unsigned cls = 0u;                               // bit 2i / 2i+1: the first / second pair of float4 i is imaginary
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned c = (4u * static_cast<unsigned>(lane + 64 * i)) % static_cast<unsigned>(2 * K);
            cls |= (c >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i);
            cls |= (c + 2 >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i + 1);
        }
        cls_lds[lane] = cls;
        if constexpr (CANON) {
#pragma unroll
            for (int i = 0; i < 3; ++i) ppk_lds[i * 64 + lane] = canon_store_offsets<(CANON ? KLO : 4), (CANON ? KC : 22)>(lane + 64 * i);
        } else {
            const int* ptab = reinterpret_cast<const int*>(p.atab + ATAB);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                ppk_lds[i * 64 + lane] = static_cast<unsigned>(ptab[i * 64 + lane]) | (static_cast<unsigned>(ptab[(3 + i) * 64 + lane]) << 16);
        }
    }
    __syncthreads();

    // ---- team geometry (wave-uniform): T consecutive identities form a team.  (static_ids: blocks are dealt to the XCDs
    //      round-robin, block b on XCD b % 8, and the T blocks {b : b % 8 == x, (b / 8) / T == h} are T CUs of ONE XCD.)
    const int T = p.team, cpc = p.cpc, NC = p.nchunks;
    const int virt = __builtin_amdgcn_readfirstlane(next_q[2]);
    const int xcd = virt & 7, cu_slot = virt >> 3;
    const int member = p.static_ids ? (cu_slot & (T - 1)) : (virt & (T - 1));
    const int team = p.static_ids ? xcd + 8 * (cu_slot / T) : virt / T;
    const int nteams = static_cast<int>(gridDim.x) / T;
    const int nk = (p.nsig > team) ? (p.nsig - team + nteams - 1) / nteams : 0;      // signals of this team
    const int nwork = nk * cpc;
    const int ngroups = (ncols + 15) >> 4;
    const double total = static_cast<double>(K) * static_cast<double>(ncols);
    gu64* mail = (gu64*)(p.mail) + static_cast<size_t>(team) * kTeamMailSlots * NC * 8;
    const unsigned t_start = static_cast<unsigned>(wall_clock64());
    // Parameters used once per chunk are re-read from the kernel-argument segment where they are used (scalar loads) instead of
    // living in scalar registers across the whole work loop: the loop holds ~130 wave-uniform values, the register file 100
    // (the compiler spilled 104 of them to lanes of a vector register: ~170 v_readlane / v_writelane per chunk).
    using kparams = const __attribute__((address_space(4))) Team128Params;
    kparams* const kp_ = (kparams*)__builtin_amdgcn_kernarg_segment_ptr();
    auto P = [&]() -> kparams* { kparams* q = kp_; asm volatile("" : "+s"(q)); return q; };

    const float* myA = atab + lane * KST;
    f2 tiny = {1.0e-37f, 0.0f};
    asm volatile("" : "+s"(tiny));

    // Giving up.  Blocks of a team wait for each other, and nothing guarantees that they are resident together when other
    // processes use the GPU (workgroups are dealt to the XCDs in order: a launch can stall on a full XCD while the blocks it
    // did place hold CUs and wait -- tools/team_stress.py with three processes).  So a wait is short (P()->spin_ticks, 0.5 ms by
    // default: a healthy one is microseconds): the wave that runs out of time stores this launch's identity in the abort
    // word, every wave sees it at its next wait or chunk and leaves, the CUs are free again, and the host has ALREADY queued
    // the same exec on the two-launch path behind this kernel, gated on that word (hssfsst.hip launch_core128): the features
    // are then simply computed there.  No error, no result of this kernel is kept.
    auto aborted = [&]() -> bool {
        return __hip_atomic_load(P()->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P()->launch;
    };
    auto gave_up = [&](unsigned) {
        if (lane == 0) {
            __hip_atomic_store(P()->abort_word, P()->launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu32*)(P()->fallbacks), P()->launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto expired = [&](unsigned since) -> bool {         // `since` = wall_clock64() when the wait began
        return static_cast<unsigned>(wall_clock64()) - since > P()->spin_ticks ||
               __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u || aborted();
    };
    if (aborted()) return;                               // (a block that starts after the launch was given up)

#ifdef HSS_TEAM_PROBE
    unsigned long long pr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pr_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // per-group phases of the transform (HSS_CANON_PROBE)
    unsigned long long pr_last = __builtin_readcyclecounter();
    const unsigned long long pr_begin = pr_last;
#define PROBE(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); pr_t[k] += now_ - pr_last; pr_last = now_; } while (0)
#else
#define PROBE(k) do { } while (0)
#endif
    // Register sets.  One set = the lane's 12 float4s of a chunk (4 groups x 3), kept as three 16-float vectors indexed
    // [4 group + component]: the group index is wave-uniform, so writing a group's float4 is four M0-indexed moves,
    // not a ladder of 48 conditional copies.  `cur` is filled by the transform, `prev` waits for statistics.
    using f16v = float __attribute__((ext_vector_type(16)));
    bool have_prev = false;
    int ko_prev = 0, grp0_prev = 0, ngrp_prev = 0;
    long long b_prev = 0;
    f16v prev[3], cur[3];
    f4 prevq[3][kTeamGpc], curq[3][kTeamGpc];            // CANON: the same sets as float4s (the group index is a switch, no indexed moves)

    // ---- draw: the next chunk of this CU's list (-> it_valid, ko, c) and its samples on their way into registers.
    //      The wave first registers a lower bound of the oldest signal it will hold unresolved -- `lower_hint`, or the
    //      signal of the list position it is about to draw or pass: the window test of its siblings must see it before
    //      the draw is visible (LDS executes a wave's operations in order; the fence only stops the compiler).
    constexpr int SREG = (FPW + NWIN - 1 + 63) / 64;
    float sreg[SREG];
    bool d_valid = false;                                // the DRAWN chunk: samples requested, not yet in LDS
    int ko_d = 0, c_d = 0;
    bool it_valid = false;                               // the LANDED chunk: its tile is in xs, next to be transformed
    int ko = 0, c = 0;
    float R2 = 0.0f;                                     // error-bound scale of the tile in xs (see "Rounding ties")
    CanonTile tile{};                                    // CANON: the landed tile's scales
    auto draw = [&](int lower_hint) {
        int qi = 0;
        if (lane == 0) {
            const int snap = __hip_atomic_load(next_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int lower = min(lower_hint, min(snap, nwork) >> P()->cpc_shift);
            __hip_atomic_store(pend + wv, lower, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            qi = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        qi = __builtin_amdgcn_readfirstlane(qi);
        d_valid = false;
        while (qi < nwork) {
            ko_d = qi >> P()->cpc_shift;
            c_d = ((member + ko_d) & (T - 1)) + T * (qi & (cpc - 1));
            if (c_d < NC) { d_valid = true; break; }
            if (lane == 0) qi = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            qi = __builtin_amdgcn_readfirstlane(qi);
        }
        if (d_valid) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const float* xsig = P()->x + (static_cast<long long>(team) + static_cast<long long>(ko_d) * nteams) * P()->xstride;
            const int t0 = P()->col0 + c_d * (16 * kTeamGpc);
#pragma unroll
            for (int k = 0; k < SREG; ++k) {
                const int gi = t0 + lane_o + 64 * k - NWIN / 2;
                sreg[k] = (gi >= 0 && gi < n) ? xsig[gi] : 0.0f;
            }
        }
    };
    // ---- land: the drawn chunk's samples from registers into the (free) LDS tile, and the tile's error-bound scale
    auto land = [&]() {
        it_valid = d_valid; ko = ko_d; c = c_d;
        d_valid = false;
        if (!it_valid) return;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        if constexpr (CANON) {
            static_assert(!CANON || SREG == 3, "canon_land takes the aligned tile as three samples per lane");
            tile = canon_land(sreg, reinterpret_cast<u2*>(xs), P()->r2scale, P()->inv_c, lane_o);
            return;
        }
        float e2 = 0.0f;
#pragma unroll
        for (int k = 0; k < SREG; ++k) {
            if (lane_o + 64 * k < FPW + NWIN - 1) xs[lane_o + 64 * k] = sreg[k];
            e2 = fmaf(sreg[k], sreg[k], e2);             // (a sample beyond the tile belongs to the next one: harmless in a bound)
        }
        R2 = P()->r2scale * __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(piece_sums(e2, 0.0f, 0.0f, 0.0f))));
        wave_sync();
    };
    draw(0x7fffffff);
    land();
    if (it_valid) draw(ko);

    // The wave works on three chunks at a time: HELD (transformed and published, features in `prev`, waiting for its
    // signal's statistics), LANDED (tile in LDS, transformed next) and DRAWN (samples on their way into registers).
    // One loop body, in this order -- a wave's memory operations retire in order and the compiler cannot count the ones
    // issued under an exec mask, so the ONE wait for loaded data per iteration sits where everything else in flight is
    // at least a group old:
    //   1. transform the landed chunk; at the start of its LAST group ask the mailbox for the held chunk's signal
    //      (published a whole chunk ago) and look at the answer right after the group: the one wait;
    //   2. land the drawn chunk (its samples were requested a chunk ago), publish the transformed chunk's block sum,
    //      draw the next chunk and request its samples;
    //   3. statistics of the held chunk's signal (mailbox passes only if step 1's answer was incomplete);
    //   4. z-score the held chunk's registers and stream them out: 12 stores that drain while step 1 runs again.
    // A landed chunk kTeamWindow signals ahead of the held one is transformed one iteration later (steps 3, 4 first).
    // With at most two chunks of a signal per CU (host-checked) a wave never holds three chunks of one signal, which is
    // what rules out waiting for a block sum that only the waiting wave itself could publish.
    for (;;) {
        if (__hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u || aborted()) break;     // the launch was given up
        const bool xf = it_valid && !(have_prev && ko - ko_prev >= kTeamWindow);
        long long b = 0;
        int grp0 = 0, ngrp = 0, ko_cur = 0;
        bool done = false;                               // the held chunk's signal: all block sums seen, summed in `acc`
        double acc = 0.0;
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned tag_prev = (P()->seq << 16) | (static_cast<unsigned>(ko_prev) & 0xffffu);
        const gu64* slot_prev = mail + static_cast<size_t>(ko_prev & (kTeamMailSlots - 1)) * NC * 8;
        if (xf) {
            PROBE(7);
            // ---- window: do not start a chunk kTeamWindow signals ahead of the oldest unresolved signal of a sibling
            const unsigned tw_window = static_cast<unsigned>(wall_clock64());
            for (;;) {
                int fl = 0x7fffffff;
                if (lane < kTeamWaves && lane != wv) fl = __hip_atomic_load(pend + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int off = 1; off < kTeamWaves; off <<= 1) fl = min(fl, __shfl_xor(fl, off));
                if (ko - __builtin_amdgcn_readfirstlane(fl) < kTeamWindow) break;
                if (expired(tw_window)) { gave_up(4u); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            PROBE(0);
            // ---- 1. transform chunk c of signal b: kTeamGpc groups of 16 frames out of the staged tile
            b = static_cast<long long>(team) + static_cast<long long>(ko) * nteams;
            grp0 = c * kTeamGpc;
            ngrp = min(kTeamGpc, ngroups - grp0);
            ko_cur = ko;
            const int c_cur = c;
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int g = lane_o >> 4, j = lane_o & 15;
            f2* row_disp = disp_base + j * LDF;
            const int t0 = P()->col0 + grp0 * 16;
            double bsum = 0.0;                                       // lane of row q: quantity q of this chunk's block sum
            unsigned long long pf[2 * kTeamPf];
#pragma unroll
            for (int r = 0; r < 2 * kTeamPf; ++r) pf[r] = 0ull;
            const bool pf_on = have_prev && NC <= 16 * kTeamPf;
            if constexpr (CANON) {
                // the set about to be filled carries nothing over from the previous chunk: say so (an empty statement that
                // "defines" the registers), otherwise 48 registers stay live around the loop for a chunk of fewer than 4 groups
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int g = 0; g < kTeamGpc; ++g) asm volatile("" : "=v"(curq[i][g]));
            }
            for (int grp = 0; grp < ngrp; ++grp) {
                if (pf_on && grp == ngrp - 1) {
#pragma unroll
                    for (int r = 0; r < kTeamPf; ++r) {
                        const int blk = min((lane_o >> 2) + 16 * r, NC - 1);
                        pf[2 * r] = __hip_atomic_load(slot_prev + blk * 8 + (lane_o & 3) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pf[2 * r + 1] = __hip_atomic_load(slot_prev + blk * 8 + (lane_o & 3) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                const int tg = t0 + grp * 16;
                if constexpr (CANON) {
                    canon_group<KLO, KC, 2>(reinterpret_cast<const u2*>(xs) + grp * 16, atab, own_base, disp_base, flag, tq, P()->wtab, P()->twtab,
                                         tile, tiny, lane_o, P()->x + b * P()->xstride, n, tg
#ifdef HSS_TEAM_PROBE
                                         , pr_c
#endif
                                         );
#ifdef HSS_TEAM_PROBE
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned long long pc_last = __builtin_readcyclecounter();
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    const int nvalid = min(16, cend - tg);
                    f2 piv;
                    const float w = canon_stats<KLO, KC>(own_base, nvalid, tile.inv, lane_o, piv);
                    const float wo = __shfl_xor(w, 16);
                    const int q = lane_o >> 4;
                    const float s1f = (q & 1) ? wo : w, s2f = (q & 1) ? w : wo;
                    const float pf32 = (q & 2) ? piv.y : piv.x;
                    const double cnt = static_cast<double>(nvalid) * static_cast<double>(K);
                    bsum += piece_moment(q, static_cast<double>(s1f), static_cast<double>(s2f), static_cast<double>(pf32), cnt);
#ifdef HSS_TEAM_PROBE
                    { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); pr_c[4] += now_ - pc_last; pc_last = now_; __builtin_amdgcn_sched_barrier(0); }
#endif
                    // the group index is wave-uniform: a four-way branch around the 12 LDS reads of the image, each arm
                    // writing its own float4 registers (no M0-indexed moves, no 16-register tuples to keep aligned)
                    static_for<kTeamGpc>([&](auto G) {
                        constexpr int gq = decltype(G)::value;
                        if (grp == gq) canon_image_to<KLO, KC>(own_base, ppk_lds, tile.inv, lane_o, [&](int i, f4 v) { curq[i][gq] = v; });
                    });
                    wave_sync();
#ifdef HSS_TEAM_PROBE
                    { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); pr_c[5] += now_ - pc_last; __builtin_amdgcn_sched_barrier(0); }
#endif
                    continue;
                }
                unsigned xaddr = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)(xs + grp * 16 + j + NT * g)));
                asm volatile("" : "+v"(xaddr));
                const lds_float* xb = (const lds_float*)static_cast<size_t>(xaddr);
                int pair = g;
                asm volatile("" : "+v"(pair));
                const bool isg0 = (pair == 0);
                const int rAi = pair, rBi = isg0 ? RQ / 2 : RQ - pair;
                f2* ownA = own_base + j * OLD + rAi - RQ * s0;
                f2* ownB = own_base + j * OLD + rBi - RQ * s0;

                f2 za[NT], zb[NT];
                float mx_unused = 0.0f;                      // (the general-band instantiation is not dispatched: it has no exact mode)
                static_for<NT / 4>([&](auto GG) {
                    constexpr int g0 = decltype(GG)::value * 4;
                    f4 acc4[4];
                    avec a2[4];
                    static_for<4>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        a2[i] = *reinterpret_cast<const avec*>(myA + (g0 + i) * 64 * KST);
                        acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i][0], xb[g0 + i], f4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<4>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i][1], xb[g0 + i + 4 * NT], acc4[i], 0, 0, 0);
                    });
                    static_for<4>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        za[bitrev_n<NT>(g0 + i)] = f2{acc4[i].x, acc4[i].y};
                        zb[bitrev_n<NT>(g0 + i)] = f2{acc4[i].z, acc4[i].w};
                    });
                });
                fft_n<NT>(za);
                fft_n<NT>(zb);
                static_for<NT / 2>([&](auto SS) {
                    constexpr int s = decltype(SS)::value;
                    const f2 pa0 = za[(NT - s) & (NT - 1)], pb = zb[NT - 1 - s], pa = za[NT - 1 - s];
                    const f2 PA = f2{isg0 ? pa0.x : pb.x, isg0 ? pa0.y : pb.y};
                    const f2 PB = f2{isg0 ? pb.x : pa.x, isg0 ? pb.y : pa.y};
                    const bool st = (s >= s0) && (s <= s1);
                    process_stripe<s, RQ, NWIN>(za[s], PA, zb[s], PB, tiny, ownA + RQ * s, ownB + RQ * s, st, row_disp, flag, tq, j, klo, K, rAi, rBi, R2, mx_unused);
                });
                if (s1 == NT / 2 && isg0) own_base[j * OLD + NWIN / 2 - RQ * s0] = f2{2.0f * za[NT / 2].x, 0.0f};
                wave_sync();
                int f_dirty = flag[0];
                const int f_ties = flag[1];
                if (__builtin_amdgcn_readfirstlane(f_ties) != 0) {
                    resolve_ties<NWIN>(tq, xs + grp * 16, disp_base, LDF, flag, klo, K, own_base, OLD, RQ * s0, RQ * (s1 + 1), P()->wtab, P()->twtab, lane_o);
                    wave_sync();
                    f_dirty = flag[0];
                }
                const bool wdirty = __builtin_amdgcn_readfirstlane(f_dirty) != 0;
                const int nvalid = min(16, cend - tg);
                const int koff = klo - RQ * s0;
                f2* src = own_base + j * OLD + koff + g;
                if (wdirty) {
                    const f2* dsp = disp_base + j * LDF + g;
#pragma unroll
                    for (int u = 0; u < 6; ++u)
                        if (g + 4 * u < K) src[4 * u] += dsp[4 * u];
                    wave_sync();
                }
                // statistics partial of the group: the arithmetic of fsst_core128_kernel's FAST epilogue, then its
                // float64 moments about zero added to the chunk's block sum exactly as signal_stats() adds a block's pieces
                {
                    const f2 piv = own_base[koff];
                    f2 v[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) v[u] = src[4 * u];
                    const int ufull = (nvalid == 16) ? (K >> 2) : 0;
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
                    const float w = piece_sums(st_s.x, st_q.x, st_s.y, st_q.y);      // row 0 / 1 / 2 / 3: S1re / S2re / S1im / S2im
                    const float wo = __shfl_xor(w, 16);                              // the other sum of this row's block of columns
                    const int q = lane_o >> 4;
                    const float s1f = (q & 1) ? wo : w, s2f = (q & 1) ? w : wo;
                    const float pf32 = (q & 2) ? piv.y : piv.x;
                    const double cnt = static_cast<double>(nvalid) * static_cast<double>(K);
                    bsum += piece_moment(q, static_cast<double>(s1f), static_cast<double>(s2f), static_cast<double>(pf32), cnt);
                }
                // the group's image [nvalid][2K] as lane-linear float4s, from the own plane into the register set
                {
                    const char* ob = reinterpret_cast<const char*>(own_base);
                    const int e0 = 4 * grp;                          // wave-uniform element index: M0-indexed moves
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const unsigned pk = ppk_lds[i * 64 + lane_o];
                        const int p0 = static_cast<int>(pk & 0xffffu), p1 = static_cast<int>(pk >> 16);
                        cur[i][e0] = *reinterpret_cast<const float*>(ob + p0);
                        cur[i][e0 + 1] = *reinterpret_cast<const float*>(ob + p0 + 8);
                        cur[i][e0 + 2] = *reinterpret_cast<const float*>(ob + p1);
                        cur[i][e0 + 3] = *reinterpret_cast<const float*>(ob + p1 + 8);
                    }
                }
                wave_sync();
                if (wdirty) {
                    for (int i = lane_o; i < 16 * LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
                    if (lane_o == 0) *flag = 0;
                    wave_sync();
                }
            }
            PROBE(1);
            // ---- the answer of the early mailbox request (nothing else of this wave is in flight behind it).
            //      Lane (blk % 16, q) adds up quantity q of blocks blk, blk + 16, ... in that order (stats_from_blocks).
            if (pf_on) {
                bool ok = true;
#pragma unroll
                for (int r = 0; r < kTeamPf; ++r) {
                    if ((lane_o >> 2) + 16 * r < NC) {
                        ok = ok && static_cast<unsigned>(pf[2 * r] >> 32) == tag_prev && static_cast<unsigned>(pf[2 * r + 1] >> 32) == tag_prev;
                        acc += __longlong_as_double(static_cast<long long>((pf[2 * r] << 32) | (pf[2 * r + 1] & 0xffffffffull)));
                    }
                }
#if defined(HSS_TEAM_ABLATE) && HSS_TEAM_ABLATE >= 1    // development: cost of waiting for the team (results invalid)
                ok = true;
#endif
                done = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
            }
            PROBE(3);
            // ---- 2. the drawn chunk's samples (requested a chunk ago) into the tile the transform is done with; then
            //      publish the chunk's block sum: lanes 16 q and 16 q + 1 hold quantity q; word {tag, upper | lower half}
            land();
            {
                const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko_cur) & 0xffffu);
                gu64* slot = mail + (static_cast<size_t>(ko_cur & (kTeamMailSlots - 1)) * NC + c_cur) * 8;
                const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(bsum));
                if ((lane_o & 15) < 2) {
                    const int q = lane_o >> 4, half = lane_o & 1;
                    const unsigned payload = half ? static_cast<unsigned>(bits) : static_cast<unsigned>(bits >> 32);
                    __hip_atomic_store(slot + q * 2 + half, (static_cast<unsigned long long>(tag) << 32) | payload,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            //      next chunk: draw it and request its samples (oldest unresolved signal of this wave from now on: the
            //      held chunk's, else the one just transformed)
            if (it_valid) draw(have_prev ? ko_prev : ko_cur);
            PROBE(2);
        }
        // ---- 3. statistics of the held chunk's signal
        f2 ms[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { ms[i][0] = f2{0.0f, 0.0f}; ms[i][1] = f2{0.0f, 0.0f}; }
        if (have_prev) {
            PROBE(7);
            // mailbox passes with ALL loads of a pass in flight at once: under load a mailbox read is a 2-3 us trip through
            // this CU's own memory queue (MI355X_MICROARCH.md, handoff-1to1)
            const unsigned tw_mail = static_cast<unsigned>(wall_clock64());
            while (!done) {
                bool ok = true;
                acc = 0.0;
                for (int blk0 = lane_r >> 2; blk0 < NC; blk0 += 16 * kTeamPf) {
                    unsigned long long w[2 * kTeamPf];
#pragma unroll
                    for (int r = 0; r < kTeamPf; ++r) {
                        const int blk = min(blk0 + 16 * r, NC - 1);
                        w[2 * r] = __hip_atomic_load(slot_prev + blk * 8 + (lane_r & 3) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[2 * r + 1] = __hip_atomic_load(slot_prev + blk * 8 + (lane_r & 3) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int r = 0; r < kTeamPf; ++r) {
                        if (blk0 + 16 * r < NC) {
                            ok = ok && static_cast<unsigned>(w[2 * r] >> 32) == tag_prev && static_cast<unsigned>(w[2 * r + 1] >> 32) == tag_prev;
                            acc += __longlong_as_double(static_cast<long long>((w[2 * r] << 32) | (w[2 * r + 1] & 0xffffffffull)));
                        }
                    }
                }
#if defined(HSS_TEAM_ABLATE) && HSS_TEAM_ABLATE >= 1
                break;
#endif
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (expired(tw_mail)) { gave_up(3u); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            PROBE(3);
#if defined(HSS_TEAM_ABLATE) && HSS_TEAM_ABLATE >= 2    // development: cost of the float64 statistics per chunk
            const float4 st = make_float4(static_cast<float>(acc), 1.0f, 0.0f, 2.0f);
#else
            const float4 st = stats_finish(acc, total, lane_r);
#endif
            // {mean, 1/std} of each of the lane's six pairs (3 float4 x 2) as ONE register pair: the packed subtract /
            // multiply of step 5 broadcast its low / high half (VOP3P op_sel)
            const unsigned cls = cls_lds[lane_r];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bool im0 = (cls >> (2 * i)) & 1u, im1 = (cls >> (2 * i + 1)) & 1u;
                ms[i][0] = f2{im0 ? st.z : st.x, im0 ? st.w : st.y};
                ms[i][1] = f2{im1 ? st.z : st.x, im1 ? st.w : st.y};
            }
            PROBE(4);
        }
        // ---- 4. z-score of the held chunk: (v - mean) * (1 / std), two roundings, exactly as fsst_normalize_kernel
        if (have_prev) {
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(ms[i][0]), "+v"(ms[i][1]));
            auto zs = [](f2 v, f2 m) -> f2 {
                f2 d, e;
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(e) : "v"(d), "v"(m));
                return e;
            };
            const int C = 2 * K;
            float4* d4 = reinterpret_cast<float4*>(P()->out + (b_prev * static_cast<long long>(ncols) + grp0_prev * 16) * C) + lane_r;
            const int per = 8 * K;                       // float4 per full group
#if !defined(HSS_TEAM_ABLATE) || HSS_TEAM_ABLATE < 3
            auto emit = [&](int g, int i) {
                f4 pv;
                if constexpr (CANON) pv = (g >= kTeamGpc - kTeamPark) ? park[((g - (kTeamGpc - kTeamPark)) * 3 + i) * 64 + lane_r] : prevq[i][g];
                else pv = f4{prev[i][4 * g], prev[i][4 * g + 1], prev[i][4 * g + 2], prev[i][4 * g + 3]};
                const f2 lo = zs(f2{pv.x, pv.y}, ms[i][0]);
                const f2 hi = zs(f2{pv.z, pv.w}, ms[i][1]);
#if defined(HSS_TEAM_STORE) && HSS_TEAM_STORE == 1        // development: plain stores
                *reinterpret_cast<f4*>(d4 + g * per + 64 * i) = f4{lo.x, lo.y, hi.x, hi.y};
#elif defined(HSS_TEAM_STORE) && HSS_TEAM_STORE == 2      // development: the arithmetic without the stores
                { f2 l2 = lo, h2 = hi; asm volatile("" :: "v"(l2), "v"(h2)); }
#elif defined(HSS_TEAM_STORE) && HSS_TEAM_STORE == 3      // development: every wave stores into one small region (L2-resident)
                __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(reinterpret_cast<float4*>(P()->out) + (blockIdx.x * 8 + wv) * 1024 + lane_r + 64 * (3 * g + i)));
#else
                __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(d4 + g * per + 64 * i));
#endif
            };
            // a chunk of four full groups -- every chunk but a signal's last -- needs one predicate per group, not three: a full
            // group is 8 K float4s, every lane has the first (8 K / 64) of its three
            if (__builtin_expect(CANON && ngrp_prev == kTeamGpc && (grp0_prev + kTeamGpc) * 16 <= ncols, 1)) {
                static_for<kTeamGpc>([&](auto G) {
                    static_for<3>([&](auto I) {
                        constexpr int g = decltype(G)::value, i = decltype(I)::value;
                        constexpr int full = CANON ? 8 * KC : 0;
                        if constexpr (64 * (i + 1) <= full) emit(g, i);
                        else if constexpr (64 * i < full) { if (lane_r + 64 * i < full) emit(g, i); }
                    });
                });
            } else {
                asm volatile("");
#pragma unroll
                for (int g = 0; g < kTeamGpc; ++g) {
                    const int lim = (g < ngrp_prev) ? min(16, ncols - (grp0_prev + g) * 16) * (K >> 1) : 0;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (lane_r + 64 * i < lim) emit(g, i);
                }
            }
#else
            if (lane_r == 0 && (CANON ? prevq[0][0].x : prev[0][0]) == 123.456f) P()->out[0] = ms[0][0].x + ms[1][1].y;
#endif
            have_prev = false;
            PROBE(5);
        }
        if (xf) {
            if constexpr (CANON) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int g = 0; g < kTeamGpc - kTeamPark; ++g) prevq[i][g] = curq[i][g];
#pragma unroll
                    for (int g = kTeamGpc - kTeamPark; g < kTeamGpc; ++g) park[((g - (kTeamGpc - kTeamPark)) * 3 + i) * 64 + lane_r] = curq[i][g];
                }
                wave_sync();
            } else { prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2]; }
            have_prev = true; ko_prev = ko_cur; grp0_prev = grp0; ngrp_prev = ngrp; b_prev = b;
        } else if (!it_valid) {
            break;
        }
    }
#ifdef HSS_TEAM_PROBE
    // [0] window wait, [1] transform, [2] publish + draw, [3] poll, [4] stats_finish, [5] z-score + stores, [7] rest, [6] lifetime
    if (lane == 0) {
        pr_t[6] = __builtin_readcyclecounter() - pr_begin;
        for (int k = 0; k < 8; ++k) atomicAdd(p.probe + k, pr_t[k]);
        for (int k = 0; k < 6; ++k) atomicAdd(p.probe + 16 + k, pr_c[k]);
        atomicAdd(p.probe + 8, 1ull);
    }
#endif
    if (lane == 0) __hip_atomic_store(pend + wv, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace hssfsst
