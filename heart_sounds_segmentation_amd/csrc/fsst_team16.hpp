// fsst_team16.hpp -- the canonical-band transform with the z-score of FSST._stack_real_imag
// (/root/reference/hss/transforms/synchrosqueeze.py:78-85) applied IN REGISTERS at FOUR waves per SIMD: every feature is
// written to HBM exactly once, already normalised (algorithmic traffic: 8 000 B in + 352 000 B out per 2000-sample window).
//
// The z-score needs the mean / unbiased std of a whole signal's (n, 2K) feature block before its first element can be stored.
// Round 3 had two single-launch answers: one CU per signal (fsst_canon_kernel<.., true>: 16 waves per CU, but the un-normalised
// tile makes a round trip through HBM -- 2.98x the algorithmic traffic at 85 % of the chip's streaming rate, its own ceiling)
// and a team kernel whose waves held a 64-frame chunk in 48 registers across the next chunk's transform (1.04x the traffic,
// but 233 VGPRs = two waves per SIMD: the transform at 0.188 instead of 0.140 ms per 1024 windows; removed this round).  This
// kernel keeps the team idea and drops what made it wide:
//   * the unit of work is ONE 16-frame group.  A team of T CUs (16 T waves) shares a signal; its groups are dealt round-robin
//     to the team's CUs, whose 16 waves draw them from a ticket counter in LDS.  With 16 T >= twice the groups of a signal,
//     a signal passes through the team in about one group time;
//   * a wave holds up to DEPTH = 2 group images -- three float4 per lane and group, 24 registers -- across further groups'
//     transforms (canon_group of fsst_canon128.hpp, the arithmetic of every other canonical-band kernel: bit-identical
//     features on every z-score path): 24 FIXED registers (v104 .. v127, see below) + 104 for the transform = the 128 of four
//     waves per SIMD (no scratch in the hot loop -- the spills sit in the float64 and offset-edge paths; listing checked);
//     a strict first-in first-out: a held group leaves when its slot is needed; the LDS regions are those of fsst_canon_kernel;
//   * a CU takes CONSECUTIVE groups of a signal (two blocks of kStatBlock = 4), keeps their statistics partials (the six float32
//     numbers of the two-launch path) in LDS, and the wave that delivers a block's last partial forms the block's float64 sums
//     and publishes them as eight tagged 8-byte words in the team's mailbox (relaxed agent-scope atomics: no fence, no cache
//     write-back);
//   * the rest of the statistics runs ONCE per signal and CU: the first wave of a CU that cannot go on without a signal's
//     statistics claims it (LDS), fetches the signal's <= 32 block sums from the mailbox -- two per lane, straight into the lane
//     that adds them -- and finishes as stats_from_blocks() does (fsst_kernels.hpp: the same instructions on the same numbers as
//     the two-launch path), leaving {mean, 1/std} x 2 in LDS for its 15 siblings.
// What limits it (profiles/r04_team_occupancy.txt, profiles/r05_team_diet.txt): instruction issue -- 527 vector + 16 matrix
// instructions per group on a SIMD that four waves share, a third of a wave's cycles spent waiting for an issue slot -- and, for
// ~7 us of the kernel, the statistics: a resolve is two trips through the memory system (the last partial becoming visible,
// the copy; ~2.3 us each under the kernel's own write stream, inside one XCD as across: tools/mail_latency3.hip) plus the sums
// against a group time of ~5 us.  Measured and rejected: a third held group (12 more registers: one image spills to scratch in
// the hot loop; parked in global memory: the waits vanish, the traffic costs 6 %), three waves per SIMD with four held groups
// (the transform loses 25 %), parking images in the free LDS, resolving early (blocking or through global_load_lds copies
// looked at a group later), releasing in the middle of the transform, teams inside one XCD, one accumulating plane (LDS float
// atomics: 0.6 lanes per clock and CU).  Offsets: a tile that rides on an offset is transformed HERE (canon_land / canon_group,
// "Offsets"), no launch is given up for it.
//
// Progress.  A wave publishes a group before it waits for anything, and it waits only for the signal of the oldest group it
// HOLDS.  Besides the held groups it has up to two tickets it has not published yet (one landed, one drawn with its samples
// in flight): these must never belong to a signal it may wait for -- a first version drew ahead unconditionally and deadlocked
// as soon as a wave fell behind its siblings (held, landed and drawn group all of ONE signal: it waited for a statistic that
// needed its own drawn group).  Hence the rule in draw(): a ticket is drawn ahead only if every position still to be handed
// out lies in a LATER signal than the group just published; otherwise the wave draws when it has nothing landed.  Then: let a*
// be the oldest signal of a team with an unpublished group.  A drawn-but-unpublished group of a* belongs to a wave that is
// transforming or waits for an older -- complete -- signal: it gets published.  An undrawn one needs a free wave of its CU: if
// all 16 were waiting they would each hold a published group of a* that precedes it in the CU's list, 17 positions of one
// signal, but a list holds at most 16 (host-checked: cpc <= WPB).  A CU cannot run more than (DEPTH + 3) x 16 list positions
// ahead of its oldest unresolved signal (every wave is then waiting), which bounds the mailbox / LDS slots in use (host:
// slots >= 2 lead + 2).
//
// Giving up.  Blocks of a team wait for each other, and nothing guarantees that they are resident together once other
// processes use the GPU (DataLoader workers, /root/reference/main.py:202-218; several ranks on one device).  Two measures:
// block identity is the ARRIVAL number (one agent-scope atomic per block), not blockIdx: the running blocks hold identities
// 0 .. R - 1, so every team below R / T is complete whatever share of the chip the launch got; and every wait is bounded in
// wall-clock time (0.5 ms; a healthy one is microseconds): the wave that runs out of time stores the launch's identity in an
// abort word, every wave sees it at its next wait and leaves, and the host has ALREADY queued the same exec behind this
// kernel, every kernel of it gated on exactly that word (hssfsst.hip launch_core128): the features are then computed there --
// same bits, no error (hssfsst_plan_fallbacks counts).  A plan is single-stream: two launches of one plan on different
// streams interleave their arrivals; a block that finds its identity outside the grid gives the launch up.
#pragma once
#include "fsst_mfma128.hpp"
#include "fsst_canon128.hpp"

#ifndef HSS_T16_TAPB
#define HSS_T16_TAPB 2
#endif
namespace hssfsst {

using gu64 = __attribute__((address_space(1))) unsigned long long;

#ifdef HSS_T16_BLKPROBE      // development (tools/blk_probe.py): per wave of the LAST launch: waits, -, ticks waited, ticks resolving, resolves
__device__ unsigned g_t16_blk[256 * 16 * 8];
#endif
constexpr int kT16PartFloats = 6;            // a group's statistics partial in the CU's LDS: S1re S2re S1im S2im p_re p_im
constexpr int kT16BlockWords = 8;            // tagged 8-byte words per BLOCK of kStatBlock groups in the mailbox: the block's four
                                             // float64 sums (sum re, sum re^2, sum im, sum im^2), each as {high, low} half
constexpr int kT16MaxBlocks = kFusedMaxGroups / kStatBlock;      // 32 blocks per signal
constexpr int kT16SlotWords = kT16MaxBlocks * kT16BlockWords;          // a signal's mailbox slot
constexpr int kT16MaxCpc = 8;                // groups of a signal per CU (two blocks)
constexpr int kT16MaxSlots = 64;             // statistics slots per CU / mailbox slots per team (signal ordinal mod slots); 32 where the LDS is short
constexpr int kT16StatFloats = 12;           // a signal's statistics in LDS: three float4 {mean, 1/std} pairs -- (re, re), (re, im), (im, im): the
                                             // z-score of a float4 of the image reads the one its two column pairs need (emit_held)
// [0] ticket counter [1] dead [2] identity | statistics-table offsets | wide-store offsets | ready[slots], claim[slots] | statistics[slots][3] float4
constexpr int t16_ctl_base(int slots) { return 16 + 64 + 192 + 2 * slots + kT16StatFloats * slots; }
// ... then the partials of the CU's own groups, PSLOTS signals deep: [PSLOTS][kT16MaxCpc][6] floats + [PSLOTS][2] block counters
constexpr int t16_ctl_floats(int pslots, int slots) { return t16_ctl_base(slots) + pslots * (kT16MaxCpc * kT16PartFloats + 2); }
// what the LDS beside the tables and 16 wave regions leaves: partials 32 signals deep and 64 statistics slots where they fit
// own planes per wave: two where the LDS has room for them (the second one holds the previous step's image: "Two planes" below)
#ifndef HSS_T16_PLANES
#define HSS_T16_PLANES 1                     // (2: a second own plane per wave -- a third held group at no instruction; measured +0.9 % on the queued kernel,
#endif                                       //  profiles/r06_team_waits.txt: the resolves are started by the first wave that needs them, whatever the depth)
template <int KLO, int KC>
constexpr int t16_planes()
{
    return HSS_T16_PLANES >= 2 && 160 * 1024 / 4 - kCanonLdsTabFloats - 16 * CanonCfg<KLO, KC>::wave_floats(2) >= t16_ctl_floats(16, kT16MaxSlots / 2) ? 2 : 1;
}
template <int KLO, int KC>
constexpr int t16_room() { return 160 * 1024 / 4 - kCanonLdsTabFloats - 16 * CanonCfg<KLO, KC>::wave_floats(t16_planes<KLO, KC>()); }
template <int KLO, int KC>
constexpr int t16_slots() { return t16_room<KLO, KC>() >= t16_ctl_floats(16, kT16MaxSlots) ? kT16MaxSlots : kT16MaxSlots / 2; }
template <int KLO, int KC>
constexpr int t16_pslots()
{
    constexpr int room = t16_room<KLO, KC>(), sl = t16_slots<KLO, KC>();
    return room >= t16_ctl_floats(32, sl) ? 32 : room >= t16_ctl_floats(16, sl) ? 16 : 0;
}

struct Team16Params {
    const float* x;       // [nsig][xstride]
    float* out;           // [nsig][ncols][2 KC]
    const float* atab;    // f16 operand table + float64 twiddles (kCanonAtabFloats floats), then the offset table (kCanonZcFloats)
    const double* wtab;   // float64 {w, dw'}[128]                } rounding-tie path
    const double* twtab;  // float64 {cos, sin}(2 pi m / 128)     }
    unsigned long long* mail;   // [teams][slots][32 blocks][8] tagged words {tag << 32 | half of a float64 block sum}
    unsigned* status;     // device status word (0 = ok)
    float r2scale_s;      // r2scale of the plan x (constant scale)^2
    float inv_c;          // 1 / constant scale
    int n, nsig, col0, ncols;
    long long xstride;
    int team;             // CUs per team (power of two)
    int cpc_shift;        // log2 of the list positions per CU and signal (ceil(ngroups / team) rounded up to a power of two, <= WPB)
    int slots;            // mailbox / statistics slots (power of two <= kT16MaxSlots)
    unsigned seq;         // launch sequence number of the plan (upper half of the mailbox tags)
    unsigned spin_ticks;  // bound of a wait in 100 MHz ticks
    unsigned* arrive;     // arrival counter of the plan (monotone over launches)
    unsigned arrive_base; // its value before this launch: block identity = arrival number - arrive_base
    unsigned* abort_word; // a wait that ran out of time stores `launch` here; every wave then leaves the kernel
    unsigned* fallbacks;  // pinned host word: the same store, for the host's eyes
    unsigned launch;      // identity of this launch (never 0)
    unsigned* done;       // host_done != nullptr: counter of the blocks whose waves have all stored their last feature (monotone over such launches) ...
    unsigned done_base;   // ... its value before this launch ...
    unsigned* host_done;  // ... and the pinned host word the LAST block stores `launch` to: a host that waits for this exec alone looks at that word
                          //     instead of synchronising the stream (no end-of-kernel flush and completion signal in its way: 6 us of a 35 us call)
    double inv_total, inv_total1;   // 1 / (K ncols), 1 / (K ncols - 1): the two divisions of stats_finish, made once on the host
};


// ---- The held images live in FIXED registers, v104 .. v127 (slot s, pair k: v[104 + 12 s + 2 k : +1]), outside the register allocator's
// reach: the kernel is compiled for 104 VGPRs (amdgpu_num_vgpr) and every access is an inline-assembly statement that names its
// register (the clobber lists make the code object reserve 128).  Left to the allocator, the 24 loop-carried registers were kept in one
// place at the loop's head and in another across the transform: 24 register moves per group, a twentieth of the kernel's vector
// instructions, whatever the source looked like (ring, strict first-in first-out, tied operands: profiles/r05_team_diet.txt).
constexpr int kT16HeldBase = 104;
#define HSS_T16_PAIRS(X) X(0, 104, 105) X(1, 106, 107) X(2, 108, 109) X(3, 110, 111) X(4, 112, 113) X(5, 114, 115) \
                         X(6, 116, 117) X(7, 118, 119) X(8, 120, 121) X(9, 122, 123) X(10, 124, 125) X(11, 126, 127)
// held pair P <- a * b (both packed pairs)
template <int P>
__device__ __forceinline__ void held_put(f2 a, f2 b)
{
#define HSS_T16_PUT(N, LO, HI) if constexpr (P == N) asm volatile("v_pk_mul_f32 v[" #LO ":" #HI "], %0, %1" :: "v"(a), "v"(b) : "v" #LO, "v" #HI);
    HSS_T16_PAIRS(HSS_T16_PUT)
#undef HSS_T16_PUT
}
// (held pair P - m.x) * m.y: the z-score of fsst_normalize_kernel, two roundings
template <int P>
__device__ __forceinline__ f2 held_zscore(f2 m)
{
    f2 d, e;
#define HSS_T16_ZS(N, LO, HI) if constexpr (P == N) asm volatile("v_pk_add_f32 %0, v[" #LO ":" #HI "], %1 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(m));
    HSS_T16_PAIRS(HSS_T16_ZS)
#undef HSS_T16_ZS
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(e) : "v"(d), "v"(m));
    return e;
}

// WPB waves per block (one block per CU), DEPTH group images held per wave (registers): (16, 2) is what the library launches
template <int KLO, int KC, int WPB, int DEPTH>
// (amdgpu_num_vgpr(52): the attribute counts in register PAIRS on this target -- 104 allocatable registers; v104 .. v127 are the held images')
__global__ __launch_bounds__(64 * WPB, WPB / 4) __attribute__((amdgpu_num_vgpr(52))) void fsst_team16_kernel(Team16Params p)
{
#ifdef HSS_T16_BLKPROBE
    const unsigned pb_entry = static_cast<unsigned>(wall_clock64());       // (absolute: the 100 MHz counter is the chip's)
#endif
    using C = CanonCfg<KLO, KC>;
    static_assert(WPB % 4 == 0 && DEPTH >= 1 && DEPTH <= 2, "whole waves per SIMD; two held groups (v104 .. v127)");
    static_assert(DEPTH == 2, "the kernel is compiled for 104 allocatable registers + 24 fixed ones");
    constexpr int K = KC, ATAB = kCanonLdsTabFloats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = p.n;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* atab = smem;
    int* next_q = reinterpret_cast<int*>(smem + ATAB);                       // [0] ticket counter [1] dead [2] identity
    unsigned* dead = reinterpret_cast<unsigned*>(smem + ATAB) + 1;
    unsigned* cls_lds = reinterpret_cast<unsigned*>(smem + ATAB + 16);       // [64] byte i: which of a signal's three float4 statistics float4 lane + 64 i reads
    unsigned* ppk_lds = reinterpret_cast<unsigned*>(smem + ATAB + 80);       // [3][64]
    unsigned* ready = reinterpret_cast<unsigned*>(smem + ATAB + 272);        // [slots] epoch (signal ordinal + 1) of the statistics in fin[]
    constexpr int MS = t16_slots<KLO, KC>(), PSLOTS = t16_pslots<KLO, KC>();
    unsigned* claim = ready + MS;                                            // [slots] epoch some wave of this CU is resolving / has resolved
    float4* fin = reinterpret_cast<float4*>(smem + ATAB + 272 + 2 * MS);     // [slots][3]
    static_assert(PSLOTS >= 16, "the CU's own partials need LDS beside the wave regions");
    float* part_lds = smem + ATAB + t16_ctl_base(MS);                        // [PSLOTS][kT16MaxCpc][6]
    int* pcnt_lds = reinterpret_cast<int*>(part_lds + PSLOTS * kT16MaxCpc * kT16PartFloats);   // [PSLOTS][2] partials delivered per block
    constexpr int PLANES = t16_planes<KLO, KC>();
    float* wbase = smem + ATAB + t16_ctl_floats(PSLOTS, MS) + wv * C::wave_floats(PLANES);
    u2* xrec = reinterpret_cast<u2*>(wbase);
    f2* own_first = reinterpret_cast<f2*>(wbase + 2 * kCanonRecs);
    int* flag = reinterpret_cast<int*>(own_first + PLANES * 16 * C::LD);
    int* tq = flag + kCanonFlagWords;

    // Block identity = ARRIVAL number ("Giving up" above).  Asked for FIRST: the trip to the counter (1.5-2 us) runs beside the table's loads, not behind them
    unsigned arrived = 0u;
    if (threadIdx.x == 64) arrived = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.arrive_base;
    for (int i = threadIdx.x; i < ATAB; i += 64 * WPB) atab[i] = p.atab[i];
    if (lane < kCanonFlagWords) flag[lane] = 0;
    if (lane < kCanonTieWords) tq[lane] = 0;
    if (threadIdx.x < 16 && threadIdx.x != 2) next_q[threadIdx.x] = 0;       // ([2]: the identity, written below)
    for (int i = threadIdx.x; i < 2 * MS; i += 64 * WPB) ready[i] = 0u;       // ready[], claim[]
    if (threadIdx.x < 2 * PSLOTS) pcnt_lds[threadIdx.x] = 0;
    if (threadIdx.x == 64) next_q[2] = static_cast<int>(arrived);
    if (wv == 0) {
        unsigned cofs = 0u;                              // byte i: 16 x (number of imaginary column pairs of float4 lane + 64 i)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned c = (4u * static_cast<unsigned>(lane + 64 * i)) % static_cast<unsigned>(2 * K);
            const unsigned nim = (c >= static_cast<unsigned>(K) ? 1u : 0u) + (c + 2 >= static_cast<unsigned>(K) ? 1u : 0u);
            cofs |= (16u * nim) << (8 * i);
            ppk_lds[i * 64 + lane] = canon_store_offsets<KLO, KC>(lane + 64 * i);
        }
        cls_lds[lane] = cofs;
    }
    __syncthreads();

    // cold parameters are re-read from the kernel arguments where they are used (rare paths), the hot ones stay in scalar registers
    using kparams = const __attribute__((address_space(4))) Team16Params;
    kparams* const kp_ = (kparams*)__builtin_amdgcn_kernarg_segment_ptr();
    auto P = [&]() -> kparams* { kparams* q = kp_; asm volatile("" : "+s"(q)); return q; };

    auto aborted = [&]() -> bool {
        return __hip_atomic_load(P()->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P()->launch;
    };
    auto gave_up = [&]() {
        if (lane == 0) {
            __hip_atomic_store(P()->abort_word, P()->launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((gu32*)(P()->fallbacks), P()->launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto is_dead = [&]() -> bool { return __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u; };
    // A wave that finds the launch given up ENDS where it stands (s_endpgm from inline assembly: no control-flow edge).  With
    // `return`s in the middle of the main loop the compiler's single-exit form of that loop merged the leaving paths into its
    // latch and paid for it on the way round: all 24 held registers moved twice per group, and a wait for the just-requested
    // samples (their registers were "merged" too).
    auto leave = [&]() { asm volatile("s_endpgm" ::: "memory"); };
    auto expired = [&](unsigned since) -> bool {
        return static_cast<unsigned>(wall_clock64()) - since > P()->spin_ticks || is_dead() || aborted();
    };

    // ---- team geometry (wave-uniform): T consecutive identities form a team
    const int virt = __builtin_amdgcn_readfirstlane(next_q[2]);
    // an identity outside the grid = arrivals of two launches of one plan interleaved (a plan is single-stream: hssfsst.h):
    // give the launch up instead of indexing outside the mailboxes (the gated fallback computes the exec)
    if (static_cast<unsigned>(virt) >= gridDim.x) { gave_up(); return; }
    // (a block that starts after the launch was given up: the look at the abort word is ASKED here and looked at behind the first draw -- its trip
    //  runs beside the first tile's samples', not in front of them; asked with the arrival's trip, its answer handed on through LDS: slower)
    const unsigned abort_seen = __hip_atomic_load(P()->abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int T = p.team, cpcs = p.cpc_shift, cpc = 1 << cpcs;
    const int member = virt & (T - 1), team = virt / T;
    const int nteams = static_cast<int>(gridDim.x) / T;
    const int nk = (p.nsig > team) ? (p.nsig - team + nteams - 1) / nteams : 0;      // signals of this team
    const int nwork = nk << cpcs;
    const int ncols = p.ncols, cend = p.col0 + p.ncols;
    const int G = (ncols + 15) >> 4;
    const int cg0 = p.col0 >> 4;                         // (the host sends only column ranges that start on a group boundary)
    const int smask = p.slots - 1;
    constexpr int nwords = kT16SlotWords;                      // tagged words per signal in the mailbox
    gu64* mail = (gu64*)(p.mail) + static_cast<size_t>(team) * static_cast<size_t>(p.slots) * nwords;
    const int nblocks = (G + kStatBlock - 1) / kStatBlock;
    const unsigned sig_bytes = static_cast<unsigned>(ncols) * (2 * K * 4);         // a signal's feature block (at most 128 groups: 32 bits)

    f2 tiny = {1.0e-37f, 0.0f};
    asm volatile("" : "+s"(tiny));

    // ---- draw: the next group of this CU's list and its tile's samples on their way into registers
    float sreg[3];
    bool d_valid = false, saw_dead = false;
    int ko_d = 0, g_d = 0, q_last = -1;                  // (q_last: the last ticket this wave drew)
    // after_ko >= 0: draw only if every position still to be handed out belongs to a signal AFTER after_ko (see "Progress":
    // a ticket the wave holds unpublished while it waits must not belong to the signal it waits for).  The counter only grows and
    // stands behind this wave's last ticket: when the position after that one already lies in a later signal there is nothing to
    // look at (one LDS round trip instead of two).  The block's `dead` word rides along with the ticket.
    // (in two halves: the LDS operations are ISSUED by draw_ask -- the main loop asks right behind the atomic that counts its partial
    //  in, one wait for both round trips -- and the ticket is looked at by draw_take)
    int ask_qi = 0x7fffffff;
    unsigned ask_dd = 0u;
    auto draw_ask = [&](int after_ko) {
        ask_qi = 0x7fffffff;
        ask_dd = 0u;
        const bool known = after_ko < 0 || ((q_last + 1) >> cpcs) > after_ko;
        if (lane == 0) {
            const bool go = known || (__hip_atomic_load(next_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> cpcs) > after_ko;
            if (go) ask_qi = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ask_dd = __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto draw_take = [&]() {
        int qi = __builtin_amdgcn_readfirstlane(ask_qi);
        const unsigned dd = ask_dd;
        saw_dead = __builtin_amdgcn_readfirstlane(dd) != 0u;
        d_valid = false;
        while (qi < nwork) {
            q_last = qi;
            ko_d = qi >> cpcs;
            g_d = (((member + ko_d) & (T - 1)) << cpcs) + (qi & (cpc - 1));       // CU c' = (member + ko) mod T: groups cpc c' ..
            if (g_d < G) { d_valid = true; break; }
            if (lane == 0) qi = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            qi = __builtin_amdgcn_readfirstlane(qi);
        }
        if (d_valid) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            // (signal index and stride are 32-bit here -- the host sends nothing else to this kernel: a 32 x 32 -> 64-bit product, not 64 x 64)
            const float* xsig = P()->x + static_cast<unsigned long long>(static_cast<unsigned>(team + ko_d * nteams)) * static_cast<unsigned>(P()->xstride);
            canon_fetch(xsig, n, ((g_d + cg0) & ~3) * 16, lane_o, sreg);
        }
    };
    auto draw = [&](int after_ko) { draw_ask(after_ko); draw_take(); };

#ifdef HSS_T16_BLKPROBE
    unsigned pb_miss = 0u, pb_blocked = 0u, pb_fin = 0u, pb_nfin = 0u;
    const unsigned long long pb_c0 = __builtin_readcyclecounter(), pb_r0 = wall_clock64();
#endif
    auto stats_ready = [&](int ko) -> bool {             // the CU already has this signal's statistics
        unsigned have = 0u;
        if (lane == 0) have = __hip_atomic_load(ready + (ko & smask), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(have)) == static_cast<unsigned>(ko) + 1u;
    };
    auto try_claim = [&](int ko) -> bool {
        unsigned mine = 0u;
        const unsigned epoch = static_cast<unsigned>(ko) + 1u;
        if (lane == 0) mine = __hip_atomic_fetch_max(claim + (ko & smask), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch ? 1u : 0u;
        return __builtin_amdgcn_readfirstlane(mine) != 0u;
    };

    // ---- Statistics of signal ordinal ko (of this team): {mean_re, 1/std_re, mean_im, 1/std_im}, once per signal and CU.
    // The mailbox holds the signal's BLOCK sums -- float64 {sum re, sum re^2, sum im, sum im^2} of every block of kStatBlock groups,
    // formed by the CU that transformed the block with the arithmetic of signal_stats()' inner loop (piece_moment, pieces in order)
    // -- as tagged halves.  Lane (blk % 16, q) of the wave that has claimed the signal fetches quantity q of the blocks blk and
    // blk + 16 (its 16-byte pairs l and l + 64: the mailbox is laid out for exactly that), adds them in that order and runs
    // stats_finish: the instructions of stats_from_blocks() on the numbers of the two-launch path.
    // the claim is this wave's: look at the mailbox until both of the lane's blocks are there
    auto blocks_to_stats = [&](int ko, unsigned t0, int lane_r) __attribute__((always_inline)) -> float4 {
        const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko) & 0xffffu);
        const gu64* slot = mail + static_cast<size_t>(ko & smask) * nwords;
        const int blk0 = lane_r >> 2;
        unsigned need = (blk0 < nblocks ? 1u : 0u) | (blk0 + 16 < nblocks ? 2u : 0u);
        double sb[2] = {0.0, 0.0};
        for (unsigned polls = 0;; ++polls) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                if ((need >> rb) & 1u) {
                    const gu64* q = slot + 2 * (lane_r + 64 * rb);
                    const unsigned long long hi = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long lo = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (static_cast<unsigned>(hi >> 32) == tag && static_cast<unsigned>(lo >> 32) == tag) {
                        sb[rb] = __longlong_as_double(static_cast<long long>((hi << 32) | (lo & 0xffffffffull)));
                        need &= ~(1u << rb);
                    }
                }
            if (__builtin_amdgcn_ballot_w64(need != 0u) == 0ull) break;
            if ((polls & 7u) == 7u && expired(t0)) { gave_up(); leave(); }
            __builtin_amdgcn_s_sleep(8);
        }
        // stats_from_blocks(): lane (blk % 16, q) adds its blocks blk, blk + 16 in that order, then stats_finish
        double acc = 0.0;
        if (blk0 < nblocks) acc += sb[0];
        if (blk0 + 16 < nblocks) acc += sb[1];
        static_assert(kT16MaxBlocks <= 32, "a lane sums at most two blocks");
        return stats_finish_lead(acc, P()->inv_total, P()->inv_total1, lane_r);
    };
    auto resolve_owned = [&](int ko, unsigned t0, int lane_r) __attribute__((always_inline)) {
        // (fifteen siblings and, soon, other CUs wait for what this wave does now: it goes first on its SIMD)
        __builtin_amdgcn_s_setprio(3);
        float4 r;
        r = blocks_to_stats(ko, t0, lane_r);
        if (lane == 0) {
            float4* f3 = fin + 3 * (ko & smask);
            f3[0] = make_float4(r.x, r.y, r.x, r.y);
            f3[1] = make_float4(r.x, r.y, r.z, r.w);
            f3[2] = make_float4(r.z, r.w, r.z, r.w);
            __hip_atomic_store(ready + (ko & smask), static_cast<unsigned>(ko) + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_sync();
        __builtin_amdgcn_s_setprio(0);
    };
    // a wave cannot go on without a signal's statistics: a sibling's result, or this wave resolves (a wave that finds the launch
    // given up does not come back)
    auto signal_statistics = [&](int ko) {
#if defined(HSS_T16_ABLATE) && (HSS_T16_ABLATE == 1 || HSS_T16_ABLATE == 2)      // development: nobody waits, nobody resolves (results invalid)
        return;
#endif
        if (stats_ready(ko)) return;
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned t0 = static_cast<unsigned>(wall_clock64());
#ifdef HSS_T16_BLKPROBE
        ++pb_miss;
        struct Fin { unsigned& acc; unsigned t; __device__ ~Fin() { acc += static_cast<unsigned>(wall_clock64()) - t; } } fin_{pb_blocked, t0};
#endif
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE == 3      // development: the resolver resolves, nobody else waits (results invalid)
        if (try_claim(ko)) resolve_owned(ko, t0, lane_r);
        return;
#endif
        for (unsigned spins = 0;; ++spins) {
            if ((spins & 15u) == 0u && try_claim(ko)) {
#ifdef HSS_T16_BLKPROBE
                const unsigned tr = static_cast<unsigned>(wall_clock64());
                resolve_owned(ko, t0, lane_r);
                pb_fin += static_cast<unsigned>(wall_clock64()) - tr; ++pb_nfin;
                return;
#else
                resolve_owned(ko, t0, lane_r); return;
#endif
            }
            if ((spins & 31u) == 31u && expired(t0)) { gave_up(); leave(); }
            if (is_dead()) leave();
            __builtin_amdgcn_s_sleep(4);
            if (stats_ready(ko)) return;
        }
    };

    // z-score of a held group from registers -- (v - mean) * (1 / std), two roundings, exactly as fsst_normalize_kernel -- and
    // its 3 streaming stores per lane.  {mean, 1 / std} of a float4's two column pairs come as ONE 16-byte LDS read: the signal's
    // statistics lie there three times -- (re, re), (re, im), (im, im) -- and which one float4 lane + 64 i needs is a constant of
    // the lane (cls_lds): no per-element selects.  The store address is scalar base + lane offset.
    auto emit_held = [&](auto SL, int ko_h, int g_h) {
        constexpr int sl = decltype(SL)::value;
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned cofs = cls_lds[lane_r];
        const char* tb = reinterpret_cast<const char*>(fin + 3 * (ko_h & smask));
        char* obase = reinterpret_cast<char*>(P()->out) + static_cast<unsigned long long>(static_cast<unsigned>(team + ko_h * nteams)) * sig_bytes +
                      static_cast<unsigned>(g_h * (16 * 2 * K * 4));                                                          // (wave-uniform)
        const unsigned voff = static_cast<unsigned>(lane_r) * 16u;
        const int nvalid = min(16, ncols - g_h * 16);
        auto put = [&](auto I) {
            constexpr int i = decltype(I)::value;
            const float4 tt = *reinterpret_cast<const float4*>(tb + ((cofs >> (8 * i)) & 0xffu));
            const f2 lo = held_zscore<6 * sl + 2 * i>(f2{tt.x, tt.y});
            const f2 hi = held_zscore<6 * sl + 2 * i + 1>(f2{tt.z, tt.w});
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE >= 2      // development: the arithmetic without the stores
            { f2 l2 = lo, h2 = hi; asm volatile("" :: "v"(l2), "v"(h2)); }
#else
            __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(obase + (voff + 1024u * static_cast<unsigned>(i))));
#endif
        };
        if (__builtin_expect(nvalid == 16, 1)) {
            static_for<3>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (64 * (i + 1) <= 8 * K) put(I);
                else if constexpr (64 * i < 8 * K) { if (lane_r + 64 * i < 8 * K) put(I); }
            });
        } else {
            asm volatile("");
            const int lim = nvalid * (K >> 1);
            static_for<3>([&](auto I) { if (lane_r + 64 * decltype(I)::value < lim) put(I); });
        }
    };

    // ---- the held groups: images in registers (feature units, un-normalised), waiting for their signals' statistics.  A strict
    // first-in first-out of DEPTH slots: the steps take the slots in turn (`slot`, a scalar; the two slots' code stands side by
    // side behind one scalar branch), a step fills its slot after the group that sat there -- the oldest the wave holds -- has
    // left.  (A ring with early releases cost a dozen register moves per group at the loop head and a page of scalar bookkeeping,
    // nothing is gained by a group leaving early; the loop unrolled DEPTH times with static slots put an image into scratch.)
    int nheld = 0;
    int ko_hs[DEPTH], g_hs[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ko_hs[d] = 0; g_hs[d] = 0; }

    // One step, in this order -- a wave's memory operations retire in order, so the ONE wait for loaded data per group (the next
    // tile's samples) sits where everything else in flight -- the previous group's stores -- is a whole transform old:
    //   1. transform the landed group;
    //   2. land the drawn group's tile (its records are free: every group stages its own tile);
    //   3. statistics partial of the transformed group -> LDS, a block's last partial -> mailbox; draw a further group and request
    //      its samples -- IF that is safe: a ticket the wave holds unpublished across its waits must belong to a later signal than
    //      any it may wait for.  Otherwise the wave draws when it has nothing landed -- holding only published groups (rare: the
    //      slow path at the top);
    //   4. the group in this step's slot leaves: its signal's statistics are waited for, z-score from registers, 3 stores;
    //   5. the transformed group's image: own plane -> the slot.
    bool c_valid = false;                                // a group is landed: (ko, g), its tile in xrec
    int ko = 0, g = 0;
    CanonTile tile{};
    auto land = [&]() {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        tile = canon_land<true>(sreg, xrec, P()->r2scale_s, P()->inv_c, lane_t, ((g_d + cg0) & ~3) * 16, n);
        ko = ko_d; g = g_d; c_valid = true; d_valid = false;
    };
    int slot = 0;                                        // the slot this step fills (steps take the slots in turn)
    // Two planes (where the LDS has them: the displaced plane's 47 kB, fsst_canon128.hpp "One plane").  The steps write the planes in
    // turn, and a step's image stays in its plane for a whole step: it moves into the registers at the END OF THE NEXT step, when the
    // other plane holds that step's image.  A group's statistics are therefore wanted THREE steps after its partial was published
    // instead of two -- a third held group at no instruction and no register (a third image in registers spilled, parked in global
    // memory it cost 6 %: profiles/r05_team_diet.txt) --, which is what the waits for statistics, 4-5 % of the kernel, were short of.
    int cur = 0;                                         // the plane this step transforms into
    bool p_valid = false;                                // the OTHER plane holds the previous step's image: group (p_ko, p_g), scale p_inv
    int p_ko = 0, p_g = 0;
    float p_inv = 0.0f;
    draw(-1);
    if (saw_dead || abort_seen == P()->launch) return;
    if (d_valid) { land(); draw(-1); }                   // (the second ticket is transformed and published before the wave's first wait)
    for (;;) {
        if (saw_dead) leave();
        if (!c_valid) {                                  // slow path: nothing landed -- the wave holds nothing unpublished
            if (!d_valid) { draw(-1); if (saw_dead) leave(); }
            if (!d_valid) break;                         // the list is done (the loop's only exit)
            land();
        }
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const unsigned b = static_cast<unsigned>(team + ko * nteams);
        const int tg = P()->col0 + g * 16;
#ifndef HSS_T16_NO_LAGPRIO
        {   // a group of a signal the CU's ticket counter has left behind is what other waves will soon wait for: it goes first
            // (what the wave's own next ticket says about the counter -- a group time old, but no trip to LDS)
            const int lag = (d_valid ? ko_d : ko + 2) - ko;
            if (lag >= 2) __builtin_amdgcn_s_setprio(2); else if (lag == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);      // (3 / 2: slower; off: +2.8 %)
        }
#endif
        f2* own_base = own_first + cur * (16 * C::LD);
        canon_group<KLO, KC, HSS_T16_TAPB, true>(xrec + ((g + cg0) & 3) * 16, atab, own_base, flag, tq, P()->wtab, P()->twtab, tile, tiny, lane_o,
                                           [&]() -> const float* { return P()->x + static_cast<unsigned long long>(b) * static_cast<unsigned>(P()->xstride); }, n, tg, P()->atab + kCanonAtabFloats);
        const float inv_cur = tile.inv;
        const int ko_cur = ko, g_cur = g;
        c_valid = false;
        // The step's ONE wait for memory, said out loud.  Outstanding here: the next tile's samples and the previous group's stores, both a
        // whole transform old.  Left to the compiler the wait sat in the middle of the fold, a hundred instructions after those samples and
        // stores were issued: its wait-count pass merges what may be pending over every path through the loop, and a register reload
        // inside the transform's rare paths made it protect the fold's registers at the head of the loop (HSS_RARE_VMEM_DONE, fsst_canon128.hpp).
        HSS_RARE_VMEM_DONE();
        if (d_valid) land();
        // ---- statistics partial -> the CU's LDS (rows 0..3 of the wave hold S1re / S2re / S1im / S2im, every lane the pivot); the
        //      wave that delivers a block's last partial forms the block's float64 sums -- signal_stats()' inner loop: the block's
        //      pieces in order, piece_moment -- and publishes them in the team's mailbox as eight tagged words
        {
            const int nvalid = min(16, cend - tg);
            f2 piv;
            const float w = canon_stats<KLO, KC>(own_base, nvalid, inv_cur, lane_o, piv);
            const int pos = g_cur & (cpc - 1);           // position among the CU's groups of this signal
            const int ps = ko_cur & (PSLOTS - 1);
            float* pe = part_lds + (ps * kT16MaxCpc + pos) * kT16PartFloats;
            // the first lane of each row writes its row's sum, lane 0 the pivot pair: two stores under literal exec masks (the
            // predicates as compares and selects were 20 instructions).  Behind them, still lane 0 alone and in ONE statement: the atomic
            // that counts the partial in, the atomic that draws the next ticket (draw_ask's, when the counter need not be looked at
            // first) and the block's `dead` word -- one wait for the three round trips.  (As builtins the compiler's atomic optimiser
            // made each a wave reduction that is read back on the spot, one round trip after the other.)  LDS operations of a wave
            // execute in order: the partial is there before it is counted.
            const unsigned pa = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)pe)) + (static_cast<unsigned>(lane_o) >> 4) * 4u;
            const unsigned pcnt_a = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) int*)(pcnt_lds + 2 * ps + (pos >> 2))));
            int before = 0;
            const bool known = ((q_last + 1) >> cpcs) > ko_cur;
            if (__builtin_expect(known, 1)) {
                const unsigned nq_a = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) int*)next_q));
                unsigned one = 1u, before_v, qi_v, dd_v;
                unsigned long long keep;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, 0x00010001\n\ts_mov_b32 exec_hi, 0x00010001\n\tds_write_b32 %4, %5\n\t"
                             "s_mov_b64 exec, 1\n\tds_write_b64 %4, %6 offset:16\n\t"
                             "ds_add_rtn_u32 %1, %7, %9\n\tds_add_rtn_u32 %2, %8, %9\n\tds_read_b32 %3, %8 offset:4\n\t"
                             "s_mov_b64 exec, %0\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(keep), "=&v"(before_v), "=&v"(qi_v), "=&v"(dd_v)
                             : "v"(pa), "v"(w), "v"(piv), "v"(pcnt_a), "v"(nq_a), "v"(one) : "memory");
                before = static_cast<int>(before_v);
                ask_qi = static_cast<int>(qi_v);         // (lane 0's are the values: draw_take reads the first lane)
                ask_dd = dd_v;
                wave_sync();
            } else {
                unsigned long long keep;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, 0x00010001\n\ts_mov_b32 exec_hi, 0x00010001\n\tds_write_b32 %1, %2\n\t"
                             "s_mov_b64 exec, 1\n\tds_write_b64 %1, %3 offset:16\n\ts_mov_b64 exec, %0"
                             : "=&s"(keep) : "v"(pa), "v"(w), "v"(piv) : "memory");
                wave_sync();
                if (lane == 0) before = __hip_atomic_fetch_add(pcnt_lds + 2 * ps + (pos >> 2), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                draw_ask(ko_cur);
            }
            const int blk = g_cur >> 2, bfirst = blk << 2;                       // kStatBlock = 4
            const int expect = min(kStatBlock, G - bfirst);
            if (__builtin_amdgcn_readfirstlane(before) + 1 == expect) {
                __builtin_amdgcn_s_setprio(3);           // (a signal's statistics wait for its last block)
                if (lane == 0) __hip_atomic_store(pcnt_lds + 2 * ps + (pos >> 2), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                wave_sync();
                int lane_b = lane;
                asm volatile("" : "+v"(lane_b));
                // lane (pc, q) of a row forms quantity q of the block's piece pc -- the four pieces side by side, not one after the other --
                // and lanes 0..3 add them IN ORDER, ((0 + m0) + m1) + m2) + m3, the sum of signal_stats()' loop to the last bit
                const int q = lane_b & 3, h = q >> 1, pc = (lane_b >> 2) & 3;
                const float* pb = part_lds + (ps * kT16MaxCpc + (pos & ~3) + pc) * kT16PartFloats;
                const double cnt = static_cast<double>(min(16, ncols - 16 * (bfirst + pc))) * static_cast<double>(K);
                const double mom = piece_moment(q, static_cast<double>(pb[2 * h]), static_cast<double>(pb[2 * h + 1]), static_cast<double>(pb[4 + h]), cnt);
                auto from_row = [&](int ctrl) -> double {  // lane i <- lane i + 4 / 8 / 12 of its row (row_shl)
                    const long long b = __double_as_longlong(mom);
                    unsigned lo, hi;
                    if (ctrl == 4) { lo = __builtin_amdgcn_update_dpp(0u, static_cast<unsigned>(b), 0x104, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0u, static_cast<unsigned>(b >> 32), 0x104, 0xf, 0xf, true); }
                    else if (ctrl == 8) { lo = __builtin_amdgcn_update_dpp(0u, static_cast<unsigned>(b), 0x108, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0u, static_cast<unsigned>(b >> 32), 0x108, 0xf, 0xf, true); }
                    else { lo = __builtin_amdgcn_update_dpp(0u, static_cast<unsigned>(b), 0x10c, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0u, static_cast<unsigned>(b >> 32), 0x10c, 0xf, 0xf, true); }
                    return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
                };
                double sbk = 0.0;
                sbk += mom;                               // (lanes 0..3: pc = 0; `expect` >= 1)
                if (expect > 1) sbk += from_row(4);
                if (expect > 2) sbk += from_row(8);
                if (expect > 3) sbk += from_row(12);
                const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko_cur) & 0xffffu);
                const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(sbk));
                gu64* e = mail + static_cast<size_t>(ko_cur & smask) * nwords + (blk * 4 + q) * 2;
                if (lane_b < 4) {
                    __hip_atomic_store(e, (static_cast<unsigned long long>(tag) << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(e + 1, (static_cast<unsigned long long>(tag) << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_s_setprio(0);
            }
        }
        draw_take();
        // ---- this step's slot: the group that sits there (the oldest the wave holds) leaves, the new group's image moves in.
        // Two trips to LDS for all of it (as separate steps -- ready word; column classes; three statistics entries, each waited for;
        // per float4 of the image its offsets, then its cells -- it was eleven, one behind the other, a tenth of the wave's time per group):
        // first the words that only depend on the lane and the slot -- is the leaving group's signal resolved, which statistics entry and
        // which cells each of the lane's three float4 takes --, then, behind one wait, the statistics entries and the new image's cells.
        // (with two planes the image that moves in is the PREVIOUS step's, from the other plane)
        auto move_in = [&](const f2* src_plane, float inv_src, int ko_src, int g_src) {
            int ko_o = 0, g_o = 0;
            static_for<DEPTH>([&](auto S) { if (slot == decltype(S)::value) { ko_o = ko_hs[decltype(S)::value]; g_o = g_hs[decltype(S)::value]; } });
            const bool full = nheld == DEPTH;
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
            const unsigned cofs = cls_lds[lane_r];
            const unsigned pk0 = ppk_lds[lane_r], pk1 = ppk_lds[64 + lane_r], pk2 = ppk_lds[128 + lane_r];
            const unsigned have = __hip_atomic_load(ready + (ko_o & smask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (every lane: a broadcast read)
            if (full) {
#if !(defined(HSS_T16_ABLATE) && (HSS_T16_ABLATE == 1 || HSS_T16_ABLATE == 2))
                if (static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(have))) != static_cast<unsigned>(ko_o) + 1u) signal_statistics(ko_o);
#endif
            } else ++nheld;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            unsigned obase = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)reinterpret_cast<const float*>(src_plane)));
            asm volatile("" : "+s"(obase));
            auto cell = [&](unsigned off) -> f2 {
                const lds_float* q = (const lds_float*)static_cast<size_t>(obase + off);
                return f2{q[0], q[2]};
            };
            const f2 lo0 = cell(pk0 & 0xffffu), hi0 = cell(pk0 >> 16), lo1 = cell(pk1 & 0xffffu), hi1 = cell(pk1 >> 16),
                     lo2 = cell(pk2 & 0xffffu), hi2 = cell(pk2 >> 16);
            const f2 sc = {inv_src, inv_src};
            static_for<DEPTH>([&](auto S) {
                constexpr int sl = decltype(S)::value;
                if (slot == sl) {
                    if (full) {
                        const char* tb = reinterpret_cast<const char*>(fin + 3 * (ko_o & smask));
                        const float4 t0 = *reinterpret_cast<const float4*>(tb + (cofs & 0xffu));
                        const float4 t1 = *reinterpret_cast<const float4*>(tb + ((cofs >> 8) & 0xffu));
                        const float4 t2 = *reinterpret_cast<const float4*>(tb + ((cofs >> 16) & 0xffu));
                        char* ob = reinterpret_cast<char*>(P()->out) + static_cast<unsigned long long>(static_cast<unsigned>(team + ko_o * nteams)) * sig_bytes +
                                   static_cast<unsigned>(g_o * (16 * 2 * K * 4));                                              // (wave-uniform)
                        const unsigned voff = static_cast<unsigned>(lane_r) * 16u;
                        const int nvalid = min(16, ncols - g_o * 16);
                        auto put = [&](auto I, float4 tt) {
                            constexpr int i = decltype(I)::value;
                            const f2 l = held_zscore<6 * sl + 2 * i>(f2{tt.x, tt.y});
                            const f2 h = held_zscore<6 * sl + 2 * i + 1>(f2{tt.z, tt.w});
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE >= 2      // development: the arithmetic without the stores
                            { f2 l2 = l, h2 = h; asm volatile("" :: "v"(l2), "v"(h2)); }
#else
                            __builtin_nontemporal_store(f4{l.x, l.y, h.x, h.y}, reinterpret_cast<f4*>(ob + (voff + 1024u * static_cast<unsigned>(i))));
#endif
                        };
                        const float4 tts[3] = {t0, t1, t2};
                        if (__builtin_expect(nvalid == 16, 1)) {
                            static_for<3>([&](auto I) {
                                constexpr int i = decltype(I)::value;
                                if constexpr (64 * (i + 1) <= 8 * K) put(I, tts[i]);
                                else if constexpr (64 * i < 8 * K) { if (lane_r + 64 * i < 8 * K) put(I, tts[i]); }
                            });
                        } else {
                            asm volatile("");
                            const int lim = nvalid * (K >> 1);
                            static_for<3>([&](auto I) { if (lane_r + 64 * decltype(I)::value < lim) put(I, tts[decltype(I)::value]); });
                        }
                    }
                    held_put<6 * sl + 0>(lo0, sc); held_put<6 * sl + 1>(hi0, sc);
                    held_put<6 * sl + 2>(lo1, sc); held_put<6 * sl + 3>(hi1, sc);
                    held_put<6 * sl + 4>(lo2, sc); held_put<6 * sl + 5>(hi2, sc);
                    ko_hs[sl] = ko_src; g_hs[sl] = g_src;
                }
            });
            slot = (slot + 1 == DEPTH) ? 0 : slot + 1;
        };
        if constexpr (PLANES == 2) {
            if (p_valid) move_in(own_first + (cur ^ 1) * (16 * C::LD), p_inv, p_ko, p_g);
            p_valid = true; p_inv = inv_cur; p_ko = ko_cur; p_g = g_cur;
            cur ^= 1;
        } else
            move_in(own_base, inv_cur, ko_cur, g_cur);
        wave_sync();
    }
    // ---- the list is done: the image that still sits in its plane moves in (the oldest held group leaves for it) ...
    if constexpr (PLANES == 2) {
        if (p_valid) {
            int ko_o = 0, g_o = 0;
            static_for<DEPTH>([&](auto S) { if (slot == decltype(S)::value) { ko_o = ko_hs[decltype(S)::value]; g_o = g_hs[decltype(S)::value]; } });
            (void)g_o;
            const f2* src_plane = own_first + (cur ^ 1) * (16 * C::LD);
            const bool full = nheld == DEPTH;
            if (full) signal_statistics(ko_o); else ++nheld;
            static_for<DEPTH>([&](auto S) {
                constexpr int sl = decltype(S)::value;
                if (slot == sl) {
                    if (full) emit_held(S, ko_hs[sl], g_hs[sl]);
                    int lane_r = lane;
                    asm volatile("" : "+v"(lane_r));
                    unsigned obase = static_cast<unsigned>(reinterpret_cast<size_t>((lds_float*)reinterpret_cast<const float*>(src_plane)));
                    const f2 sc = {p_inv, p_inv};
                    static_for<3>([&](auto I) {
                        constexpr int i = decltype(I)::value;
                        const unsigned pk = ppk_lds[i * 64 + lane_r];
                        const lds_float* q0 = (const lds_float*)static_cast<size_t>(obase + (pk & 0xffffu));
                        const lds_float* q1 = (const lds_float*)static_cast<size_t>(obase + (pk >> 16));
                        held_put<6 * sl + 2 * i>(f2{q0[0], q0[2]}, sc);
                        held_put<6 * sl + 2 * i + 1>(f2{q1[0], q1[2]}, sc);
                    });
                    ko_hs[sl] = p_ko; g_hs[sl] = p_g;
                }
            });
            slot = (slot + 1 == DEPTH) ? 0 : slot + 1;
        }
    }
    // ---- the list is done: the held groups leave, oldest first
    for (int i = 0; i < nheld; ++i) {
        const int so = (slot + DEPTH - nheld + i) % DEPTH;
        static_for<DEPTH>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            if (so == sl) { signal_statistics(ko_hs[sl]); emit_held(S, ko_hs[sl], g_hs[sl]); }
        });
    }
    // ---- a host that waits for this launch alone: a wave counts itself in (LDS) once its stores have left it; the block's last wave makes the
    //      block's stores visible at system scope -- the features may lie in pinned host memory; one L2 write-back per block, not per wave --
    //      and counts the block in; the grid's last block says so in pinned host memory.  (A launch that was given up never gets here: the
    //      host sees the give-up word instead.)
    if (P()->host_done != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        unsigned last = 0u;
        if (lane == 0) last = __hip_atomic_fetch_add(next_q + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == WPB - 1 ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane(last) != 0u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // (system scope)
            if (lane == 0) {
                const unsigned before = __hip_atomic_fetch_add(P()->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P()->done_base;
                if (before + 1u == gridDim.x) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
                    __hip_atomic_store((gu32*)(P()->host_done), P()->launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
#ifdef HSS_T16_BLKPROBE
    if (lane == 0 && virt < 256) {
        unsigned* e = g_t16_blk + (virt * 16 + wv) * 8;
        e[0] = pb_miss; e[1] = pb_entry; e[7] = static_cast<unsigned>(wall_clock64()); e[2] = pb_blocked; e[3] = pb_fin; e[4] = pb_nfin;
        e[5] = static_cast<unsigned>(__builtin_readcyclecounter() - pb_c0); e[6] = static_cast<unsigned>(wall_clock64() - pb_r0);
    }
#endif
}

}  // namespace hssfsst
