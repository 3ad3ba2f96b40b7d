// fsst_teamq.hpp -- fsst_team16_kernel (fsst_team16.hpp: the canonical-band transform with the z-score of FSST._stack_real_imag,
// /root/reference/hss/transforms/synchrosqueeze.py:78-85, applied in registers, features written once) with the work handed out PER TEAM:
//
//   * one ticket counter per team in global memory: ticket t = group t mod G of the team's signal t / G.  Every wave of the team's
//     T CUs draws its groups there, one step ahead (the request leaves in the middle of a transform, its answer is back behind it,
//     the samples of that group are requested then and land a step later): a signal's groups go to whichever waves are free, all of
//     them within a fraction of a step, wherever their CUs stand.  (fsst_team16_kernel gives every CU a fixed eight groups of every
//     signal: a team then advances at the pace of its slowest CU -- the waits for statistics, 12 % of the queued kernel,
//     profiles/r06_push_stats.txt -- and a signal's partials are published over two and a half steps.)
//   * a group's statistics partial -- the six float32 numbers of the two-launch path -- goes to the team's mailbox as six tagged
//     8-byte words (no LDS staging per CU, no block counters: a block's four groups sit on four CUs now);
//   * the wave that transforms a signal's LAST group finishes the signal for the team: it looks at the mailbox until all partials are
//     there, forms the float64 piece moments and the block sums in the order of signal_stats() (fsst_kernels.hpp: pieces of a block
//     one after the other, lane (block % 16, q) over blocks b, b + 16, stats_finish) -- the two-launch path's arithmetic on its
//     numbers, bit-identical statistics -- and leaves {mean, 1/std} x 2 as four tagged words;
//   * every wave asks for the statistics of the group that will leave at the end of its step from the middle of the transform
//     (a 32-byte global_load_lds into its own LDS words: no register held) and z-scores from its own table; only a wave that is early
//     looks again until they are there (bounded; a launch that cannot finish gives itself up as fsst_team16_kernel does).
//
// Progress.  Tickets are handed out in signal order and a wave's tickets only grow.  A wave that waits (for a signal's statistics,
// or -- the finisher -- for its partials) holds at most two unpublished tickets, both younger than every ticket of the signal it
// waits for; all tickets of that signal have been drawn (the wave's own younger ticket proves it) by waves that publish them without
// waiting for that signal or a later one -- except when the team's waves in flight hold fewer tickets than a signal has groups
// (a team squeezed onto a CU or two by other processes): then the wait runs into its bound and the launch gives itself up, the gated
// kernels behind it compute the exec (hssfsst_plan_fallbacks counts).
#pragma once
#include "fsst_team16.hpp"

namespace hssfsst {

constexpr int kTqTicketStride = 64;          // words between two teams' ticket counters
constexpr int kTqMaxTeams = 256;             // counters per row (a row's layout must not depend on the launch's team count: the previous launch cleared it)
constexpr int kTqGroupWords = 8;             // tagged 8-byte words per group in the mailbox: S1re S2re S1im S2im p_re p_im (+ 2 spare: one 64-byte line)

struct TeamqParams {
    const float* x;       // [nsig][xstride]
    float* out;           // [nsig][ncols][2 KC]
    const float* atab;    // f16 operand table + float64 twiddles (kCanonAtabFloats floats), then the offset table (kCanonZcFloats)
    const double* wtab;   // float64 {w, dw'}[128]                } rounding-tie path
    const double* twtab;  // float64 {cos, sin}(2 pi m / 128)     }
    unsigned long long* mail;   // [teams][slots][slot_words] tagged words {tag << 32 | float32}: G groups x 8, then the signal's four statistics (+ 4 spare)
    unsigned* tickets;    // [2][kTqMaxTeams][kTqTicketStride] ticket counters (one per 256 bytes: sixteen counters in one cache line shared ONE memory-side
                          // atomic unit, 820 tickets per microsecond asked of a word that serves 88): this launch uses row tick_par and clears the other one
    float r2scale_s;      // r2scale of the plan x (constant scale)^2
    float inv_c;          // 1 / constant scale
    int n, nsig, col0, ncols;
    long long xstride;
    int team;             // CUs per team (power of two)
    int slots;            // mailbox slots per team (power of two)
    int slot_words;       // 8 G + 8
    int tick_par;         // 0 / 1
    unsigned g_magic;     // floor(2^32 / G) + 1: ticket / G by a multiply
    unsigned seq;         // launch sequence number of the plan (upper half of the mailbox tags)
    unsigned spin_ticks;  // bound of a wait in 100 MHz ticks
    unsigned* arrive;     // arrival counter of the plan (monotone over launches)
    unsigned arrive_base; // its value before this launch: block identity = arrival number - arrive_base
    unsigned* abort_word; // a wait that ran out of time stores `launch` here; every wave then leaves the kernel
    unsigned* fallbacks;  // pinned host word: the same store, for the host's eyes
    unsigned launch;      // identity of this launch (never 0)
    double inv_total, inv_total1;   // 1 / (K ncols), 1 / (K ncols - 1): the two divisions of stats_finish, made once on the host
};

constexpr int kTqStageWords = 8;             // per wave, in the block's control words (BELOW 64 KiB: the LDS target of global_load_lds is M0's sixteen bits)
constexpr int kTqRawFloats = 192;            // per wave, in the control words too: the next tile's samples as they come from memory (global -> LDS, no register)
constexpr int tq_ctl_floats() { return 16 + 64 + 192 + 16 * kTqStageWords + 16 * kTqRawFloats; }
constexpr int kTqWaveStat = 12;              // per wave: three float4 {mean, 1/std} pairs -- (re, re), (re, im), (im, im)
template <int KLO, int KC>
constexpr int tq_wave_floats(int planes) { return CanonCfg<KLO, KC>::wave_floats(planes) + kTqWaveStat; }
template <int KLO, int KC>
constexpr int tq_planes() { return 160 * 1024 / 4 - kCanonLdsTabFloats - tq_ctl_floats() - 16 * tq_wave_floats<KLO, KC>(2) >= 0 ? 2 : 1; }

#if defined(HSS_TQ_BLKPROBE)
__device__ unsigned g_tq_blk[256 * 16 * 8];              // per wave of the LAST launch: misses, -, ticks waited, finisher ticks, signals finished, groups
#endif

template <int KLO, int KC, int WPB, int DEPTH>
__global__ __launch_bounds__(64 * WPB, WPB / 4) __attribute__((amdgpu_num_vgpr(52))) void fsst_teamq_kernel(TeamqParams p)
{
    using C = CanonCfg<KLO, KC>;
    static_assert(2 * 16 * C::LD >= kFusedMaxGroups * 6, "the finisher lays a signal's partials out in the plane its next transform will write");
    static_assert(WPB == 16 && DEPTH == 2, "the kernel is compiled for 104 allocatable registers + 24 fixed ones (v104 .. v127: two held images)");
    constexpr int K = KC, ATAB = kCanonLdsTabFloats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = p.n;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* atab = smem;
    int* ctl = reinterpret_cast<int*>(smem + ATAB);                          // [1] dead [2] identity
    unsigned* dead = reinterpret_cast<unsigned*>(smem + ATAB) + 1;
    unsigned* cls_lds = reinterpret_cast<unsigned*>(smem + ATAB + 16);       // [64] byte i: which of the three statistics float4 the lane's float4 lane + 64 i reads
    unsigned* ppk_lds = reinterpret_cast<unsigned*>(smem + ATAB + 80);       // [3][64]
    unsigned* stage = reinterpret_cast<unsigned*>(smem + ATAB + 272) + wv * kTqStageWords;     // [8] the fetched statistics words {value, tag} x 4
    float* xraw = smem + ATAB + 272 + 16 * kTqStageWords + wv * kTqRawFloats;                  // [192] the drawn tile's samples (lane + 64 k)
    constexpr int PLANES = tq_planes<KLO, KC>();
    float* wbase = smem + ATAB + tq_ctl_floats() + wv * tq_wave_floats<KLO, KC>(PLANES);
    u2* xrec = reinterpret_cast<u2*>(wbase);
    f2* own_first = reinterpret_cast<f2*>(wbase + 2 * kCanonRecs);
    int* flag = reinterpret_cast<int*>(own_first + PLANES * 16 * C::LD);
    int* tq = flag + kCanonFlagWords;
    float4* wstat = reinterpret_cast<float4*>(tq + kCanonTieWords);          // [3] this wave's z-score table of the group that leaves
    static_assert((ATAB + tq_ctl_floats()) * 4 < 65536 && (ATAB + 272) % 4 == 0 && (ATAB + tq_ctl_floats()) % 4 == 0 &&
                  tq_wave_floats<KLO, KC>(PLANES) % 4 == 0 && (2 * kCanonRecs + PLANES * 2 * 16 * C::LD + kCanonFlagWords + kCanonTieWords) % 4 == 0,
                  "16-byte aligned statistics words in the first 64 KiB, 16-byte aligned tables");

    for (int i = threadIdx.x; i < ATAB; i += 64 * WPB) atab[i] = p.atab[i];
    if (lane < kCanonFlagWords) flag[lane] = 0;
    if (lane < kCanonTieWords) tq[lane] = 0;
    if (threadIdx.x < 16 && threadIdx.x != 2) ctl[threadIdx.x] = 0;
    if (lane < 8) stage[lane] = 0u;                                          // (no tag is 0)
    // Block identity = ARRIVAL number (fsst_team16.hpp "Giving up")
    if (threadIdx.x == 64)
        ctl[2] = static_cast<int>(__hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.arrive_base);
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

    using kparams = const __attribute__((address_space(4))) TeamqParams;
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
    auto leave = [&]() { asm volatile("s_endpgm" ::: "memory"); };          // (fsst_team16.hpp: no control-flow edge out of the main loop)
    auto expired = [&](unsigned since) -> bool {
        return static_cast<unsigned>(wall_clock64()) - since > P()->spin_ticks || is_dead() || aborted();
    };

    // ---- team geometry (wave-uniform): T consecutive identities form a team
    const int virt = __builtin_amdgcn_readfirstlane(ctl[2]);
    if (static_cast<unsigned>(virt) >= gridDim.x) { gave_up(); return; }    // (two launches of one plan on different streams: hssfsst.h)
    const int T = p.team;
    const int team = virt / T;
    const int nteams = static_cast<int>(gridDim.x) / T;
    // the next launch's ticket counter: cleared by every block of the team (idempotent), long before that launch starts
    if (threadIdx.x == 0) __hip_atomic_store((gu32*)(p.tickets) + ((p.tick_par ^ 1) * kTqMaxTeams + team) * kTqTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (aborted()) return;
    const int nk = (p.nsig > team) ? (p.nsig - team + nteams - 1) / nteams : 0;      // signals of this team
    const int ncols = p.ncols, cend = p.col0 + p.ncols;
    const int G = (ncols + 15) >> 4;
    const unsigned nwork = static_cast<unsigned>(nk) * static_cast<unsigned>(G);      // tickets of this team
    const int cg0 = p.col0 >> 4;                         // (the host sends only column ranges that start on a group boundary)
    const int smask = p.slots - 1;
    const int nwords = p.slot_words;
    gu64* mail = (gu64*)(p.mail) + static_cast<size_t>(team) * static_cast<size_t>(p.slots) * static_cast<size_t>(nwords);
    const int nblocks = (G + kStatBlock - 1) / kStatBlock;
    const unsigned sig_bytes = static_cast<unsigned>(ncols) * (2 * K * 4);
    gu32* tk_ctr = (gu32*)(p.tickets) + (p.tick_par * kTqMaxTeams + team) * kTqTicketStride;

    f2 tiny = {1.0e-37f, 0.0f};
    asm volatile("" : "+s"(tiny));
#if defined(HSS_TQ_BLKPROBE)
    unsigned pb_miss = 0u, pb_blocked = 0u, pb_fin = 0u, pb_nfin = 0u, pb_groups = 0u;
#endif

    const unsigned xraw_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<size_t>((__attribute__((address_space(3))) float*)xraw))));
    // ---- tickets.  tk_req: lane 0 asks for the team's next ticket (the answer is looked at behind the step's wait for memory);
    //      tk_take: ticket -> (signal ordinal, group), its tile's samples on their way into registers
    unsigned v_tk = 0u;
    bool tk_pending = false, tk_done = false;            // a request is in flight; the counter has run past the team's work
    bool fin_hold = false;                               // the wave has drawn a signal's last group: no further ticket until it has finished that signal
    bool d_valid = false;
    int ko_d = 0, g_d = 0;
    auto tk_req = [&]() {
        if (lane == 0) v_tk = __hip_atomic_fetch_add(tk_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk_pending = true;
    };
    auto tk_take = [&]() {
        const unsigned t = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v_tk)));
        tk_pending = false;
        d_valid = false;
        if (t >= nwork) { tk_done = true; return; }
        unsigned q = __umulhi(t, P()->g_magic);
        unsigned r = t - q * static_cast<unsigned>(G);
        if (r >= static_cast<unsigned>(G)) { r += static_cast<unsigned>(G); --q; }       // (the magic number over-estimates by at most one)
        ko_d = static_cast<int>(q); g_d = static_cast<int>(r); d_valid = true;
        // (the wave that transforms a signal's last group finishes the signal -- a wait for every partial of it -- and must hold nothing
        //  unpublished while it waits: a finisher with a landed group of the next signal but one made that signal wait for this one and a
        //  transform, and so on down the list -- the signals completed two steps apart instead of every half)
        fin_hold = g_d == G - 1;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const float* xsig = P()->x + static_cast<unsigned long long>(static_cast<unsigned>(team + ko_d * nteams)) * static_cast<unsigned>(P()->xstride);
        // The aligned tile's three samples per lane (canon_fetch: a raw buffer of n floats, an offset outside it reads 0 -- oracle step 1's zero
        // padding) go from memory straight into the wave's LDS words: buffer_load ... lds, no register is held across the transform (three were,
        // and with the ticket's a fourth: the allocator spilled THEM in front of every fold, which waits for them there).  From inline assembly:
        // behind the builtin the compiler's wait-count pass waits for the copy in front of the next LDS operation it cannot tell apart from the
        // target -- the fold's first operand read.  Nothing reads xraw before the step's explicit wait.  (A tile that reaches over an end of the
        // signal is cleared first: whether a load outside the buffer writes its zero to LDS is not something to build on.)
        const int t0 = ((g_d + cg0) & ~3) * 16;
        if (__builtin_expect(t0 < 64 || t0 + 128 > n, 0)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) xraw[lane_o + 64 * k] = 0.0f;
            wave_sync();
        }
        const unsigned long long xa = reinterpret_cast<unsigned long long>(xsig);
        u4 desc = {static_cast<unsigned>(xa), static_cast<unsigned>(xa >> 32) & 0xffffu, static_cast<unsigned>(n) * 4u, 0x00020000u};
        desc.x = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(desc.x)));
        desc.y = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(desc.y)));
        const int voff = (t0 + lane_o - 64) * 4;
        // (the instruction's offset goes to the memory address AND to the LDS address: M0 stays)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds\n\t"
                     "buffer_load_dword %0, %1, 0 offen offset:256 lds\n\t"
                     "buffer_load_dword %0, %1, 0 offen offset:512 lds"
                     :: "v"(voff), "s"(desc), "s"(xraw_lds) : "m0", "memory");
    };

    // ---- Statistics of signal ordinal ko (of this team), ONCE per signal and team: the wave of the signal's last group looks at the
    //      mailbox until every group's partial is there and finishes (file header).  Lane (b = lane / 4, q = lane % 4) takes quantity q
    //      of the blocks b and b + 16: three words -- S1, S2 of its part, that part's pivot -- of each of the block's four groups.
    auto finish_signal = [&](int ko, float* fbuf) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(3);
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned t0 = static_cast<unsigned>(wall_clock64());
        const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko) & 0xffffu);
        gu64* slotp = mail + static_cast<size_t>(ko & smask) * static_cast<size_t>(nwords);
        // (i) the signal's 8 G words, lane l the words l + 64 i, eight loads in flight at a time, until every partial is there; a word that
        //     is there goes to fbuf -- the plane this wave's next transform will write, free now -- as a float.  (Lane (block, q) fetching
        //     its own 24 words held 70 registers: inlined in the main loop that made the allocator spill on the hot path.)
        const int nw = G * kTqGroupWords;
        unsigned need = 0u;                              // bit i: word lane + 64 i is still wanted
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int w = lane_r + 64 * i;
            if (w < nw && (w & 7) < 6) need |= 1u << i;
        }
        for (unsigned polls = 0;; ++polls) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (__builtin_amdgcn_ballot_w64(((need >> (8 * half)) & 0xffu) != 0u) == 0ull) continue;
                unsigned long long wd[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int w = min(lane_r + 64 * (8 * half + i), nw - 1);
                    wd[i] = __hip_atomic_load(slotp + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned bit = 1u << (8 * half + i);
                    if ((need & bit) && static_cast<unsigned>(wd[i] >> 32) == tag) {
                        const int w = lane_r + 64 * (8 * half + i);
                        fbuf[(w >> 3) * 6 + (w & 7)] = __uint_as_float(static_cast<unsigned>(wd[i]));      // (six floats per group: 6 G <= 768 fit a plane)
                        need &= ~bit;
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(need != 0u) == 0ull) break;
            if ((polls & 7u) == 7u && expired(t0)) { gave_up(); leave(); }
            if (is_dead()) leave();
            __builtin_amdgcn_s_sleep(4);
        }
        wave_sync();
#if defined(HSS_TQ_BLKPROBE)
        pb_fin += static_cast<unsigned>(wall_clock64()) - t0; ++pb_nfin;
#endif
        // (ii) signal_stats() (fsst_kernels.hpp): lane (b = lane / 4, q = lane % 4) takes quantity q of the blocks b and b + 16: a block's
        //      pieces one after the other, then the blocks in that order, then stats_finish
        const int q = lane_r & 3, h = q >> 1, b16 = lane_r >> 2;
        double acc = 0.0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int blk = b16 + 16 * rb;
            if (blk < nblocks) {
                double s = 0.0;
#pragma unroll
                for (int pc = 0; pc < kStatBlock; ++pc) {
                    const int gg = blk * kStatBlock + pc;
                    if (gg < G) {
                        const float* pp = fbuf + gg * 6;
                        const double cnt = static_cast<double>(min(16, ncols - 16 * gg)) * static_cast<double>(K);
                        s += piece_moment(q, static_cast<double>(pp[2 * h]), static_cast<double>(pp[2 * h + 1]), static_cast<double>(pp[4 + h]), cnt);
                    }
                }
                acc += s;
            }
        }
        static_assert(kFusedMaxGroups / kStatBlock <= 32 && kFusedMaxGroups * kTqGroupWords <= 1024, "a lane sums at most two blocks; sixteen words per lane");
        const float4 r = stats_finish_lead(acc, P()->inv_total, P()->inv_total1, lane_r);
        if (lane == 0) {
            gu64* f = slotp + G * kTqGroupWords;
            const unsigned long long th = static_cast<unsigned long long>(tag) << 32;
            __hip_atomic_store(f + 0, th | __float_as_uint(r.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f + 1, th | __float_as_uint(r.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f + 2, th | __float_as_uint(r.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f + 3, th | __float_as_uint(r.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wave_sync();
        __builtin_amdgcn_s_setprio(0);
    };
    // (From inline assembly: behind the builtin the compiler's wait-count pass puts an s_waitcnt vmcnt(0) in front of the next LDS
    //  operation it cannot tell apart from the copy's target.  Nothing reads `stage` before the step's explicit wait.)
    const unsigned stage_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<size_t>((__attribute__((address_space(3))) unsigned*)stage))));
    auto stats_prefetch = [&](int ko) {
        const gu64* f = mail + static_cast<size_t>(ko & smask) * static_cast<size_t>(nwords) + G * kTqGroupWords;
        if (lane < 2) {
            const gu64* fl = f + 2 * lane;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" :: "v"(fl), "s"(stage_lds) : "m0", "memory");
        }
    };
    // The statistics of signal ko into this wave's z-score table: from the words the wave asked for in time (the rule), else by
    // looking at the signal's four words until they are there -- every wait bounded; a wave that finds the launch given up does not come back.
    auto stats_take = [&](int ko) __attribute__((always_inline)) {
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko) & 0xffffu);
        const u4* st4 = reinterpret_cast<const u4*>(stage);
        u4 a = st4[0], b = st4[1];                       // (every lane the same 32 bytes: broadcast reads)
        bool ok = a.y == tag && a.w == tag && b.y == tag && b.w == tag;
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE >= 1      // development: nobody waits for the statistics (results invalid)
        if (false) {
#else
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(ok ? 1 : 0) == 0, 0)) {
#endif
#if defined(HSS_TQ_BLKPROBE)
            ++pb_miss;
#endif
            const unsigned t0 = static_cast<unsigned>(wall_clock64());
            const gu64* f = mail + static_cast<size_t>(ko & smask) * static_cast<size_t>(nwords) + G * kTqGroupWords;
            for (unsigned tries = 0;; ++tries) {
                if (lane_r < 4) {
                    const unsigned long long w = __hip_atomic_load(f + lane_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    reinterpret_cast<unsigned long long*>(stage)[lane_r] = w;
                }
                wave_sync();
                a = st4[0]; b = st4[1];
                ok = a.y == tag && a.w == tag && b.y == tag && b.w == tag;
                if (__builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0) break;
                if ((tries & 3u) == 3u && expired(t0)) { gave_up(); leave(); }
                if (is_dead()) leave();
                __builtin_amdgcn_s_sleep(4);
            }
#if defined(HSS_TQ_BLKPROBE)
            pb_blocked += static_cast<unsigned>(wall_clock64()) - t0;
#endif
        }
        if (lane_r < 3) {
            const unsigned x = lane_r == 2 ? b.x : a.x, y = lane_r == 2 ? b.z : a.z, z = lane_r == 0 ? a.x : b.x, w = lane_r == 0 ? a.z : b.z;
            wstat[lane_r] = make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
        }
        wave_sync();
    };

    // z-score of a held group from registers and its 3 streaming stores per lane (fsst_team16.hpp emit_held)
    auto emit_held = [&](auto SL, int ko_h, int g_h) {
        constexpr int sl = decltype(SL)::value;
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned cofs = cls_lds[lane_r];
        const char* tb = reinterpret_cast<const char*>(wstat);
        char* obase = reinterpret_cast<char*>(P()->out) + static_cast<unsigned long long>(static_cast<unsigned>(team + ko_h * nteams)) * sig_bytes +
                      static_cast<unsigned>(g_h * (16 * 2 * K * 4));                                                          // (wave-uniform)
        const unsigned voff = static_cast<unsigned>(lane_r) * 16u;
        const int nvalid = min(16, ncols - g_h * 16);
        auto put = [&](auto I) {
            constexpr int i = decltype(I)::value;
            const float4 tt = *reinterpret_cast<const float4*>(tb + ((cofs >> (8 * i)) & 0xffu));
            const f2 lo = held_zscore<6 * sl + 2 * i>(f2{tt.x, tt.y});
            const f2 hi = held_zscore<6 * sl + 2 * i + 1>(f2{tt.z, tt.w});
            __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(obase + (voff + 1024u * static_cast<unsigned>(i))));
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

    // ---- the held groups (fsst_team16.hpp): images in the fixed registers, a strict first-in first-out of DEPTH slots; with two planes the
    //      previous step's image waits in the other plane and moves into the registers at the end of the next step
    int nheld = 0;
    int ko_hs[DEPTH], g_hs[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ko_hs[d] = 0; g_hs[d] = 0; }
    auto uni = [&](int v) -> int { return __builtin_amdgcn_readfirstlane(v); };
    bool c_valid = false;                                // a group is landed: (ko, g), its tile in xrec
    int ko = 0, g = 0;
    CanonTile tile{};
    auto land = [&]() {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        asm volatile("" ::: "memory");                   // (the samples are in xraw: the step's explicit wait stands in front of this)
        const float sreg[3] = {xraw[lane_t], xraw[lane_t + 64], xraw[lane_t + 128]};
        tile = canon_land<true>(sreg, xrec, P()->r2scale_s, P()->inv_c, lane_t, ((g_d + cg0) & ~3) * 16, n);
        // (a tile's scales are the same in every lane: in scalar registers across the transform, not in four vector ones)
        tile.R2s = __int_as_float(uni(__float_as_int(tile.R2s))); tile.inv = __int_as_float(uni(__float_as_int(tile.inv)));
        tile.r2s = __int_as_float(uni(__float_as_int(tile.r2s))); tile.mean_s = __int_as_float(uni(__float_as_int(tile.mean_s)));
        ko = ko_d; g = g_d; c_valid = true; d_valid = false;
    };
    int slot = 0;
    int cur = 0;                                         // the plane this step transforms into
    bool p_valid = false;                                // the OTHER plane holds the previous step's image: group (p_ko, p_g), scale p_inv
    int p_ko = 0, p_g = 0;
    float p_inv = 0.0f;

    // the first two tickets: the second one's samples are on their way while the first group is transformed
    tk_req();
    HSS_RARE_VMEM_DONE();
    tk_take();
    if (d_valid) {
        tk_req();
        HSS_RARE_VMEM_DONE();
        land();
        tk_take();
    }
    int fin_ko = -1;                                     // >= 0: this wave has published that signal's last group and owes the team its statistics
    for (;;) {
        if (__builtin_expect(fin_ko >= 0, 0)) {
#if !(defined(HSS_T16_ABLATE) && (HSS_T16_ABLATE == 1 || HSS_T16_ABLATE == 2))
            finish_signal(fin_ko, reinterpret_cast<float*>(own_first + cur * (16 * C::LD)));
#endif
            fin_ko = -1; fin_hold = false;
        }
        if (!c_valid) {                                  // slow path: nothing landed
            if (!d_valid) {
                if (tk_done) break;                      // the team's tickets are handed out (the loop's only exit)
                if (!tk_pending) tk_req();
                HSS_RARE_VMEM_DONE();
                tk_take();
                if (!d_valid) continue;
            }
            HSS_RARE_VMEM_DONE();
            land();
        }
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const unsigned b = static_cast<unsigned>(team + ko * nteams);
        const int tg = P()->col0 + g * 16;
        f2* own_base = own_first + cur * (16 * C::LD);
        // the group that leaves at the end of this step: its signal's statistics are asked for from the middle of the transform -- and the
        // team's next ticket with them
        int pf_ko = -1;
        if (nheld == DEPTH && (PLANES == 1 || p_valid))
            static_for<DEPTH>([&](auto S) { if (slot == decltype(S)::value) pf_ko = ko_hs[decltype(S)::value]; });
        const bool want_tk = !tk_pending && !tk_done && !fin_hold;      // (a drawn group's samples are in flight: it lands behind this transform, the new ticket's are asked for then)
        auto mid = [&]() {
            if (pf_ko >= 0) stats_prefetch(pf_ko);
            if (want_tk) { if (lane == 0) v_tk = __hip_atomic_fetch_add(tk_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        };
        if (want_tk) tk_pending = true;                  // (said here, not inside the transform: the loop's state stays visibly wave-uniform)
#if defined(HSS_T16_PF_AT) && HSS_T16_PF_AT == 0
        mid();
#endif
        canon_group<KLO, KC, HSS_T16_TAPB, true>(xrec + ((g + cg0) & 3) * 16, atab, own_base, flag, tq, P()->wtab, P()->twtab, tile, tiny, lane_o,
                                           [&]() -> const float* { return P()->x + static_cast<unsigned long long>(b) * static_cast<unsigned>(P()->xstride); }, n, tg, P()->atab + kCanonAtabFloats,
                                           nullptr, mid);
        const float inv_cur = tile.inv;
        const int ko_cur = ko, g_cur = g;
        c_valid = false;
        // the step's ONE wait for memory: the next tile's samples, the ticket, the statistics words, the previous group's stores
        HSS_RARE_VMEM_DONE();
        if (d_valid) land();
        // ---- the group's statistics partial -> the team's mailbox: six tagged words (lanes 0 / 16 / 32 / 48 hold S1re / S2re / S1im / S2im,
        //      every lane the pivot pair)
        {
            const int nvalid = min(16, cend - tg);
            f2 piv;
            const float w = canon_stats<KLO, KC>(own_base, nvalid, inv_cur, lane_o, piv);
            const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko_cur) & 0xffffu);
            char* e = reinterpret_cast<char*>(P()->mail) + (static_cast<unsigned long long>(static_cast<unsigned>(team * P()->slots + (ko_cur & smask))) * static_cast<unsigned>(nwords) +
                                                            static_cast<unsigned>(g_cur * kTqGroupWords)) * 8ull;                  // (wave-uniform)
            const bool rowlead = (lane_o & 15) == 0;
            const float val = rowlead ? w : (lane_o == 1 ? piv.x : piv.y);
            const unsigned idx = rowlead ? static_cast<unsigned>(lane_o) >> 4 : 3u + static_cast<unsigned>(lane_o);
            if (rowlead || lane_o == 1 || lane_o == 2)
                __hip_atomic_store((gu64*)reinterpret_cast<unsigned long long*>(e + idx * 8u), (static_cast<unsigned long long>(tag) << 32) | __float_as_uint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(HSS_TQ_BLKPROBE)
            ++pb_groups;
#endif
            // (the signal's last group: this wave finishes the signal for the team -- at the top of the loop, behind this step's leaving group,
            //  where nothing of a transform is live: inlined here the finisher's fifty registers made the allocator spill around it on the hot path)
            if (__builtin_expect(g_cur == G - 1, 0)) fin_ko = ko_cur;
        }
        if (tk_pending) tk_take();                       // (its answer came back with the step's wait)
        // ---- this step's slot: the group that sits there (the oldest the wave holds) leaves, an image moves in (fsst_team16.hpp)
        auto move_in = [&](const f2* src_plane, float inv_src, int ko_src, int g_src) {
            int ko_o = 0, g_o = 0;
            static_for<DEPTH>([&](auto S) { if (slot == decltype(S)::value) { ko_o = ko_hs[decltype(S)::value]; g_o = g_hs[decltype(S)::value]; } });
            const bool full = nheld == DEPTH;
            int lane_r = lane;
            asm volatile("" : "+v"(lane_r));
            const unsigned cofs = cls_lds[lane_r];
            const unsigned pk0 = ppk_lds[lane_r], pk1 = ppk_lds[64 + lane_r], pk2 = ppk_lds[128 + lane_r];
            const unsigned dd = __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (every lane: a broadcast read)
            if (__builtin_amdgcn_readfirstlane(static_cast<int>(dd)) != 0) leave();
            if (full) stats_take(ko_o); else ++nheld;
            unsigned obase = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(reinterpret_cast<size_t>((lds_float*)reinterpret_cast<const float*>(src_plane)))));
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
                        const char* tb = reinterpret_cast<const char*>(wstat);
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
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE == 2      // development: the arithmetic without the stores
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
    // ---- the tickets are handed out: the image that still sits in its plane moves in (the oldest held group leaves for it) ...
    if constexpr (PLANES == 2) {
        if (p_valid) {
            int ko_o = 0;
            static_for<DEPTH>([&](auto S) { if (slot == decltype(S)::value) ko_o = ko_hs[decltype(S)::value]; });
            const f2* src_plane = own_first + (cur ^ 1) * (16 * C::LD);
            const bool full = nheld == DEPTH;
            if (full) stats_take(ko_o); else ++nheld;
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
    // ... and the held groups leave, oldest first
    for (int i = 0; i < nheld; ++i) {
        const int so = (slot + DEPTH - nheld + i) % DEPTH;
        static_for<DEPTH>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            if (so == sl) { stats_take(ko_hs[sl]); emit_held(S, ko_hs[sl], g_hs[sl]); }
        });
    }
#if defined(HSS_TQ_BLKPROBE)
    if (lane == 0 && virt < 256) {
        unsigned* e = g_tq_blk + (virt * 16 + wv) * 8;
        e[0] = pb_miss; e[1] = 0u; e[2] = pb_blocked; e[3] = pb_fin; e[4] = pb_nfin; e[5] = pb_groups;
    }
#endif
}

}  // namespace hssfsst
