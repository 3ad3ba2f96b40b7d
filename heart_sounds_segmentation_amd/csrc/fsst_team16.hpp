// fsst_team16.hpp -- the canonical-band transform with the z-score of FSST._stack_real_imag
// (/root/reference/hss/transforms/synchrosqueeze.py:78-85) applied IN REGISTERS at FOUR waves per SIMD: every feature is
// written to HBM exactly once, already normalised (algorithmic traffic: 8 000 B in + 352 000 B out per 2000-sample window).
//
// Round 3 had two single-launch z-score kernels: one CU per signal (16 waves per CU, but the un-normalised tile makes a
// round trip through HBM: 2.98x the algorithmic traffic at 85 % of the chip's streaming rate -- its own ceiling) and the team
// kernel of fsst_team128.hpp (1.04x the traffic, but a wave held a 4-group chunk in 48 registers across the next chunk's
// transform: 233 VGPRs = two waves per SIMD, the transform at 0.188 instead of 0.140 ms per 1024 windows).  This kernel
// keeps the team idea and drops what made it wide:
//   * the unit of work is ONE 16-frame group, not a 64-frame chunk.  A team of T CUs (16 T waves) shares a signal; its
//     groups are dealt round-robin to the team's CUs, whose 16 waves draw them from a ticket counter in LDS.  With
//     16 T >= the groups of a signal, all groups of a signal are in flight at once: a signal passes through the team in about
//     ONE group time, so a finished group waits for its signal's statistics for about one group time, not four;
//   * a wave therefore holds ONE group image -- three float4 per lane, 12 registers -- across exactly one further group's
//     transform (canon_group of fsst_canon128.hpp, the arithmetic of every other canonical-band kernel: bit-identical
//     features on every z-score path): 12 + the transform's ~105 fit the 128 registers of four waves per SIMD, and the LDS
//     regions are those of fsst_canon_kernel;
//   * a group's statistics partial (the six float32 numbers of the two-launch path) is published as six tagged 8-byte words
//     in the team's mailbox (relaxed agent-scope atomics: no fence, no cache write-back);
//   * the float64 part of the statistics runs ONCE per signal and CU, not once per chunk and wave: the first wave of a CU
//     that needs a signal's statistics claims it (LDS), collects the signal's partials from the mailbox, runs the very
//     instructions of signal_stats() (fsst_kernels.hpp) on them and leaves {mean, 1/std} x 2 in LDS for its 15 siblings.
//
// Progress.  A wave publishes a group before it waits for anything, and it waits only for the signal of the group it HOLDS.
// Besides the held group it has up to two tickets it has not published yet (one landed, one drawn with its samples in flight):
// these must never belong to the signal it waits for -- a first version drew ahead unconditionally and deadlocked as soon as a
// wave fell behind its siblings (held group, landed group and drawn group all of ONE signal: it waited for a statistic that
// needed its own drawn group).  Hence the rule in draw(): a ticket is drawn ahead only if every position still to be handed
// out lies in a LATER signal than the group just published; otherwise the wave draws when it has nothing landed.  Then: let a*
// be the oldest signal of a team with an unpublished group.  A drawn-but-unpublished group of a* belongs to a wave that is
// transforming or waits for an older -- complete -- signal: it gets published.  An undrawn one needs a free wave of its CU: if
// all 16 were waiting they would each hold a published group of a* that precedes it in the CU's list, 17 positions of one
// signal, but a list holds at most 16 (host-checked: cpc <= 16).  A CU cannot run more than 3 x 16 list positions ahead of its
// oldest unresolved signal (every wave is then waiting), which bounds the mailbox / LDS slots in use (host: slots >= 2 lead + 2).
// Every wait is bounded in wall-clock time; a wait that runs out gives the LAUNCH up (abort word) and the gated launches
// queued behind it compute the exec (hssfsst.hip), exactly as for fsst_team128_kernel -- see there for why (co-residency with
// other processes' kernels is not guaranteed).
#pragma once
#include "fsst_mfma128.hpp"
#include "fsst_canon128.hpp"

namespace hssfsst {

using gu64 = __attribute__((address_space(1))) unsigned long long;

constexpr int kT16Waves = 16;                // waves per block: four per SIMD, 128 VGPRs each
constexpr int kT16MailWords = 6;              // tagged 8-byte words per group in the mailbox: S1re S2re S1im S2im p_re p_im
constexpr int kT16MaxSlots = 128;            // statistics slots per CU / mailbox slots per team (signal ordinal mod slots)
constexpr int kT16CtlFloats = 16 + 64 + 192 + 2 * kT16MaxSlots + 4 * kT16MaxSlots;
                                             // [0] ticket counter [1] dead [2] identity | column classes | wide-store offsets |
                                             // ready[slots], claim[slots] | float4 statistics[slots]

#ifdef HSS_T16_DEBUG
__device__ unsigned g_t16_dbg[128];
#endif
#ifdef HSS_T16_PROBE     // development: shader-clock totals per phase over all waves (results valid, kernel slowed by the stamps)
__device__ unsigned long long g_t16_probe[16];
#define T16P(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); pr_t[k] += now_ - pr_last; pr_last = now_; } while (0)
#else
#define T16P(k) do { } while (0)
#endif

struct Team16Params {
    const float* x;       // [nsig][xstride]
    float* out;           // [nsig][ncols][2 KC]
    const float* atab;    // f16 operand table + float64 twiddles (kCanonAtabFloats floats)
    const double* wtab;   // float64 {w, dw'}[128]                } rounding-tie path
    const double* twtab;  // float64 {cos, sin}(2 pi m / 128)     }
    unsigned long long* mail;   // [teams][slots][ngroups][6] tagged words {tag << 32 | float32 bits}: S1re S2re S1im S2im p_re p_im
    unsigned* status;     // device status word (0 = ok)
    float r2scale_s;      // r2scale of the plan x (constant scale)^2
    float inv_c;          // 1 / constant scale
    int n, nsig, col0, ncols;
    long long xstride;
    int team;             // CUs per team (power of two)
    int cpc_shift;        // log2 of the list positions per CU and signal (ceil(ngroups / team) rounded up to a power of two, <= 16)
    int slots;            // mailbox / statistics slots (power of two <= kT16MaxSlots)
    unsigned seq;         // launch sequence number of the plan (upper half of the mailbox tags)
    unsigned spin_ticks;  // bound of a wait in 100 MHz ticks
    unsigned* arrive;     // arrival counter of the plan (monotone over launches)
    unsigned arrive_base; // its value before this launch: block identity = arrival number - arrive_base
    unsigned* abort_word; // a wait that ran out of time stores `launch` here; every wave then leaves the kernel
    unsigned* fallbacks;  // pinned host word: the same store, for the host's eyes
    unsigned launch;      // identity of this launch (never 0)
};

template <int KLO, int KC>
__global__ __launch_bounds__(64 * kT16Waves, HSS_MW128) void fsst_team16_kernel(Team16Params p)
{
    using C = CanonCfg<KLO, KC>;
    constexpr int WPB = kT16Waves, K = KC, ATAB = kCanonAtabFloats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = p.n;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* atab = smem;
    int* next_q = reinterpret_cast<int*>(smem + ATAB);
    unsigned* dead = reinterpret_cast<unsigned*>(smem + ATAB) + 1;
    unsigned* cls_lds = reinterpret_cast<unsigned*>(smem + ATAB + 16);       // [64]
    unsigned* ppk_lds = reinterpret_cast<unsigned*>(smem + ATAB + 80);       // [3][64]
    unsigned* ready = reinterpret_cast<unsigned*>(smem + ATAB + 272);        // [slots] epoch (signal ordinal + 1) of the statistics in fin[]
    unsigned* claim = ready + kT16MaxSlots;                                  // [slots] epoch some wave of this CU is resolving / has resolved
    float4* fin = reinterpret_cast<float4*>(smem + ATAB + 272 + 2 * kT16MaxSlots);
    float* wbase = smem + ATAB + kT16CtlFloats + wv * C::wave_floats();
    u2* xrec = reinterpret_cast<u2*>(wbase);
    f2* own_base = reinterpret_cast<f2*>(wbase + 2 * kCanonRecs);
    f2* disp_base = own_base + 16 * C::LD;
    int* flag = reinterpret_cast<int*>(disp_base + 16 * C::LDF);
    int* tq = flag + 4;

    for (int i = threadIdx.x; i < ATAB; i += 64 * WPB) atab[i] = p.atab[i];
    for (int i = lane; i < 16 * C::LDF; i += 64) disp_base[i] = f2{0.0f, 0.0f};
    if (lane < 4) flag[lane] = 0;
    if (lane < kCanonTieWords) tq[lane] = 0;
    if (threadIdx.x < 16 && threadIdx.x != 2) next_q[threadIdx.x] = 0;       // ([2]: the identity, written below)
    for (int i = threadIdx.x; i < 2 * kT16MaxSlots; i += 64 * WPB) ready[i] = 0u;        // ready[] and claim[]
    // Block identity = ARRIVAL number (fsst_team128.hpp: the blocks that are running hold identities 0 .. R - 1, so every team
    // below R / T is complete whatever share of the chip this launch was given).
    if (threadIdx.x == 64)
        next_q[2] = static_cast<int>(__hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.arrive_base);
    if (wv == 0) {
        unsigned cls = 0u;                               // bit 2i / 2i+1: the first / second pair of float4 i is imaginary
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned c = (4u * static_cast<unsigned>(lane + 64 * i)) % static_cast<unsigned>(2 * K);
            cls |= (c >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i);
            cls |= (c + 2 >= static_cast<unsigned>(K) ? 1u : 0u) << (2 * i + 1);
            ppk_lds[i * 64 + lane] = canon_store_offsets<KLO, KC>(lane + 64 * i);
        }
        cls_lds[lane] = cls;
    }
    __syncthreads();

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
    auto expired = [&](unsigned since) -> bool {
        return static_cast<unsigned>(wall_clock64()) - since > P()->spin_ticks || is_dead() || aborted();
    };

    // ---- team geometry (wave-uniform): T consecutive identities form a team
    const int virt = __builtin_amdgcn_readfirstlane(next_q[2]);
    // an identity outside the grid = arrivals of two launches of one plan interleaved (a plan is single-stream: hssfsst.h):
    // give the launch up instead of indexing outside the mailboxes (the gated fallback computes the exec)
    if (static_cast<unsigned>(virt) >= gridDim.x) { gave_up(); return; }
    if (aborted()) return;                               // (a block that starts after the launch was given up)
    const int T = p.team, cpcs = p.cpc_shift, cpc = 1 << cpcs;
    const int member = virt & (T - 1), team = virt / T;
    const int nteams = static_cast<int>(gridDim.x) / T;
    const int nk = (p.nsig > team) ? (p.nsig - team + nteams - 1) / nteams : 0;      // signals of this team
    const int nwork = nk << cpcs;
    const int ncols = p.ncols, cend = p.col0 + p.ncols;
    const int G = (ncols + 15) >> 4;
    const int cg0 = p.col0 >> 4;                         // (the host sends only column ranges that start on a group boundary)
    const int smask = p.slots - 1;
    gu64* mail = (gu64*)(p.mail) + static_cast<size_t>(team) * static_cast<size_t>(p.slots) * G * kT16MailWords;

    f2 tiny = {1.0e-37f, 0.0f};
    asm volatile("" : "+s"(tiny));
#ifdef HSS_T16_PROBE
    unsigned long long pr_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pr_last = __builtin_readcyclecounter();
    const unsigned long long pr_begin = pr_last;
#endif

    // ---- draw: the next group of this CU's list and its tile's samples on their way into registers
    float sreg[3];
    bool d_valid = false;
    int ko_d = 0, g_d = 0;
    // after_ko >= 0: draw only if every position still to be handed out belongs to a signal AFTER after_ko (see "Progress":
    // a ticket the wave holds unpublished while it waits must not belong to the signal it waits for)
    auto draw = [&](int after_ko) {
        int qi = 0x7fffffff;
        if (lane == 0) {
            const bool go = after_ko < 0 || (__hip_atomic_load(next_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> cpcs) > after_ko;
            if (go) qi = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        qi = __builtin_amdgcn_readfirstlane(qi);
        d_valid = false;
        while (qi < nwork) {
            ko_d = qi >> cpcs;
            g_d = ((member + ko_d) & (T - 1)) + T * (qi & (cpc - 1));
            if (g_d < G) { d_valid = true; break; }
            if (lane == 0) qi = __hip_atomic_fetch_add(next_q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            qi = __builtin_amdgcn_readfirstlane(qi);
        }
        if (d_valid) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const float* xsig = P()->x + (static_cast<long long>(team) + static_cast<long long>(ko_d) * nteams) * P()->xstride;
            canon_fetch(xsig, n, ((g_d + cg0) & ~3) * 16, lane_o, sreg);
        }
    };

    // ---- the held group: image in registers (feature units, un-normalised), waiting for its signal's statistics
    bool h_valid = false;
    int ko_h = 0, g_h = 0;
    int dbg_ko_cur = 0, dbg_g_cur = 0; (void)dbg_ko_cur; (void)dbg_g_cur;
    f4 held[3];

    // statistics of signal ordinal ko (of this team) -> {mean_re, 1/std_re, mean_im, 1/std_im}; false = the launch was given up.
    // The first wave of the CU that asks claims the signal and resolves it for its siblings: it copies the signal's partials
    // from the mailbox -- lane-linear 16-byte loads (two tagged words: each word carries its own tag, so a load that is not
    // atomic as a whole is still validated word by word), every word fetched once; a lane whose words have not all arrived
    // asks again for those alone -- into LDS as the two-launch path's partials [G][6] and runs signal_stats() on them: the very
    // instructions of fsst_stats_kernel on the very numbers.  The copy lives in the wave's displaced plane + flags + bitmap
    // (contiguous, 3 088 B >= 128 groups x 24 B; all zero between groups, and zero again when the wave is done).
    auto signal_statistics = [&](int ko, float4& st) -> bool {
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE >= 1      // development: nobody waits, nobody resolves (results invalid)
        st = make_float4(0.0f, 1.0f, 0.0f, 1.0f);
        return true;
#endif
        const int sl = ko & smask;
        const unsigned epoch = static_cast<unsigned>(ko) + 1u;
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned t0 = static_cast<unsigned>(wall_clock64());
        for (unsigned spins = 0;; ++spins) {
            unsigned have = 0u;
            if (lane == 0) have = __hip_atomic_load(ready + sl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(have) == epoch) break;
            unsigned mine = 0u;
            if (spins == 0u && lane == 0) mine = __hip_atomic_fetch_max(claim + sl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch ? 1u : 0u;
            if (__builtin_amdgcn_readfirstlane(mine) != 0u) {
                const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko) & 0xffffu);
                const gu64* slot = mail + static_cast<size_t>(sl) * G * kT16MailWords;
                float* stage = reinterpret_cast<float*>(disp_base);
                const int nwords = G * kT16MailWords;                                  // (even: 16-byte pairs)
                constexpr int RND = (kFusedMaxGroups * kT16MailWords + 127) / 128;         // pairs per lane
                unsigned need = 0u;                                                       // bit r: pair lane + 64 r still missing
#pragma unroll
                for (int r = 0; r < RND; ++r) need |= (2 * (lane_r + 64 * r) < nwords ? 1u : 0u) << r;
                for (unsigned polls = 0;; ++polls) {
                    using ull2 = unsigned long long __attribute__((ext_vector_type(2)));
                    ull2 w[RND];
                    const int last = (nwords >> 1) - 1;
#pragma unroll
                    for (int r = 0; r < RND; ++r) {
                        const gu64* q = slot + 2 * min(lane_r + 64 * r, last);
                        w[r].x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[r].y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int r = 0; r < RND; ++r)
                        if ((need >> r) & 1u) {
                            if (static_cast<unsigned>(w[r].x >> 32) == tag && static_cast<unsigned>(w[r].y >> 32) == tag) {
                                reinterpret_cast<f2*>(stage)[lane_r + 64 * r] = f2{__uint_as_float(static_cast<unsigned>(w[r].x)), __uint_as_float(static_cast<unsigned>(w[r].y))};
                                need &= ~(1u << r);
                            }
                        }
                    if (__builtin_amdgcn_ballot_w64(need != 0u) == 0ull) break;
                    if ((polls & 7u) == 7u && expired(t0)) {
#ifdef HSS_T16_DEBUG
                        const unsigned miss = __builtin_popcountll(__builtin_amdgcn_ballot_w64(need != 0u));
                        if (!aborted() && need != 0u) {
                            const unsigned at = atomicAdd(&g_t16_dbg[0], 1u);
                            if (at < 15u) {
                                int r0 = __builtin_ctz(need);
                                g_t16_dbg[4 * at + 4] = (static_cast<unsigned>(team) << 24) | (static_cast<unsigned>(member) << 16) | static_cast<unsigned>(ko);
                                g_t16_dbg[4 * at + 5] = static_cast<unsigned>(lane_r + 64 * r0) | (need << 16);
                                g_t16_dbg[4 * at + 6] = static_cast<unsigned>(w[r0].x >> 32);
                                g_t16_dbg[4 * at + 7] = tag;
                            }
                        }
                        if (lane == 0 && !aborted()) __hip_atomic_store((gu32*)(P()->status), (1u << 28) | (miss << 20) | ((unsigned)member << 16) | (static_cast<unsigned>(ko) & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                        gave_up(); return false;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
                wave_sync();
                T16P(5);
                int ncols_o = ncols, K_o = K, G_o = G;
                asm volatile("" : "+s"(ncols_o), "+s"(K_o), "+s"(G_o));
                const float4 r = signal_stats<kT16MailWords>(stage, G_o, 16, ncols_o, K_o, lane_r);
                wave_sync();
                for (int i = lane_r; i < (16 * C::LDF * 2 + 4 + kCanonTieWords) / 2; i += 64) reinterpret_cast<f2*>(stage)[i] = f2{0.0f, 0.0f};
                if (lane == 0) {
                    fin[sl] = r;
                    __hip_atomic_store(ready + sl, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                wave_sync();
                T16P(6);
#ifdef HSS_T16_PROBE
                pr_t[9] += 1;
#endif
                break;
            }
            if ((spins & 31u) == 31u && expired(t0)) {
#ifdef HSS_T16_DEBUG
                if (lane == 0 && !aborted() && !is_dead()) __hip_atomic_store((gu32*)(P()->status), (2u << 28) | ((unsigned)member << 16) | (static_cast<unsigned>(ko) & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
                gave_up(); return false;
            }
            if (is_dead()) {
#ifdef HSS_T16_DEBUG
                if (lane == 0) {
                    const unsigned at = atomicAdd(&g_t16_dbg[1], 1u);
                    if (at < 20u) { g_t16_dbg[64 + 2 * at] = (static_cast<unsigned>(team) << 24) | (static_cast<unsigned>(member) << 16) | (static_cast<unsigned>(wv) << 8) | static_cast<unsigned>(ko);
                                    g_t16_dbg[65 + 2 * at] = (static_cast<unsigned>(g_h) << 16) | (static_cast<unsigned>(dbg_ko_cur) << 8) | static_cast<unsigned>(dbg_g_cur); }
                }
#endif
                return false;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        st = fin[sl];
        T16P(4);
        return true;
    };

    // z-score of the held group from registers -- (v - mean) * (1 / std), two roundings, exactly as fsst_normalize_kernel -- and
    // its 12 streaming stores
    auto emit_held = [&](const float4& st) {
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const unsigned cls = cls_lds[lane_r];
        auto zs = [](f2 v, f2 m) -> f2 {
            f2 d, e;
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(v), "v"(m));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(e) : "v"(d), "v"(m));
            return e;
        };
        const long long b = static_cast<long long>(team) + static_cast<long long>(ko_h) * nteams;
        float4* d4 = reinterpret_cast<float4*>(P()->out + (b * static_cast<long long>(ncols) + g_h * 16) * (2 * K)) + lane_r;
        const int nvalid = min(16, ncols - g_h * 16);
        auto put = [&](int i) {
            const bool im0 = (cls >> (2 * i)) & 1u, im1 = (cls >> (2 * i + 1)) & 1u;
            const f2 m0 = f2{im0 ? st.z : st.x, im0 ? st.w : st.y}, m1 = f2{im1 ? st.z : st.x, im1 ? st.w : st.y};
            const f2 lo = zs(f2{held[i].x, held[i].y}, m0);
            const f2 hi = zs(f2{held[i].z, held[i].w}, m1);
#if defined(HSS_T16_ABLATE) && HSS_T16_ABLATE >= 2      // development: the arithmetic without the stores
            { f2 l2 = lo, h2 = hi; asm volatile("" :: "v"(l2), "v"(h2)); }
#else
            __builtin_nontemporal_store(f4{lo.x, lo.y, hi.x, hi.y}, reinterpret_cast<f4*>(d4 + 64 * i));
#endif
        };
        if (__builtin_expect(nvalid == 16, 1)) {
            static_for<3>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (64 * (i + 1) <= 8 * K) put(i);
                else if constexpr (64 * i < 8 * K) { if (lane_r + 64 * i < 8 * K) put(i); }
            });
        } else {
            asm volatile("");
            const int lim = nvalid * (K >> 1);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (lane_r + 64 * i < lim) put(i);
        }
    };

    // One loop body, in this order -- a wave's memory operations retire in order, so the ONE wait for loaded data per group
    // (the next tile's samples) sits where everything else in flight -- the previous group's stores -- is a whole transform old:
    //   1. transform the landed group; 2. land the drawn group's tile (its records are free: every group stages its own tile),
    //   3. statistics partial of the transformed group -> mailbox; draw a further group and request its samples -- IF that is
    //      safe: the wave is about to wait for the statistics of signals up to this group's, so a ticket it holds unpublished
    //      across those waits must belong to a later signal (the ticket counter is looked at first; positions only grow).
    //      Otherwise the wave draws when it has nothing landed -- holding only published groups (rare: the slow path at the top);
    //   4. the HELD group leaves (statistics of its signal: LDS, or this wave resolves them; z-score from registers, 3 stores);
    //   5. the transformed group's image: own plane -> the held registers.
    bool c_valid = false;                                // a group is landed: (ko, g), its tile in xrec
    int ko = 0, g = 0;
    CanonTile tile{};
    auto land = [&]() {
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        tile = canon_land(sreg, xrec, P()->r2scale_s, P()->inv_c, lane_t);
        ko = ko_d; g = g_d; c_valid = true; d_valid = false;
    };
    draw(-1);
    if (d_valid) { land(); draw(-1); }                   // (the second ticket is transformed and published before the wave's first wait)
    for (;;) {
        if (is_dead()) return;                           // the launch was given up (a wave that waits also looks at the abort word)
        if (!c_valid) {                                  // slow path: nothing landed -- the wave holds nothing unpublished
            if (!d_valid) draw(-1);
            if (!d_valid) break;
            land();
        }
        T16P(8);
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const long long b = static_cast<long long>(team) + static_cast<long long>(ko) * nteams;
        const int tg = P()->col0 + g * 16;
        canon_group<KLO, KC, 2>(xrec + ((g + cg0) & 3) * 16, atab, own_base, disp_base, flag, tq, P()->wtab, P()->twtab, tile, tiny, lane_o,
                                P()->x + b * P()->xstride, n, tg);
        T16P(0);
        const float inv_cur = tile.inv;
        const int ko_cur = ko, g_cur = g;
        dbg_ko_cur = ko_cur; dbg_g_cur = g_cur;
        c_valid = false;
        if (d_valid) land();
        T16P(1);
        // ---- statistics partial -> the team's mailbox: rows 0..3 of the wave hold S1re / S2re / S1im / S2im, every lane the pivot
        {
            const int nvalid = min(16, cend - tg);
            f2 piv;
            const float w = canon_stats<KLO, KC>(own_base, nvalid, inv_cur, lane_o, piv);
            const unsigned tag = (P()->seq << 16) | (static_cast<unsigned>(ko_cur) & 0xffffu);
            gu64* e = mail + (static_cast<size_t>(ko_cur & smask) * G + g_cur) * kT16MailWords;
            const bool odd = lane_o & 1;
            const int word = odd ? 4 + (lane_o >> 4) : (lane_o >> 4);
            const float val = odd ? ((lane_o >> 4) ? piv.y : piv.x) : w;
            if ((lane_o & 15) == 0 || ((lane_o & 15) == 1 && lane_o < 32))
                __hip_atomic_store(e + word, (static_cast<unsigned long long>(tag) << 32) | __float_as_uint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        draw(ko_cur);
        T16P(2);
        // ---- the held group leaves
        if (h_valid) {
            float4 st;
            if (!signal_statistics(ko_h, st)) return;
            emit_held(st);
            T16P(3);
        }
        // ---- the new group's image: own plane -> registers (feature units)
        canon_image<KLO, KC>(own_base, ppk_lds, inv_cur, lane_o, held);
        wave_sync();
        h_valid = true; ko_h = ko_cur; g_h = g_cur;
        T16P(7);
#ifdef HSS_T16_PROBE
        pr_t[10] += 1;
#endif
    }
    if (h_valid) {
        float4 st;
        if (!signal_statistics(ko_h, st)) return;
        emit_held(st);
    }
#ifdef HSS_T16_PROBE
    // [0] transform [1] land [2] stats + publish + draw [3] emit [4] wait (waiter, incl. resolver total) [5] resolver: poll [6] resolver: compute [7] image [8] loop top
    if (lane == 0) {
        pr_t[11] = __builtin_readcyclecounter() - pr_begin;
        for (int k = 0; k < 12; ++k) atomicAdd(g_t16_probe + k, pr_t[k]);
        atomicAdd(g_t16_probe + 12, 1ull);
    }
#endif
}

}  // namespace hssfsst
