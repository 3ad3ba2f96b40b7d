// hssfsst.hip -- host side of libhssfsst.so: the C ABI declared in include/hssfsst.h.
// Plan creation does, once and in fp64, everything of the reference's `ssq.fsst(x, fs, window)`
// (call site /root/reference/hss/transforms/synchrosqueeze.py:48) that depends only on
// (fs, window); exec launches the gfx950 kernels of fsst_kernels.hpp.
// There is no CPU compute path here by design (the CPU restatement is oracle/, test-only).
#include "../../include/hssfsst.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <mutex>
#include <dlfcn.h>
#include <cctype>
#include <string>
#include <atomic>
#include <new>
#include <thread>
#include <vector>

#include "fsst_kernels.hpp"
#include "fsst_mfma128.hpp"
#include "fsst_canon128.hpp"
#include "fsst_team16.hpp"
#ifndef HSS_T16_WPB
#define HSS_T16_WPB 16
#define HSS_T16_DEPTH 2
#endif
#include "fsst_dft.hpp"
#include "fsst_gather.hpp"
#include "fourier_resample.hpp"
#include <cstdlib>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(HSSFSST_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// Development / A-B switches, read ONCE per process (first use) from the environment: HSSFSST_DEBUG="key[=value],key,..." with the
// keys below, or -- the older spelling the tests and tools use -- one variable per key, HSSFSST_<KEY in capitals>[=value].
// Defaults are the measured best; no switch changes a result (every path gives the same bits, which is what most of them
// exist to show).  Nothing else in the library reads the environment.
struct DebugSwitches {
    bool no_fused = false;        // z-score always as a second kernel
    bool no_canon = false;        // the canonical band on the general kernels (fsst_mfma128.hpp)
    bool no_team = false;         // never the team kernel
    bool team_only = false;       // the team kernel or two launches, never one CU per signal
    bool team_force_fallback = false;   // every team launch finds itself given up (tests of the gated fallback)
    bool force_dft = false;       // every window length on the any-length kernel
    bool force_generic = false;   // every radix length on the generic VALU kernel
    bool no_mfma256 = false;      // nwin 256 / 512 on the generic kernel
    bool split_stats = false;     // a separate statistics launch on the two-launch path
    bool no_stream_fuse = false;  // a streaming step as copy + transform + merge-and-normalise launches
    bool no_pair = false;         // nwin 256 / 512: one wave per wave region (no wave pairs)
    int team = 0;                 // CUs per team (0: chosen by the library)
    unsigned team_spin_us = 500;  // bound of a wait inside the team kernel
    int oneplane_kb = 40;         // generic kernel: one shared LDS plane above this many KB
    int chunks = 0;               // STACK: k-chunk two-stream pipeline (0 / 1: off)
    int zgrid = 0, zslices = 0;   // z-score sweep geometry (0: chosen by the library)
};
const DebugSwitches& debug_switches()
{
    static const DebugSwitches sw = [] {
        DebugSwitches d;
        auto set = [&](const std::string& key, const char* val) {
            const int iv = val ? std::atoi(val) : 0;
            const bool on = !val || val[0] == '\0' || iv != 0 || val[0] == 'y' || val[0] == 't';
            if (key == "no_fused") d.no_fused = on; else if (key == "no_canon") d.no_canon = on;
            else if (key == "no_team") d.no_team = on; else if (key == "team_only") d.team_only = on;
            else if (key == "team_force_fallback") d.team_force_fallback = on; else if (key == "force_dft") d.force_dft = on;
            else if (key == "force_generic") d.force_generic = on; else if (key == "no_mfma256") d.no_mfma256 = on;
            else if (key == "split_stats") d.split_stats = on; else if (key == "no_stream_fuse") d.no_stream_fuse = on; else if (key == "no_pair") d.no_pair = on; else if (key == "team") d.team = iv;
            else if (key == "team_spin_us") d.team_spin_us = static_cast<unsigned>(iv > 0 ? iv : 500);
            else if (key == "oneplane_kb") d.oneplane_kb = iv; else if (key == "chunks") d.chunks = iv;
            else if (key == "zgrid") d.zgrid = iv; else if (key == "zslices") d.zslices = iv;
        };
        static const char* const keys[] = {"no_fused", "no_canon", "no_team", "team_only", "team_force_fallback", "force_dft", "force_generic",
                                           "no_mfma256", "split_stats", "no_stream_fuse", "no_pair", "team", "team_spin_us", "oneplane_kb", "chunks", "zgrid", "zslices"};
        for (const char* k : keys) {                       // HSSFSST_<KEY>
            std::string name = "HSSFSST_";
            for (const char* c = k; *c; ++c) name += static_cast<char>(std::toupper(static_cast<unsigned char>(*c)));
            if (const char* v = std::getenv(name.c_str())) set(k, (name == "HSSFSST_FORCE_GENERIC" && v[0] != '1') ? "0" : v);
        }
        if (const char* all = std::getenv("HSSFSST_DEBUG")) {   // HSSFSST_DEBUG="no_fused,team=16"
            std::string item;
            for (const char* c = all;; ++c) {
                if (*c == ',' || *c == '\0') {
                    if (!item.empty()) {
                        const size_t eq = item.find('=');
                        if (eq == std::string::npos) set(item, nullptr);
                        else set(item.substr(0, eq), item.c_str() + eq + 1);
                    }
                    item.clear();
                    if (*c == '\0') break;
                } else if (*c != ' ') item += *c;
            }
        }
        return d;
    }();
    return sw;
}

// Makes `device` current for the scope and restores the caller's device on every exit path (a process
// that drives several GPUs must not find its current HIP device changed by a library call).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVICE_SCOPE(dev)                                                                      \
    DeviceGuard device_guard_(dev);                                                            \
    if (device_guard_.err != hipSuccess)                                                       \
        return fail(HSSFSST_EHIP, "selecting device %d failed: %s", (dev), hipGetErrorString(device_guard_.err))

// Knot slopes of the not-a-knot cubic spline through (1..n, w): the derivative window of
// ssq.fsst's instantaneous-frequency estimator before its fs/(2*pi) scaling (MATLAB fsst.m, local
// function dtwin).  Tridiagonal system (unit spacing):
//     s0 + 2 s1 = (5 d0 + d1)/2;  s_{i-1} + 4 s_i + s_{i+1} = 3 (w_{i+1} - w_{i-1});
//     2 s_{n-2} + s_{n-1} = (5 d_{n-2} + d_{n-3})/2,            d_i = w_{i+1} - w_i
// solved by Gaussian elimination with partial pivoting specialised to tridiagonal matrices
// (second super-diagonal as fill-in), O(n).
int spline_knot_slopes(const double* w, int n, double* s)
{
    if (n < 1) return HSSFSST_EINVAL;
    if (n == 1) { s[0] = 0.0; return 0; }
    if (n == 2) { s[0] = s[1] = w[1] - w[0]; return 0; }
    if (n == 3) {
        const double d0 = w[1] - w[0], d1 = w[2] - w[1];
        s[0] = d0 - 0.5 * (d1 - d0); s[1] = 0.5 * (d0 + d1); s[2] = d1 + 0.5 * (d1 - d0);
        return 0;
    }
    std::vector<double> dl(n, 0.0), d(n, 0.0), du(n, 0.0), du2(n, 0.0), b(n, 0.0);
    d[0] = 1.0; du[0] = 2.0; b[0] = (5.0 * (w[1] - w[0]) + (w[2] - w[1])) / 2.0;
    for (int i = 1; i < n - 1; ++i) {
        dl[i] = 1.0; d[i] = 4.0; du[i] = 1.0; b[i] = 3.0 * (w[i + 1] - w[i - 1]);
    }
    dl[n - 1] = 2.0; d[n - 1] = 1.0;
    b[n - 1] = (5.0 * (w[n - 1] - w[n - 2]) + (w[n - 2] - w[n - 3])) / 2.0;
    for (int i = 0; i < n - 1; ++i) {            // row i+1 has sub-diagonal dl[i+1]
        if (std::fabs(d[i]) >= std::fabs(dl[i + 1])) {
            if (d[i] == 0.0) return HSSFSST_EINVAL;
            const double f = dl[i + 1] / d[i];
            d[i + 1] -= f * du[i];
            if (i + 2 < n) du[i + 1] -= f * du2[i];
            b[i + 1] -= f * b[i];
        } else {                                 // swap rows i and i+1
            const double f = d[i] / dl[i + 1];
            const double di = dl[i + 1], dui = d[i + 1], du2i = (i + 2 < n) ? du[i + 1] : 0.0;
            const double bi = b[i + 1];
            d[i + 1] = du[i] - f * dui;
            if (i + 2 < n) du[i + 1] = du2[i] - f * du2i;
            b[i + 1] = b[i] - f * bi;
            d[i] = di; du[i] = dui; du2[i] = du2i; b[i] = bi;
        }
    }
    if (d[n - 1] == 0.0) return HSSFSST_EINVAL;
    s[n - 1] = b[n - 1] / d[n - 1];
    s[n - 2] = (b[n - 2] - du[n - 2] * s[n - 1]) / d[n - 2];
    for (int i = n - 3; i >= 0; --i) s[i] = (b[i] - du[i] * s[i + 1] - du2[i] * s[i + 2]) / d[i];
    return 0;
}

// FSST._truncate_frequencies (synchrosqueeze.py:91-111): f is a float32 tensor (:52) compared
// with the Python bounds in float32, both inclusive; f_k = k*fs/nwin, Nyquist row = fs/2.
void band_rows(int nwin, double fs, double f_lo, double f_hi, int* klo, int* K)
{
    const int nf = nwin / 2 + 1;
    const double res = fs / static_cast<double>(nwin);
    int first = -1, cnt = 0;
    for (int k = 0; k < nf; ++k) {
        double fk = res * k;
        if ((nwin % 2) == 0 && k == nwin / 2) fk = fs / 2.0;
        const float f32 = static_cast<float>(fk);
        if (f32 >= static_cast<float>(f_lo) && f32 <= static_cast<float>(f_hi)) {
            if (first < 0) first = k;
            ++cnt;
        }
    }
    *klo = first < 0 ? 0 : first;
    *K = cnt;
}

constexpr int kTile = 64;
#ifndef HSS_FPW128
#define HSS_FPW128 64
#endif
constexpr int kFpw128 = HSS_FPW128;      // frames per wave tile of the nwin = 128 kernel

}  // namespace

struct hssfsst_plan {
    int device = -1;
    int nwin = 0, R = 0, nf = 0, klo = 0, K = 0, mode = 0;
    double fs = 0.0;
    float* d_ctab = nullptr;      // generic kernel: class-folded scalar tables
    float* d_dtab = nullptr;      // any-length kernel (fsst_dft.hpp): A operand [source block][k-step][64 lanes]
    int dft = 0;                  // 1: this plan runs the any-length kernel
    float r2scale = 0.0f;         // 4 nwin max |(w + i dw') / 2|^2: error-bound scale of the rounding-tie path
    double* d_wtab = nullptr;     // float64 {w, dw' in bin units}[nwin], then {cos, sin}(2 pi m / nwin)[nwin]: rounding-tie path
    float* d_atab = nullptr;      // nwin == 128 / 256 / 512: MFMA A-operand constants [pass][taps][64 lanes][k-step]
    int rq = 0;                   // first-stage radix of the MFMA kernel, 0 = generic kernel
    float* d_atab16 = nullptr;    // canonical-band kernels (fsst_canon128.hpp): f16 split A operand [16 taps][64 lanes][8 halves]
    float canon_inv_c = 0.0f;     // ... 1 / (power-of-two scale of those constants)
    float canon_r2s = 0.0f;       // ... r2scale x scale^2
    int canon_slots = 0;          // resident blocks of fsst_canon_kernel<.., false> (0 = not queried yet)
    int nt = 16;                  // taps (per-lane FFT size) of the MFMA kernel: nwin = nt * rq
    float* d_partials = nullptr;  size_t partials_cap = 0;   // floats (kPartFloats per statistics piece)
    unsigned* d_status = nullptr;                            // fused z-score: status word (0 = ok) as the device sees it ...
    volatile unsigned* h_status = nullptr;                   // ... and the same word in pinned host memory: read without a sync
    unsigned long long* d_mail = nullptr; size_t mail_cap = 0;   // team kernel: mailboxes [teams][slots][32 blocks][8] (8-byte words)
    unsigned team_seq = 0;                                   // launch sequence number (upper half of the mailbox tags)
    unsigned* d_arrive = nullptr; unsigned arrive_total = 0; // team kernel: [0] arrival counter (and its value after the launches so far), [1] abort word, [2] blocks done
    unsigned done_total = 0;                                 // ... [2]'s value after the flagged launches so far
    bool flag_done = false; unsigned flag_launch = 0;        // exec_impl asks the next team launch to say in pinned host memory (h_fallback[2]) when its last wave is done; that launch
    unsigned team_launch = 0;                                // identity of the last team launch (never 0)
    volatile unsigned* h_fallback = nullptr; unsigned* d_fallback = nullptr;   // pinned host word: identity of the last team launch that gave up
    unsigned seen_fallback = 0; int fallbacks = 0;           // ... as last seen by the host, and how many distinct ones
    int giveups_in_a_row = 0, team_pause = 0;                // a GPU shared with many processes: after kTeamGiveUps give-ups in a row the team kernel sits out
                                                             // the plan's next kTeamPause execs that would take it (each give-up costs its 0.5 ms bound first)
    const unsigned* gate = nullptr; unsigned gate_val = 0;   // set by a team launch: the two-launch kernels that follow it in the same exec are its gated fallback
    int team16_cus = 0;                                      // CUs usable by the team kernel (fsst_team16.hpp; 0 = not queried yet, -1 = none)
    char last_kernel[112] = "";                              // the transform kernel of the last exec: instantiation, waves per block, grid (hssfsst_plan_last_kernel)
    int last_fused = 0;                                      // the last exec ran a single-launch z-score kernel
    int zpath_pref = 0;                                      // HSSFSST_ZPATH_*: preference among the z-score paths
    int last_zpath = 0;                                      // ... which one: 1 = one CU per signal, 2 = team kernel
    int core128_slots = 0;                    // resident blocks of the core kernel on this device (0 = not queried yet)
    int stream_slots = 0;                     // the same for the streaming-step kernel
    unsigned* d_stream_arrive = nullptr; int stream_arrive_cap = 0;   // streaming step: blocks delivered per channel
    double* d_stream_pieces = nullptr; long long stream_pieces_cap = 0;   // and the groups' float64 sums [channels][groups][4]
    int fused_slots = 0;                      // CUs usable by the fused kernel (0 = not queried yet, -1 = none)
    float* d_stats = nullptr;     size_t stats_cap = 0;      // floats (4 per signal)
    float* d_xstage = nullptr;    size_t xstage_cap = 0;     // floats
    float* d_ostage = nullptr;    size_t ostage_cap = 0;     // floats
    // small host-to-host execs (the unchanged dataset loop: one 2000-sample frame per call): pinned, device-mapped staging that the
    // kernels read and write in place
    float* h_xpin = nullptr; float* d_xpin = nullptr; size_t xpin_cap = 0;
    float* h_opin = nullptr; float* d_opin = nullptr; size_t opin_cap = 0;
    struct PinBuf { float* h; float* d; size_t cap; bool used; };
    std::vector<PinBuf> pin_pool;                        // hssfsst_exec_pinned: pinned, device-mapped result buffers lent to the caller
    bool defer_fallback = false;                             // this exec synchronises before it returns: no gated launches behind a team launch,
    unsigned deferred_launch = 0, deferred_first = 0;        // the host looks at the pinned give-up word afterwards and redoes the exec itself
    long long* d_starts = nullptr; size_t starts_cap = 0;    // frame-list staging (hssfsst_exec_list with host starts)
    float* d_frames = nullptr;    size_t frames_cap = 0;     // frames gathered from a list, dense [batch][n]
    int timing = 0;               // the exec being queued records kernel events
    int timing_every = 0;         // hssfsst_plan_set_timing(n): every n-th exec is timed (0: off)
    unsigned timing_seq = 0;
    bool timing_closed = false;   // the exec being queued has already recorded its closing kernel event (team path: right behind the team kernel)
    std::vector<hipEvent_t> ev;   // per timed exec: (before, after) per core launch + one closing event
    size_t ev_used = 0;           // events used since timing was enabled
    std::vector<int> ev_chunks;   // core launches of each timed exec
    hipStream_t aux = nullptr;    // side stream: z-score of chunk i overlaps the core of chunk i+1
    std::vector<hipEvent_t> sync_ev;
};

namespace {

// The dynamic-LDS limit is a property of the kernel instantiation (per device), not of a plan: raise it ONCE to
// the full 160 KiB, so that plans with different band widths sharing an instantiation cannot lower it under each
// other and the hot path makes no driver call for it.
constexpr int kMaxLdsBytes = 160 * 1024;
template <class Kern>
int allow_full_lds(Kern kern, int device, std::atomic<unsigned long long>& done)
{
    const unsigned long long bit = 1ull << (device & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes));
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

// which kernel instantiation the exec that is being queued runs (hssfsst_plan_last_kernel: the dispatch made observable)
void name_kernel(hssfsst_plan* pl, int waves, long long grid, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    const int k = vsnprintf(pl->last_kernel, sizeof(pl->last_kernel), fmt, ap);
    va_end(ap);
    if (k > 0 && static_cast<size_t>(k) < sizeof(pl->last_kernel))
        snprintf(pl->last_kernel + k, sizeof(pl->last_kernel) - static_cast<size_t>(k), " [%d waves/block, grid %lld]", waves, grid);
}

int out_floats_per_sample(const hssfsst_plan* p) { return p->mode == HSSFSST_MODE_ABS ? p->K : 2 * p->K; }

template <int R>
int launch_core(hssfsst_plan* pl, hssfsst::CoreParams cp, long long nblocks, hipStream_t st)
{
    constexpr int NWIN = 32 * R;
    constexpr int XS = ((kTile + NWIN - 1 + 3) / 4) * 4;
    size_t lds = (static_cast<size_t>(XS) + static_cast<size_t>(4 * pl->K) * (kTile + 1)) * sizeof(float);
    cp.oneplane = 0;
    // wide band: own and displaced values share one plane.  Needed above 160 KiB (nwin 512, > ~150 kept rows) and
    // already worth it above 40 KiB, where LDS is what limits the resident waves (measured, 1024 x 2000, band
    // [25,200] Hz: nwin 256 core 2.25 -> 1.44 ms, nwin 512 18.9 -> 9.4 ms; HSSFSST_ONEPLANE_KB overrides)
    const int one_thr = debug_switches().oneplane_kb;
    if (lds > static_cast<size_t>(one_thr) * 1024) {
        lds = (static_cast<size_t>(XS) + static_cast<size_t>(2 * pl->K) * (kTile + 1)) * sizeof(float);
        cp.oneplane = 1;
    }
    if (lds > 160 * 1024) return fail(HSSFSST_EUNSUPPORTED, "LDS request %zu B exceeds 160 KiB", lds);
    auto kern = hssfsst::fsst_core_kernel<R, kTile>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    name_kernel(pl, 1, nblocks, "fsst_core_kernel<%d, %d>", R, kTile);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(nblocks)), dim3(kTile), lds, st, cp);
    HIP_TRY(hipGetLastError());
    return 0;
}

// floats of LDS of a core launch: WPB waves, PAIR: two waves per wave region (fsst_mfma128.hpp "PAIR")
inline size_t core128_lds_bytes(const hssfsst_plan* pl, int rq, int nt, int wpb, bool pair)
{
    const size_t regions = pair ? wpb / 2 : wpb;
    return (hssfsst::core128_atab_floats(rq, nt) + hssfsst::kCtlFloats + regions *
            (static_cast<size_t>(hssfsst::wave_lds_floats(kFpw128, pl->klo, pl->K, rq, nt)) + (pair ? hssfsst::kPairFloats : 0))) * sizeof(float);
}

int ensure_status(hssfsst_plan* pl);
template <int NT, int RQ, bool FAST, int WPB, int S1C = -1, bool PAIR = false>
int launch_core128_wpb(hssfsst_plan* pl, const hssfsst::Core128Params& cp, int64_t nchunks, hipStream_t st)
{
    size_t lds = core128_lds_bytes(pl, RQ, NT, WPB, PAIR);
#ifdef HSS_LDS_PAD                                       // development: one block per CU whatever its size
    if (lds < 100 * 1024) lds = 100 * 1024;
#endif
    auto kern = hssfsst::fsst_core128_kernel<NT, RQ, kFpw128, FAST, WPB, S1C, false, false, PAIR>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->core128_slots == 0) {                        // persistent grid = what is resident at once
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        if (per_cu < 1) per_cu = 1;
        if (cus < 1) cus = 1;
        pl->core128_slots = per_cu * cus;
    }
    int64_t blocks = nchunks;                            // small launches: one chunk per block, spread over the CUs
    if (blocks > pl->core128_slots) blocks = pl->core128_slots;
    name_kernel(pl, WPB, blocks, "fsst_core128_kernel<%d, %d, %d, %s, %d, %d, false%s>", NT, RQ, kFpw128, FAST ? "true" : "false", WPB, S1C, PAIR ? ", pairs" : "");
    if constexpr (PAIR) {                                // (a pair's bounded wait reports through the status word)
        if (int rcs = ensure_status(pl)) return rcs;
        hssfsst::Core128Params cq = cp;
        cq.status = pl->d_status;
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(64 * WPB), lds, st, cq);
    } else
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(64 * WPB), lds, st, cp);
    HIP_TRY(hipGetLastError());
    return 0;
}

// One rolling step in ONE launch (fsst_core128_kernel<.., STREAM>, fsst_mfma128.hpp): the chunk's groups, one per ticket, blocks
// bound to channels; tape append, transform, running-moments merge and normalisation.  Returns 1 when it launched, 0 when the
// step should take the three-launch route (a shape whose plain transform would not be this kernel's one-group chunks).
int ensure_status(hssfsst_plan* pl);

template <int NT, int RQ, int WPB, bool PAIR>
int launch_stream(hssfsst_plan* pl, float* tape_at, long long tape_len, const float* x_new_dev, long long x_stride, int channels, int chunk,
                  float* out, double* state, float* mirror, hipStream_t st)
{
    const int ngroups = (chunk + 15) / 16;
    const hssfsst::Core128Regions reg = hssfsst::core128_regions(ngroups, channels);
    if (!(reg.npc[0] == 0 && reg.npc[1] == 0 && reg.gpc[2] == 1)) return 0;      // (the plain kernel's chunks are not single groups)
    const size_t lds = core128_lds_bytes(pl, RQ, NT, WPB, PAIR);
    if (lds > kMaxLdsBytes) return 0;
    auto kern = hssfsst::fsst_core128_kernel<NT, RQ, kFpw128, true, WPB, -1, false, true, PAIR>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->stream_slots == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        pl->stream_slots = (per_cu < 1 ? 1 : per_cu) * (cus < 1 ? 1 : cus);
    }
    if (state && pl->stream_arrive_cap < channels) {
        if (pl->d_stream_arrive) { HIP_TRY(hipFree(pl->d_stream_arrive)); pl->d_stream_arrive = nullptr; pl->stream_arrive_cap = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pl->d_stream_arrive), static_cast<size_t>(channels) * sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(pl->d_stream_arrive, 0, static_cast<size_t>(channels) * sizeof(unsigned), st));
        pl->stream_arrive_cap = channels;
    }
    if (state && pl->stream_pieces_cap < static_cast<long long>(channels) * ngroups) {
        if (pl->d_stream_pieces) { HIP_TRY(hipFree(pl->d_stream_pieces)); pl->d_stream_pieces = nullptr; pl->stream_pieces_cap = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pl->d_stream_pieces), static_cast<size_t>(channels) * ngroups * 4 * sizeof(double)));
        pl->stream_pieces_cap = static_cast<long long>(channels) * ngroups;
    }
    if (int rcs = ensure_status(pl)) return rcs;
    int bpc = pl->stream_slots / channels;                // blocks per channel: spread a small step over the chip
    if (bpc > ngroups) bpc = ngroups;
    if (bpc < 1) bpc = 1;
    hssfsst::Core128Params cp{};
    cp.x = tape_at; cp.xstride = tape_len; cp.out = out; cp.partials = nullptr; cp.atab = pl->d_atab;
    cp.wtab = pl->d_wtab; cp.twtab = pl->d_wtab + 2 * pl->nwin; cp.r2scale = pl->r2scale;
    cp.n = pl->nwin - 1 + chunk; cp.klo = pl->klo; cp.K = pl->K; cp.mode = pl->mode; cp.nsig = channels;
    cp.col0 = pl->nwin / 2; cp.ncols = chunk; cp.reg = reg;
    cp.xnew = x_new_dev; cp.xnew_stride = x_stride; cp.hist = pl->nwin - 1; cp.bpc = bpc; cp.state = state; cp.arrive = pl->d_stream_arrive; cp.pieces = pl->d_stream_pieces; cp.mirror = mirror;
    const long long grid = static_cast<long long>(channels) * bpc;
    cp.status = pl->d_status;
    name_kernel(pl, WPB, grid, "fsst_core128_kernel<%d, %d, %d, true, %d, -1, false, stream%s>", NT, RQ, kFpw128, WPB, PAIR ? ", pairs" : "");
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(64 * WPB), lds, st, cp);
    HIP_TRY(hipGetLastError());
    return 1;
}

int grow(void** ptr, size_t* cap, size_t need, size_t elem)
{
    if (need <= *cap) return 0;
    if (*ptr) { HIP_TRY(hipFree(*ptr)); *ptr = nullptr; *cap = 0; }
    hipError_t e = hipMalloc(ptr, need * elem);
    if (e != hipSuccess) { *ptr = nullptr; return fail(HSSFSST_ENOMEM, "hipMalloc(%zu B): %s", need * elem, hipGetErrorString(e)); }
    *cap = need;
    return 0;
}

int ensure_status(hssfsst_plan* pl);

// Fused z-score launch (nwin = 128, STACK, wide-store epilogue; fsst_mfma128.hpp "Fused z-score"): one persistent block
// per CU, every CU owns whole signals.  Returns 1 when it launched, 0 when this exec should take the two-kernel path
// (signal too long for the LDS partials, or a batch that would leave CUs idle for a whole signal), < 0 on error.
template <int S1C>
int launch_fused128(hssfsst_plan* pl, hssfsst::Core128Params cp, int64_t batch, int ngroups, hipStream_t st)
{
    constexpr int WPB = 16;
    if (ngroups > hssfsst::kFusedMaxGroups || (ngroups + kFpw128 / 16 - 1) / (kFpw128 / 16) < hssfsst::kFusedMinChunks) return 0;
    const size_t lds = (hssfsst::core128_atab_floats(8, 16) + hssfsst::kCtlFusedFloats + static_cast<size_t>(WPB) *
                        hssfsst::wave_lds_floats(kFpw128, pl->klo, pl->K, 8, 16)) * sizeof(float);
    if (lds > static_cast<size_t>(kMaxLdsBytes)) return 0;
    auto kern = hssfsst::fsst_core128_kernel<16, 8, kFpw128, true, WPB, S1C, true>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->fused_slots == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        pl->fused_slots = (per_cu >= 1 && cus >= 1) ? cus : -1;       // one block per CU
    }
    if (pl->fused_slots < 1) return 0;
    // signals are dealt to the blocks round-robin and a signal is never split: the last round must be nearly full
    // (a quarter-full last round of 4 costs 4 / 3.25 = 23 %), otherwise the chunk-balanced two-kernel path wins
    const int64_t grid = pl->fused_slots;
    const int64_t rounds = (batch + grid - 1) / grid;
    if (batch < grid || rounds * grid * 100 > batch * 112) return 0;
    if (int rcs = ensure_status(pl)) return rcs;
    cp.status = pl->d_status;
    name_kernel(pl, WPB, grid, "fsst_core128_kernel<16, 8, %d, true, %d, %d, true>", kFpw128, WPB, S1C);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(64 * WPB), lds, st, cp);
    HIP_TRY(hipGetLastError());
    return 1;
}

// The same for the general epilogue (any K: nwin 256 / 512, or nwin 128 with an odd or wide band), STACK only.  The z-score
// tickets sweep a chunk as float4s: every signal's feature block has to start on a 16-byte boundary.
template <int NT, int RQ, int WPB, int S1C>
int launch_fused_general(hssfsst_plan* pl, hssfsst::Core128Params cp, int64_t batch, int ngroups, hipStream_t st)
{
    if (ngroups > hssfsst::kFusedMaxGroups || (ngroups + kFpw128 / 16 - 1) / (kFpw128 / 16) < hssfsst::kFusedMinChunks) return 0;
    if (((static_cast<long long>(cp.ncols) * 2 * pl->K) & 3) != 0 || (reinterpret_cast<uintptr_t>(cp.out) & 15) != 0) return 0;
    const size_t lds = (hssfsst::core128_atab_floats(RQ, NT) + hssfsst::kCtlFloats + static_cast<size_t>(WPB) *
                        hssfsst::wave_lds_floats(kFpw128, pl->klo, pl->K, RQ, NT)) * sizeof(float);
    if (lds > static_cast<size_t>(kMaxLdsBytes)) return 0;
    auto kern = hssfsst::fsst_core128_kernel<NT, RQ, kFpw128, false, WPB, S1C, true>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->fused_slots == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        pl->fused_slots = (per_cu >= 1 && cus >= 1) ? cus : -1;       // one block per CU
    }
    if (pl->fused_slots < 1) return 0;
    const int64_t grid = pl->fused_slots;
    const int64_t rounds = (batch + grid - 1) / grid;
    if (batch < grid || rounds * grid * 100 > batch * 112) return 0;  // (as launch_fused128: a nearly full last round)
    if (int rcs = ensure_status(pl)) return rcs;
    cp.status = pl->d_status;
    name_kernel(pl, WPB, grid, "fsst_core128_kernel<%d, %d, %d, false, %d, %d, true>", NT, RQ, kFpw128, WPB, S1C);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(64 * WPB), lds, st, cp);
    HIP_TRY(hipGetLastError());
    return 1;
}

// The status word of the in-kernel waits lives in pinned, device-mapped host memory: a kernel that gave up writes it
// with a system-scope store, and the host looks at it without synchronising (at the start of the next exec, in
// hssfsst_plan_check after a synchronisation, at plan destruction).
int ensure_status(hssfsst_plan* pl)
{
    if (pl->d_status) return 0;
    void* h = nullptr;
    HIP_TRY(hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped));
    *static_cast<volatile unsigned*>(h) = 0u;
    void* d = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
    pl->h_status = static_cast<volatile unsigned*>(h);
    pl->d_status = static_cast<unsigned*>(d);
    return 0;
}

// Team kernels: [0] arrival counter, [1] abort word on the device; the identity of the last launch that gave up in pinned host memory.
int ensure_team_words(hssfsst_plan* pl, hipStream_t st)
{
    if (pl->d_arrive) return 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pl->d_arrive), 4 * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(pl->d_arrive, 0, 4 * sizeof(unsigned), st));
    pl->arrive_total = 0;
    pl->done_total = 0;
    void* h = nullptr;
    HIP_TRY(hipHostMalloc(&h, 4 * sizeof(unsigned), hipHostMallocMapped));
    for (int i = 0; i < 4; ++i) static_cast<volatile unsigned*>(h)[i] = 0u;
    void* d = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
    pl->h_fallback = static_cast<volatile unsigned*>(h);
    pl->d_fallback = static_cast<unsigned*>(d);
    return 0;
}

// Team kernel at four waves per SIMD (fsst_team16.hpp): canonical band, STACK, one 16-frame group per ticket, one group image
// held in registers.  Returns 1 when it launched, 0 when this exec should take another path, < 0 on error.
template <int KLO, int KC, int WPB, int DEPTH>
int launch_team16(hssfsst_plan* pl, const hssfsst::Core128Params& cp, int64_t batch, int ngroups, hipStream_t st)
{
    using namespace hssfsst;
    const int G = ngroups;
    if (G < 1 || G > kFusedMaxGroups) return 0;             // (the resolver's LDS copy of a signal's partials: 128 groups)
    // (at least 84 KiB: one block per CU whatever its size -- the teams count on it)
    constexpr int PSLOTS = t16_pslots<KLO, KC>();        // signals whose partials a CU keeps in LDS at a time
    constexpr int MS = t16_slots<KLO, KC>();             // statistics / mailbox slots the kernel's LDS has room for
    size_t lds = (kCanonLdsTabFloats + t16_ctl_floats(PSLOTS, MS) + static_cast<size_t>(WPB) * CanonCfg<KLO, KC>::wave_floats(t16_planes<KLO, KC>())) * sizeof(float);
    if (lds > static_cast<size_t>(kMaxLdsBytes)) return 0;
    if (lds < 84 * 1024) lds = 84 * 1024;
    auto kern = fsst_team16_kernel<KLO, KC, WPB, DEPTH>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->team16_cus == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        pl->team16_cus = (per_cu >= 1 && cus >= 1) ? cus : -1;
    }
    if (pl->team16_cus < 1) return 0;
    // team size: the smallest power of two that leaves a CU at most 16 groups of a signal (its 16 waves then have all of them in
    // flight at once and the kernel's progress argument holds); HSSFSST_TEAM=n overrides upwards (A/B)
    const int team_env = debug_switches().team;
    int T = 1;
    while ((WPB / 2) * T < G) T *= 2;                    // (cpc <= WPB is the kernel's progress argument; cpc <= WPB / 2 measured faster:
                                                         //  a signal's groups are handed out within half a round of the CU's waves)
    if (team_env > T) { int t2 = T; while (2 * t2 <= team_env && 2 * t2 <= G) t2 *= 2; T = t2; }
    if (T > pl->team16_cus || T > 64) return 0;
    int cpc = 1, cpc_shift = 0;                          // list positions per CU and signal (power of two; surplus ones are skipped)
    while (cpc * T < G) { cpc *= 2; ++cpc_shift; }
    if (cpc > WPB || cpc > kT16MaxCpc || G / T < 1) return 0;
    if (cpc < 4 && T > 1) return 0;                      // (a CU publishes whole blocks of four groups: HSSFSST_TEAM beyond G / 4)
    // as many teams as the chip has room for, but no more than there are signals: the dataset loop's one frame per call
    // (/root/reference/hss/datasets/heart_sounds.py:166-168) starts one team's 16 blocks, not 256 of which 240 find nothing to do
    int nteams = pl->team16_cus / T;
    if (batch < nteams) nteams = static_cast<int>(batch);
    const int grid = nteams * T;
    if ((batch + nteams - 1) / nteams > 65535) return 0;
    if (cp.xstride < 1 || cp.xstride > 0x7fffffffLL || batch > 0x7fffffffLL) return 0;      // (the kernel's 32-bit signal index and stride)
    // slots: a CU runs at most held_pos list positions ahead of its oldest unresolved signal = lead signals; a slot is reused
    // 2 lead + 2 signals later at the earliest (fsst_team16.hpp "Progress")
    const int held_pos = WPB * (DEPTH + 2 + t16_planes<KLO, KC>());   // list positions a CU's waves hold: DEPTH held (+ one in its plane) + transformed + landed + drawn each
    const int lead = (held_pos + G / T - 1) / (G / T) + 1;
    int slots = 8;
    while (slots < 2 * lead + 2) slots *= 2;
    if (slots > MS) return 0;
    if (lead + 1 > PSLOTS) return 0;                     // (very short signals: more signals in flight per CU than its LDS keeps partials for)
    int rc;
    if ((rc = ensure_status(pl)) != 0) return rc;
    const size_t words = static_cast<size_t>(nteams) * slots * kT16SlotWords;
    if (words > pl->mail_cap) {
        if ((rc = grow(reinterpret_cast<void**>(&pl->d_mail), &pl->mail_cap, words, sizeof(unsigned long long))) != 0) return rc;
        HIP_TRY(hipMemsetAsync(pl->d_mail, 0, pl->mail_cap * sizeof(unsigned long long), st));
        pl->team_seq = 0;
    }
    if (++pl->team_seq > 0xffffu) {                      // tags would repeat: start over from clean mailboxes
        HIP_TRY(hipMemsetAsync(pl->d_mail, 0, pl->mail_cap * sizeof(unsigned long long), st));
        pl->team_seq = 1;
    }
    Team16Params tp{};
    tp.x = cp.x; tp.out = cp.out; tp.atab = pl->d_atab16; tp.wtab = cp.wtab; tp.twtab = cp.twtab;
    tp.mail = pl->d_mail; tp.status = pl->d_status; tp.r2scale_s = pl->canon_r2s; tp.inv_c = pl->canon_inv_c;
    tp.n = cp.n; tp.nsig = cp.nsig; tp.col0 = cp.col0; tp.ncols = cp.ncols; tp.xstride = cp.xstride;
    tp.team = T; tp.cpc_shift = cpc_shift; tp.slots = slots; tp.seq = pl->team_seq;
    {   // the two float64 divisions of stats_finish (correctly rounded here as there: the same bits)
        const double total = static_cast<double>(KC) * static_cast<double>(cp.ncols);
        tp.inv_total = 1.0 / total; tp.inv_total1 = 1.0 / (total - 1.0);
    }
    const unsigned spin_us = debug_switches().team_spin_us;
    tp.spin_ticks = (spin_us < 10u ? 10u : spin_us > 10000000u ? 10000000u : spin_us) * 100u;
    if ((rc = ensure_team_words(pl, st)) != 0) return rc;
    if (++pl->team_launch == 0u) pl->team_launch = 1u;
    tp.abort_word = pl->d_arrive + 1; tp.fallbacks = pl->d_fallback; tp.launch = pl->team_launch;
    const bool force_fallback = debug_switches().team_force_fallback;   // tests: every team launch finds itself given up
    if (force_fallback) {
        HIP_TRY(hipMemcpyAsync(pl->d_arrive + 1, &pl->team_launch, sizeof(unsigned), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(pl->d_fallback, &pl->team_launch, sizeof(unsigned), hipMemcpyHostToDevice, st));
    }
    tp.arrive = pl->d_arrive; tp.arrive_base = pl->arrive_total;
    pl->arrive_total += static_cast<unsigned>(grid);     // (a plan is single-stream: every block of the earlier launches has arrived)
    pl->flag_launch = 0u;
    if (pl->flag_done && !force_fallback) {              // (hssfsst_exec_pinned & co: the host waits for this exec alone)
        tp.done = pl->d_arrive + 2; tp.done_base = pl->done_total; tp.host_done = pl->d_fallback + 2;
        pl->done_total += static_cast<unsigned>(grid);
        pl->flag_launch = pl->team_launch;
    }
    name_kernel(pl, WPB, grid, "fsst_team16_kernel<%d, %d, %d, %d> teams of %d", KLO, KC, WPB, DEPTH, T);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(64 * WPB), lds, st, tp);
    HIP_TRY(hipGetLastError());
    return 1;
}

// Canonical-class kernels (fsst_canon128.hpp): nwin = 128, STACK / STACK_UNNORM, the kept band a compile-time constant.  Two bands
// are instantiated: rows 4..25 = [25, 200] Hz at fs = 1000 (/root/reference/main.py:153-158, the reference's own configuration) and
// rows 2..25 = [25, 400] Hz at fs = 2000 -- the pass band of the Springer / Schmidt heart-sound segmenters on recordings at
// PhysioNet 2016's native rate (/root/reference/hss/datasets/heart_sounds.py:36-113 loads them un-resampled).  Every other band keeps
// fsst_core128_kernel.  (A third band is one line here: the template takes any even band of <= 24 rows inside rows 0..31.)
constexpr int kCanonBands[][2] = {{4, 22}, {2, 24}};
constexpr int kCanonKlo = kCanonBands[0][0], kCanonK = kCanonBands[0][1];

int canon_band(const hssfsst_plan* pl)                   // index into kCanonBands, or -1
{
    if (debug_switches().no_canon || !pl->d_atab16 || pl->nwin != 128) return -1;     // (no_canon: A/B and cross-check tests)
    if (!(pl->mode == HSSFSST_MODE_STACK || pl->mode == HSSFSST_MODE_STACK_UNNORM)) return -1;
    for (int i = 0; i < static_cast<int>(sizeof(kCanonBands) / sizeof(kCanonBands[0])); ++i)
        if (pl->klo == kCanonBands[i][0] && pl->K == kCanonBands[i][1]) return i;
    return -1;
}
bool plan_is_canon(const hssfsst_plan* pl) { return canon_band(pl) >= 0; }
// f(KLO, KC) with the plan's band as integral constants
template <class F>
int canon_dispatch(const hssfsst_plan* pl, F&& f)
{
    switch (canon_band(pl)) {
    case 0: return f(std::integral_constant<int, kCanonBands[0][0]>{}, std::integral_constant<int, kCanonBands[0][1]>{});
    case 1: return f(std::integral_constant<int, kCanonBands[1][0]>{}, std::integral_constant<int, kCanonBands[1][1]>{});
    default: return fail(HSSFSST_EINVAL, "canon_dispatch: not a canonical-class plan");
    }
}

hssfsst::CanonParams canon_params(const hssfsst_plan* pl, const hssfsst::Core128Params& cp)
{
    hssfsst::CanonParams q{};
    q.x = cp.x; q.out = cp.out; q.partials = cp.partials; q.atab = pl->d_atab16; q.wtab = cp.wtab; q.twtab = cp.twtab;
    q.r2scale_s = pl->canon_r2s; q.inv_c = pl->canon_inv_c;
    q.n = cp.n; q.mode = cp.mode; q.nsig = cp.nsig; q.col0 = cp.col0; q.ncols = cp.ncols; q.xstride = cp.xstride; q.reg = cp.reg;
    q.status = cp.status;
    q.gate = pl->gate; q.gate_val = pl->gate_val;
    return q;
}

template <int KLO, int KC>
int launch_canon_band(hssfsst_plan* pl, const hssfsst::Core128Params& cp, int64_t nchunks, hipStream_t st)
{
    constexpr int WPB = 16;
    const size_t lds = (hssfsst::kCanonLdsTabFloats + hssfsst::kCanonCtlFloats + static_cast<size_t>(WPB) * hssfsst::CanonCfg<KLO, KC>::wave_floats()) * sizeof(float);
    static_assert((hssfsst::kCanonLdsTabFloats + hssfsst::kCanonCtlFloats + 16 * hssfsst::CanonCfg<KLO, KC>::wave_floats()) * sizeof(float) <= 160 * 1024, "16 wave regions must fit");
    auto kern = hssfsst::fsst_canon_kernel<KLO, KC, false>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->canon_slots == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        if (per_cu < 1) per_cu = 1;
        if (cus < 1) cus = 1;
        pl->canon_slots = per_cu * cus;
    }
    int64_t blocks = nchunks;
    if (blocks > pl->canon_slots) blocks = pl->canon_slots;
    // (a gated launch -- the fallback behind a team launch -- almost always finds its gate closed: a quarter of the chip keeps
    //  the empty launch at ~2 us instead of ~4; when it does run, the GPU is shared anyway)
    if (pl->gate != nullptr && blocks > 64) blocks = 64;
    if (pl->gate == nullptr) name_kernel(pl, WPB, blocks, "fsst_canon_kernel<%d, %d, false>", KLO, KC);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(64 * WPB), lds, st, canon_params(pl, cp));
    HIP_TRY(hipGetLastError());
    return 0;
}
int launch_canon(hssfsst_plan* pl, const hssfsst::Core128Params& cp, int64_t nchunks, hipStream_t st)
{
    return canon_dispatch(pl, [&](auto KL, auto KN) { return launch_canon_band<decltype(KL)::value, decltype(KN)::value>(pl, cp, nchunks, st); });
}

// One CU per signal with the z-score in the same launch (see launch_fused128): 1 = launched, 0 = take another path
// gated = the fallback queued behind a team launch (pl->gate set): any batch size -- the block count is what a launch that
// almost always finds its gate closed should cost, not what would be fast.
template <int KLO, int KC>
int launch_canon_fused_band(hssfsst_plan* pl, hssfsst::Core128Params cp, int64_t batch, int ngroups, hipStream_t st, bool gated)
{
    constexpr int WPB = 16, GPC = hssfsst::kCanonTileFrames / 16;
    if (ngroups > hssfsst::kFusedMaxGroups || (ngroups + GPC - 1) / GPC < hssfsst::kFusedMinChunks) return 0;
    constexpr size_t lds = (hssfsst::kCanonLdsTabFloats + hssfsst::kCanonCtlFusedFloats + static_cast<size_t>(WPB) * hssfsst::CanonCfg<KLO, KC>::wave_floats()) * sizeof(float);
    // (rows 2..25: 16 wave regions of 8.6 kB and the two signals' partials do not fit the 160 KiB together: team kernel or two launches)
    if constexpr (lds > static_cast<size_t>(kMaxLdsBytes)) return 0;
    else {
    auto kern = hssfsst::fsst_canon_kernel<KLO, KC, true>;
    static std::atomic<unsigned long long> lds_ok{0};
    if (int rc = allow_full_lds(kern, pl->device, lds_ok)) return rc;
    if (pl->fused_slots == 0) {
        int per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * WPB, lds));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pl->device));
        pl->fused_slots = (per_cu >= 1 && cus >= 1) ? cus : -1;
    }
    if (pl->fused_slots < 1) return 0;
    int64_t grid = pl->fused_slots;
    const int64_t rounds = (batch + grid - 1) / grid;
    // (gated = behind a team launch: almost always the gate is closed and 64 blocks keep the empty launch short)
    if (gated) grid = batch < 64 ? batch : 64;
    else if (batch < grid || rounds * grid * 100 > batch * 112) return 0;
    if (int rcs = ensure_status(pl)) return rcs;
    cp.status = pl->d_status;
    if (!gated) name_kernel(pl, WPB, grid, "fsst_canon_kernel<%d, %d, true>", KLO, KC);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(64 * WPB), lds, st, canon_params(pl, cp));
    HIP_TRY(hipGetLastError());
    return 1;
    }
}
int launch_canon_fused(hssfsst_plan* pl, const hssfsst::Core128Params& cp, int64_t batch, int ngroups, hipStream_t st, bool gated = false)
{
    return canon_dispatch(pl, [&](auto KL, auto KN) { return launch_canon_fused_band<decltype(KL)::value, decltype(KN)::value>(pl, cp, batch, ngroups, st, gated); });
}

// What the host has just learnt about the plan's team launches: `gave_up` of them (seen since the last look) among `total` it knows to have
// finished.  kTeamGiveUps give-ups without a success in between -> the next kTeamPause execs skip the team kernel (hssfsst.h).
constexpr int kTeamGiveUps = 4, kTeamPause = 256;
void note_team_outcome(hssfsst_plan* p, bool gave_up)
{
    if (!gave_up) { p->giveups_in_a_row = 0; return; }
    if (++p->giveups_in_a_row >= kTeamGiveUps) { p->giveups_in_a_row = 0; p->team_pause = kTeamPause; }
}

int plan_next_event(hssfsst_plan* p, hipEvent_t* out_ev)
{
    if (p->ev.size() <= p->ev_used) {
        hipEvent_t e2 = nullptr;
        HIP_TRY(hipEventCreate(&e2));
        p->ev.push_back(e2);
    }
    *out_ev = p->ev[p->ev_used++];
    return 0;
}

int launch_core128(hssfsst_plan* pl, const float* dx, long long xstride, float* dout, float* partials, int n, int col0,
                   int ncols, int64_t batch, hipStream_t st, bool try_fused, bool* did_fuse)
{
    hssfsst::Core128Params cp{};
    cp.xstride = xstride;
    cp.x = dx; cp.out = dout; cp.partials = partials; cp.atab = pl->d_atab;
    cp.wtab = pl->d_wtab; cp.twtab = pl->d_wtab + 2 * pl->nwin; cp.r2scale = pl->r2scale;
    cp.n = n; cp.klo = pl->klo; cp.K = pl->K; cp.mode = pl->mode; cp.nsig = static_cast<int>(batch);
    cp.col0 = col0; cp.ncols = ncols;
    cp.reg = hssfsst::core128_regions((ncols + 15) / 16, batch);
    const int64_t nchunks = batch * hssfsst::core128_chunks_per_signal(cp.reg);
    const bool fast = (pl->mode == HSSFSST_MODE_STACK || pl->mode == HSSFSST_MODE_STACK_UNNORM) &&
                      (pl->K & 1) == 0 && pl->K <= 24;
    // waves per block: as many wave regions as fit the 160 KiB of LDS beside the shared tables
    const int rq = pl->rq, nt = pl->nt;
    const size_t fixed = (hssfsst::core128_atab_floats(rq, nt) + hssfsst::kCtlFloats) * sizeof(float);
    const size_t per_wave = static_cast<size_t>(hssfsst::wave_lds_floats(kFpw128, pl->klo, pl->K, rq, nt)) * sizeof(float);
    const size_t room = 160 * 1024;
    // bands that start in stripe 0 of the own plane and end in stripe 3 (the canonical [25, 200] Hz at fs = 1000 for
    // nwin 128 and 256) get kernels with compile-time stripe tests
    const bool canon = hssfsst::own_s0(pl->klo, rq) == 0 && hssfsst::own_s1(pl->klo, pl->K, rq) == 3;
    *did_fuse = false;
    // (tiles are aligned in absolute columns: a column range must start on a 16-frame group boundary -- on a 64-frame tile
    //  boundary for the team kernel, whose chunks are whole tiles; other ranges take fsst_core128_kernel)
    const bool canon16 = fast && nt == 16 && rq == 8 && plan_is_canon(pl) && (col0 & 15) == 0;
    if (try_fused && fast && nt == 16 && rq == 8 && pl->mode == HSSFSST_MODE_STACK) {
        const int ngroups = (ncols + 15) / 16;
        // two single-launch z-score kernels.  The team kernel (fsst_team16.hpp: features written once, from registers) is at least as
        // fast as one CU per signal on full batches and 1.3-1.8x faster on small or ragged ones (profiles/r04_batch_sweep.txt): it
        // goes first wherever it applies -- the canonical band, signals of at most 128 groups; one CU per signal (the tile makes a
        // round trip through HBM inside the launch) for the other even bands of <= 24 rows, and as the team kernel's gated fallback
        const bool env_no_team = debug_switches().no_team, env_team_only = debug_switches().team_only;      // A/B and tests
        const bool no_team = env_no_team || pl->zpath_pref == HSSFSST_ZPATH_ONE_CU;
        const bool team_only = env_team_only || pl->zpath_pref == HSSFSST_ZPATH_TEAM;
        int rc = 0;
        // (the host learns of give-ups where it synchronises anyway -- host-output execs, hssfsst_plan_fallbacks, hssfsst_plan_check -- and lets
        //  the team kernel sit out a while when they come in a row: note_team_outcome)
        const bool paused = pl->team_pause > 0 && !team_only;
        if (paused && canon16 && !no_team) --pl->team_pause;
        if (!no_team && canon16 && !paused) {
            rc = canon_dispatch(pl, [&](auto KL, auto KN) { return launch_team16<decltype(KL)::value, decltype(KN)::value, HSS_T16_WPB, HSS_T16_DEPTH>(pl, cp, batch, ngroups, st); });
            if (rc == 1) {
                // the team kernel may give the launch up (its blocks wait for each other; other processes on the GPU can keep
                // them apart: fsst_team16.hpp "Progress"): the same exec is queued behind it, every kernel of it gated on the
                // abort word -- a few microseconds of empty launches when nothing went wrong
                pl->last_zpath = 2;
                if (pl->timing) {                        // the kernel's own time: the closing event goes in front of the gated launches
                    hipEvent_t evt = nullptr;
                    if (int rce = plan_next_event(pl, &evt)) return rce;
                    HIP_TRY(hipEventRecord(evt, st));
                    pl->timing_closed = true;
                }
#ifdef HSS_NO_GATE                                        // development: what the gated launch behind every team launch costs (UNSAFE: no fallback)
                if (true) {
#else
                if (pl->defer_fallback) {                // (exec_impl: a host-output exec synchronises anyway and checks the give-up word then)
#endif
                    if (pl->deferred_launch == 0u) pl->deferred_first = pl->team_launch;
                    pl->deferred_launch = pl->team_launch;
                    *did_fuse = true;
                    return 0;
                }
                pl->gate = pl->d_arrive + 1; pl->gate_val = pl->team_launch;
                // ONE gated launch where the one-CU-per-signal kernel applies (signals of 16 .. 32 chunks; its batch
                // conditions are about speed only): 4 us behind the team kernel instead of 11 for transform + statistics +
                // z-score launches -- a third of a 50-window exec
                const int rc1 = launch_canon_fused(pl, cp, batch, ngroups, st, true);
                if (rc1 < 0) { pl->gate = nullptr; return rc1; }
                if (rc1 == 1) { pl->gate = nullptr; *did_fuse = true; return 0; }
                const int rc2 = launch_canon(pl, cp, nchunks, st);
                if (rc2 != 0) { pl->gate = nullptr; return rc2; }
                *did_fuse = false;                       // (exec_impl adds the gated statistics + z-score launches)
                return 0;
            }
            if (rc < 0) return rc;
        }
        if (!team_only) {
            rc = canon16 ? launch_canon_fused(pl, cp, batch, ngroups, st)
                 : canon ? launch_fused128<3>(pl, cp, batch, ngroups, st) : launch_fused128<-1>(pl, cp, batch, ngroups, st);
        }
        if (rc < 0) return rc;
        if (rc == 1) { *did_fuse = true; pl->last_zpath = 1; return 0; }
    }
    if (try_fused && !fast && rq == 8 && nt == 16 && pl->mode == HSSFSST_MODE_STACK && pl->zpath_pref != HSSFSST_ZPATH_TEAM &&
        fixed + 16 * per_wave <= room) {
        // nwin 128, a band the wide-store epilogue does not take (odd K or K > 24)
        const int rc = launch_fused_general<16, 8, 16, -1>(pl, cp, batch, (ncols + 15) / 16, st);
        if (rc < 0) return rc;
        if (rc == 1) { *did_fuse = true; pl->last_zpath = 1; return 0; }
    }
#ifndef HSS_DEV_ONLY128
    if (try_fused && !fast && rq == 16 && pl->mode == HSSFSST_MODE_STACK && pl->zpath_pref != HSSFSST_ZPATH_TEAM) {
        // nwin 256 / 512 (general epilogue): one CU per signal, the z-score as tickets of the same launch; waves per block
        // as on the two-launch path (what fits the LDS: 8 for the canonical band at 256 points, 3 at 512)
        const int ngroups = (ncols + 15) / 16;
        int rc = 0;
        if (nt == 16 && fixed + 8 * per_wave <= room)
            rc = canon ? launch_fused_general<16, 16, 8, 3>(pl, cp, batch, ngroups, st) : launch_fused_general<16, 16, 8, -1>(pl, cp, batch, ngroups, st);
        // (512 points: two launches, on wave pairs: 2.6 ms per 1024 windows against 3.0 for the single launch, whose
        //  instantiations are gone)
        if (rc < 0) return rc;
        if (rc == 1) { *did_fuse = true; pl->last_zpath = 1; return 0; }
    }
#endif
    if (nt == 16 && rq == 8) {
        if (canon16) return launch_canon(pl, cp, nchunks, st);
#ifdef HSS_WPB_CANON
        if (fast && canon) return launch_core128_wpb<16, 8, true, HSS_WPB_CANON, 3>(pl, cp, nchunks, st);
#endif
        if (fast && canon) return launch_core128_wpb<16, 8, true, 16, 3>(pl, cp, nchunks, st);
        if (fast) return launch_core128_wpb<16, 8, true, 16>(pl, cp, nchunks, st);   // K <= 24: 16 regions always fit
        if (fixed + 16 * per_wave <= room) return launch_core128_wpb<16, 8, false, 16>(pl, cp, nchunks, st);
        if (fixed + 8 * per_wave <= room) return launch_core128_wpb<16, 8, false, 8>(pl, cp, nchunks, st);
        if (fixed + 4 * per_wave <= room) return launch_core128_wpb<16, 8, false, 4>(pl, cp, nchunks, st);
#ifndef HSS_DEV_ONLY128
    } else if (nt == 16 && rq == 16) {                                               // nwin = 256
        // (wave pairs -- fsst_mfma128.hpp "PAIR" -- lose here: 16 waves at 128 registers spill, core 0.770 vs 0.587 ms per 1024
        //  windows; 12 waves at 170 registers: 0.739 ms)
        // (8 waves per block at most: two per SIMD, up to 256 VGPRs, no scratch)
        if (fast && fixed + 8 * per_wave <= room) return launch_core128_wpb<16, 16, true, 8>(pl, cp, nchunks, st);
        if (!fast && canon && fixed + 8 * per_wave <= room) return launch_core128_wpb<16, 16, false, 8, 3>(pl, cp, nchunks, st);
        if (!fast && fixed + 8 * per_wave <= room) return launch_core128_wpb<16, 16, false, 8>(pl, cp, nchunks, st);
        if (!fast && fixed + 4 * per_wave <= room) return launch_core128_wpb<16, 16, false, 4>(pl, cp, nchunks, st);
        if (fast && fixed + 4 * per_wave <= room) return launch_core128_wpb<16, 16, true, 4>(pl, cp, nchunks, st);
    } else {                                                                         // nt == 32, rq == 16: nwin = 512
        if (!debug_switches().no_pair) {                                             // (two waves per SIMD at most: 32-point spectra in registers)
            if (fast && core128_lds_bytes(pl, 16, 32, 8, true) <= room) return launch_core128_wpb<32, 16, true, 8, -1, true>(pl, cp, nchunks, st);
            if (!fast && core128_lds_bytes(pl, 16, 32, 8, true) <= room) return launch_core128_wpb<32, 16, false, 8, -1, true>(pl, cp, nchunks, st);
            if (!fast && core128_lds_bytes(pl, 16, 32, 6, true) <= room) return launch_core128_wpb<32, 16, false, 6, -1, true>(pl, cp, nchunks, st);
            if (!fast && core128_lds_bytes(pl, 16, 32, 4, true) <= room) return launch_core128_wpb<32, 16, false, 4, -1, true>(pl, cp, nchunks, st);
        }
        if (fast && fixed + 8 * per_wave <= room) return launch_core128_wpb<32, 16, true, 8>(pl, cp, nchunks, st);
        if (!fast && fixed + 8 * per_wave <= room) return launch_core128_wpb<32, 16, false, 8>(pl, cp, nchunks, st);
        if (!fast && fixed + 6 * per_wave <= room) return launch_core128_wpb<32, 16, false, 6>(pl, cp, nchunks, st);
        if (fast && fixed + 4 * per_wave <= room) return launch_core128_wpb<32, 16, true, 4>(pl, cp, nchunks, st);
        if (!fast && fixed + 4 * per_wave <= room) return launch_core128_wpb<32, 16, false, 4>(pl, cp, nchunks, st);
        if (!fast && fixed + 3 * per_wave <= room) return launch_core128_wpb<32, 16, false, 3>(pl, cp, nchunks, st);
        if (!fast && fixed + 2 * per_wave <= room) return launch_core128_wpb<32, 16, false, 2>(pl, cp, nchunks, st);
        if (fast && fixed + 2 * per_wave <= room) return launch_core128_wpb<32, 16, true, 2>(pl, cp, nchunks, st);
#endif
    }
    return fail(HSSFSST_EUNSUPPORTED, "LDS request %zu B per wave exceeds the 160 KiB budget", per_wave);
}

}  // namespace

extern "C" {
#ifdef HSS_FUSE_PROBE
// development: read and clear the ticket probes of fsst_canon_kernel<.., true> (fsst_canon128.hpp)
int hssfsst_dev_fuse_probe(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hssfsst::g_fuse_probe), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(hssfsst::g_fuse_probe), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef HSS_STREAM_PROBE
int hssfsst_dev_stream_probe(unsigned long long* out, int nwaves)      // out[nwaves][8] of the last launch; cleared
{
    if (nwaves > hssfsst::kStreamProbeWaves) nwaves = hssfsst::kStreamProbeWaves;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hssfsst::g_stream_probe), sizeof(unsigned long long) * 8 * nwaves) != hipSuccess) return -1;
    std::vector<unsigned long long> z(static_cast<size_t>(hssfsst::kStreamProbeWaves) * 8, 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(hssfsst::g_stream_probe), z.data(), z.size() * sizeof(unsigned long long)) != hipSuccess) return -1;
    return 0;
}
#endif

int hssfsst_version(void) { return HSSFSST_VERSION; }
#ifdef HSS_T16_BLKPROBE      // development only (tools/blk_probe.py)
int hssfsst_dev_t16_blk(unsigned* out)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hssfsst::g_t16_blk), sizeof(unsigned) * 256 * 16 * 8) == hipSuccess ? 0 : -2;
}
#endif
const char* hssfsst_last_error(void) { return g_err; }

int hssfsst_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int hssfsst_dtwin(const double* window, int nwin, double fs, double* dwindow)
{
    if (!window || !dwindow || nwin < 1 || !(fs > 0.0)) return fail(HSSFSST_EINVAL, "hssfsst_dtwin: bad argument");
    const int rc = spline_knot_slopes(window, nwin, dwindow);
    if (rc != 0) return fail(rc, "hssfsst_dtwin: singular spline system");
    const double scale = fs / (2.0 * M_PI);
    for (int i = 0; i < nwin; ++i) dwindow[i] *= scale;
    return 0;
}

int hssfsst_band(int nwin, double fs, double f_lo, double f_hi, int* klo, int* K)
{
    if (nwin < 1 || !(fs > 0.0) || !klo || !K) return fail(HSSFSST_EINVAL, "hssfsst_band: bad argument");
    band_rows(nwin, fs, f_lo, f_hi, klo, K);
    return 0;
}

double hssfsst_update_mean(double m, double x, int64_t k) { return m + (x - m) / static_cast<double>(k); }

double hssfsst_update_variance(double x, double m, double var, int64_t k)
{
    const double delta = x - m;
    return var + delta * (x - (m + delta / static_cast<double>(k)));
}

int hssfsst_plan_create(hssfsst_plan** out, int device, int nwin, const double* window, double fs,
                        int has_band, double f_lo, double f_hi, int mode)
{
    if (!out) return fail(HSSFSST_EINVAL, "plan_create: out is NULL");
    *out = nullptr;
    if (!window || nwin < 1 || !(fs > 0.0) || mode < 0 || mode > HSSFSST_MODE_STACK_UNNORM)
        return fail(HSSFSST_EINVAL, "plan_create: bad argument (nwin=%d fs=%g mode=%d)", nwin, fs, mode);
    if (nwin > 65535) return fail(HSSFSST_EUNSUPPORTED, "plan_create: window length %d exceeds 65535", nwin);
    const bool force_dft = debug_switches().force_dft;    // cross-check: every length on the any-length kernel
    const bool radix_len = nwin == 32 || nwin == 64 || nwin == 128 || nwin == 256 || nwin == 512;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(HSSFSST_ENODEVICE, "plan_create: no HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= ndev) return fail(HSSFSST_EINVAL, "plan_create: device %d out of range [0,%d)", device, ndev);
    DEVICE_SCOPE(device);

    hssfsst_plan* p = new (std::nothrow) hssfsst_plan();
    if (!p) return fail(HSSFSST_ENOMEM, "plan_create: host allocation failed");
    p->device = device; p->nwin = nwin; p->R = nwin / 32; p->nf = nwin / 2 + 1; p->mode = mode; p->fs = fs;
    if (has_band) band_rows(nwin, fs, f_lo, f_hi, &p->klo, &p->K);
    else { p->klo = 0; p->K = p->nf; }

    // derivative window in BIN units: dw * nwin/fs with dw = slope * fs/(2 pi)  =>  slope * nwin/(2 pi)
    std::vector<double> dwb(nwin);
    int rc = spline_knot_slopes(window, nwin, dwb.data());
    if (rc != 0) { delete p; return fail(rc, "plan_create: singular spline system"); }
    for (int i = 0; i < nwin; ++i) dwb[i] *= static_cast<double>(nwin) / (2.0 * M_PI);

    p->dft = (!radix_len || force_dft) ? 1 : 0;
    const int R = p->dft ? 2 : p->R;                      // (the class-folded tables below are only read by the radix kernels)
    const int ncls = R / 2 + 1;
    std::vector<float> tab(static_cast<size_t>(ncls) * 32 * 4 * R, 0.0f);
    if (!p->dft)
    for (int r = 0; r < ncls; ++r) {
        const bool packed = (r == 0) || (2 * r == R);
        for (int n = 0; n < 32; ++n) {
            float* row = tab.data() + (static_cast<size_t>(r) * 32 + n) * 4 * R;
            for (int q = 0; q < R; ++q) {
                const double ang = -2.0 * M_PI * (static_cast<double>(r) * q / R + static_cast<double>(r) * n / nwin);
                const double c = std::cos(ang), s = std::sin(ang);
                const double wv = window[n + 32 * q], dv = dwb[n + 32 * q];
                if (packed) {            // 0.5 (w + i dw') * phase
                    row[q] = static_cast<float>(0.5 * (wv * c - dv * s));
                    row[R + q] = static_cast<float>(0.5 * (wv * s + dv * c));
                } else {                 // w * phase | dw' * phase
                    row[q] = static_cast<float>(wv * c);
                    row[R + q] = static_cast<float>(wv * s);
                    row[2 * R + q] = static_cast<float>(dv * c);
                    row[3 * R + q] = static_cast<float>(dv * s);
                }
            }
        }
    }
    e = hipMalloc(reinterpret_cast<void**>(&p->d_ctab), tab.size() * sizeof(float));
    if (e != hipSuccess) { delete p; return fail(HSSFSST_ENOMEM, "plan_create: hipMalloc: %s", hipGetErrorString(e)); }
    e = hipMemcpy(p->d_ctab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p->d_ctab); delete p; return fail(HSSFSST_EHIP, "plan_create: hipMemcpy: %s", hipGetErrorString(e)); }
    {   // float64 tables of the rounding-tie path: the window pair and the twiddles
        std::vector<double> wt(static_cast<size_t>(4) * nwin);
        double cmax2 = 0.0;
        for (int i = 0; i < nwin; ++i) cmax2 = std::fmax(cmax2, 0.25 * (window[i] * window[i] + dwb[i] * dwb[i]));
        p->r2scale = static_cast<float>(4.0 * nwin * cmax2);
        for (int i = 0; i < nwin; ++i) {
            wt[2 * i] = window[i]; wt[2 * i + 1] = dwb[i];
            const double ang = 2.0 * M_PI * static_cast<double>(i) / static_cast<double>(nwin);
            wt[2 * nwin + 2 * i] = std::cos(ang); wt[2 * nwin + 2 * i + 1] = std::sin(ang);
        }
        if (nwin == 128 || nwin == 256 || nwin == 512) {
            // MFMA kernels, heavily undecided groups (resolve_group_f64, fsst_mfma128.hpp): the A operand of the fold -- the
            // constants C_r[n, q] of the float32 table below, [pass][tap][k-step][lane] -- in float64 for
            // v_mfma_f64_16x16x4_f64, behind the twiddles (16 / 64 / 128 kB).  The float64 instruction hands lane group g the
            // rows g, g + 4, g + 8, g + 12 of D (measured: tools/mfma_f64_layout.hip) where the float32 one hands it rows
            // 4 g .. 4 g + 3: row i of A is class pair 4 pz + (i & 3), component i >> 2.
            const int nt = (nwin == 512) ? 32 : 16, rq = nwin / nt, npass = rq / 8, kst = rq / 4;
            wt.resize(static_cast<size_t>(4) * nwin + static_cast<size_t>(hssfsst::fold64_doubles(rq, nt)));
            for (int pz = 0; pz < npass; ++pz)
                for (int n = 0; n < nt; ++n)
                    for (int ks = 0; ks < kst; ++ks)
                        for (int l = 0; l < 64; ++l) {
                            const int i = l & 15, q = (l >> 4) + 4 * ks;
                            const int gg = i & 3, sub = i >> 2, m = 4 * pz + gg;
                            const int r = (sub < 2) ? m : (m ? rq - m : rq / 2);
                            const double ang = -2.0 * M_PI * (static_cast<double>(r) * q / rq + static_cast<double>(r) * n / nwin);
                            const double c = std::cos(ang), sn = std::sin(ang);
                            const double sg = (r & 1) ? -0.5 : 0.5;
                            const double wv = window[n + nt * q], dv = dwb[n + nt * q];
                            wt[static_cast<size_t>(4) * nwin + (((pz * nt + n) * kst) + ks) * 64 + l] =
                                (sub & 1) ? sg * (wv * sn + dv * c) : sg * (wv * c - dv * sn);
                        }
        }
        e = hipMalloc(reinterpret_cast<void**>(&p->d_wtab), wt.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(p->d_wtab, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (p->d_wtab) (void)hipFree(p->d_wtab);
            (void)hipFree(p->d_ctab); delete p;
            return fail(HSSFSST_EHIP, "plan_create: float64 table upload: %s", hipGetErrorString(e));
        }
    }
    const bool force_generic = debug_switches().force_generic;
    const bool mfma_long = !debug_switches().no_mfma256;   // A/B: nwin 256 / 512 on the generic kernel
    bool use_mfma = !p->dft && (nwin == 128 || ((nwin == 256 || nwin == 512) && mfma_long)) && !force_generic;
    if (p->dft) {
        // A[i][k] of v_mfma_f32_16x16x4_f32 for source block blk, k-step ks: lane l holds row i = l & 15, k = l >> 4.
        // Row i: source k' = 4 blk + (i >> 2), component i & 3 of {V.re, V.im, Vd'.re, Vd'.im}; tap n = 4 ks + k:
        //   V  [k'] = sum_n x[t + n] w  [n] e^{-2 pi i k' (n + m) / N},  m = floor(N / 2) (the modified-STFT phase, step 5)
        //   Vd'[k'] = sum_n x[t + n] dw'[n] e^{-2 pi i k' (n + m) / N}
        // (k' (n + m) is reduced modulo N in integers before the angle is formed)
        const int nf = p->nf, nk4 = (nwin + 3) / 4, nblk4 = (nf + 3) / 4, m = nwin / 2;
        const size_t per_wave = static_cast<size_t>(hssfsst::dft_wave_lds_floats(nk4, p->K > 0 ? p->K : 1)) * sizeof(float);
        if (per_wave > static_cast<size_t>(kMaxLdsBytes)) {
            (void)hipFree(p->d_ctab); (void)hipFree(p->d_wtab); delete p;
            return fail(HSSFSST_EUNSUPPORTED, "plan_create: window length %d with %d kept rows needs %zu B of LDS per wave (> 160 KiB): "
                                              "narrow the band", nwin, p->K, per_wave);
        }
        std::vector<float> dt(static_cast<size_t>(nblk4) * nk4 * 64, 0.0f);
        for (int blk = 0; blk < nblk4; ++blk)
            for (int ks = 0; ks < nk4; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, n = 4 * ks + (l >> 4), kp = 4 * blk + (i >> 2), sub = i & 3;
                    if (kp >= nf || n >= nwin) continue;
                    const long long red = (static_cast<long long>(kp) * (n + m)) % nwin;
                    const double ang = -2.0 * M_PI * static_cast<double>(red) / static_cast<double>(nwin);
                    const double amp = (sub < 2) ? window[n] : dwb[n];
                    dt[(static_cast<size_t>(blk) * nk4 + ks) * 64 + l] = static_cast<float>(amp * ((sub & 1) ? std::sin(ang) : std::cos(ang)));
                }
        e = hipMalloc(reinterpret_cast<void**>(&p->d_dtab), dt.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->d_dtab, dt.data(), dt.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (p->d_dtab) (void)hipFree(p->d_dtab);
            (void)hipFree(p->d_ctab); (void)hipFree(p->d_wtab); delete p;
            return fail(HSSFSST_EHIP, "plan_create: DFT table upload (%zu B): %s", dt.size() * sizeof(float), hipGetErrorString(e));
        }
    }
    const int nt0 = (nwin == 512) ? 32 : 16, rq0 = nwin / nt0;
    if (use_mfma) {                                      // enough wave regions of this band must fit beside the A table
        const int min_waves = (nwin == 512) ? 2 : 4;
        const size_t need = (hssfsst::core128_atab_floats(rq0, nt0) + hssfsst::kCtlFloats + min_waves *
                             static_cast<size_t>(hssfsst::wave_lds_floats(kFpw128, p->klo, p->K, rq0, nt0))) * sizeof(float);
        if (need > 160 * 1024) use_mfma = false;         // (long windows with very wide bands: generic kernel)
    }
    if (use_mfma) {
        // A[i][k] of v_mfma_f32_16x16x4_f32 for pass pz, tap n, k-step ks: lane l holds row i = l & 15, k = l >> 4.
        // Row i: lane group gg = i >> 2 owns class pair m = 4 pz + gg: ca = m, cb = (m ? RQ - m : RQ / 2); sub = i & 3:
        // {ca re, ca im, cb re, cb im}.  Entry = component of
        //   C_r[n, q] = (-1)^r * 0.5 (w + i dw')[n + NT q] * exp(-2 pi i (r q / RQ + r n / nwin)),  q = k + 4 ks.
        const int nt = nt0, rq = rq0, npass = rq / 8, kst = rq / 4;
        p->rq = rq; p->nt = nt;
        const int atab_floats = hssfsst::core128_atab_floats(rq, nt);
        std::vector<float> at(atab_floats + 6 * 64);      // A table, then the FAST epilogue's store offsets (ints)
        for (int pz = 0; pz < npass; ++pz)
            for (int n = 0; n < nt; ++n)
                for (int ks = 0; ks < kst; ++ks)
                    for (int l = 0; l < 64; ++l) {
                        const int i = l & 15, q = (l >> 4) + 4 * ks;
                        const int gg = i >> 2, sub = i & 3, m = 4 * pz + gg;
                        const int r = (sub < 2) ? m : (m ? rq - m : rq / 2);
                        const double ang = -2.0 * M_PI * (static_cast<double>(r) * q / rq + static_cast<double>(r) * n / nwin);
                        const double c = std::cos(ang), sn = std::sin(ang);
                        const double sg = (r & 1) ? -0.5 : 0.5;
                        const double wv = window[n + nt * q], dv = dwb[n + nt * q];
                        const double re = sg * (wv * c - dv * sn), im = sg * (wv * sn + dv * c);
                        at[((pz * nt + n) * 64 + l) * kst + ks] = static_cast<float>((sub & 1) ? im : re);      // [pass][tap][lane][k-step]: the kernel's LDS layout
                    }
        static_assert(sizeof(int) == sizeof(float), "offset table shares the float buffer");
        {
            int offs[6 * 64];
            hssfsst::core128_store_offsets(p->klo, p->K > 0 ? p->K : 2, offs, rq);
            std::memcpy(at.data() + atab_floats, offs, sizeof(offs));
        }
        e = hipMalloc(reinterpret_cast<void**>(&p->d_atab), at.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->d_atab, at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (p->d_atab) (void)hipFree(p->d_atab);
            (void)hipFree(p->d_ctab); (void)hipFree(p->d_wtab); delete p;
            return fail(HSSFSST_EHIP, "plan_create: A-table upload: %s", hipGetErrorString(e));
        }
        if (nwin == 128) {
            // fsst_canon128.hpp: the same constants C_r[n, q] as pairs of halves c1 + c2, scaled by 2^sc into [2^13, 2^14).
            // Entry (tap n, lane l = (kk, row i), half h): fold term q = kk + 4 (h >> 2), c1 for even h, c2 for odd h
            // (products x1 c1, x1 c2, x2 c1, x2 c2 against the sample record {x1, x1, x2, x2}).
            auto comp = [&](int n, int i, int q) -> double {
                const int gg = i >> 2, sub = i & 3, m = gg;
                const int r = (sub < 2) ? m : (m ? 8 - m : 4);
                const double ang = -2.0 * M_PI * (static_cast<double>(r) * q / 8 + static_cast<double>(r) * n / nwin);
                const double c = std::cos(ang), sn = std::sin(ang);
                const double sg = (r & 1) ? -0.5 : 0.5;
                const double wv = window[n + 16 * q], dv = dwb[n + 16 * q];
                return (sub & 1) ? sg * (wv * sn + dv * c) : sg * (wv * c - dv * sn);
            };
            double cmax = 0.0;
            for (int n = 0; n < 16; ++n) for (int i = 0; i < 16; ++i) for (int q = 0; q < 8; ++q) cmax = std::fmax(cmax, std::fabs(comp(n, i, q)));
            int ex = 0;
            if (cmax > 0.0 && std::isfinite(cmax)) (void)std::frexp(cmax, &ex);          // cmax = f 2^ex, f in [0.5, 1)
            const int sc = 14 - ex;                                                       // cmax 2^sc in [2^13, 2^14)
            const double cs = std::ldexp(1.0, sc);
            // (+ the float64 twiddles of the rounding-tie path, 2 kB: the kernels copy the whole table into LDS)
            std::vector<unsigned short> ht(static_cast<size_t>(hssfsst::kCanonAtabFloats) * 2);
            static_assert(hssfsst::kCanonAtabFloats == 16 * 64 * 4 + 4 * 128, "f16 operand table + 128 {cos, sin} doubles");
            for (int i = 0; i < 128; ++i) {
                const double ang = 2.0 * M_PI * static_cast<double>(i) / 128.0;
                const double cs2[2] = {std::cos(ang), std::sin(ang)};
                std::memcpy(ht.data() + static_cast<size_t>(16) * 64 * 8 + static_cast<size_t>(i) * 8, cs2, sizeof(cs2));
            }
            for (int n = 0; n < 16; ++n)
                for (int l = 0; l < 64; ++l)
                    for (int h = 0; h < 8; ++h) {
                        const int i = l & 15, kk = l >> 4, q = kk + 4 * (h >> 2);
                        const double v = comp(n, i, q) * cs;
                        const _Float16 c1 = static_cast<_Float16>(v);
                        const _Float16 c2 = static_cast<_Float16>(v - static_cast<double>(c1));
                        const _Float16 pick = (h & 1) ? c2 : c1;
                        unsigned short bits;
                        std::memcpy(&bits, &pick, sizeof(bits));
                        ht[(static_cast<size_t>(n) * 64 + l) * 8 + h] = bits;
                    }
            // fsst_canon128.hpp "Offsets": what a frame of ones contributes to the spectra the source stage starts from.
            //   Zc[k] = (-1)^k 0.5 sum_{m inside the signal} (w + i dw')[m] e^{-2 pi i k m / 128}  (x cs: plane units),
            // lane group g holds the classes g and (g ? 8 - g : 4): za[s] = Z[8 s + g], zb[s] = Z[8 s + (g ? 8 - g : 4)], s = 0..15.
            // Frame 0 = interior (every m), 1 + t = the frame of output column t < 64 (m >= 64 - t), 65 + r = the frame r < 63 samples
            // before the end (m <= r + 64).  Layout (fsst_canon128.hpp kCanonZcFloats): interior [g][za[0..15] | zb[0..15] | -];
            // left edge [g][entry][t]; right edge [g][entry][r], r = 63 .. 79 = the interior once more
            std::vector<float> zc(static_cast<size_t>(hssfsst::kCanonZcFloats), 0.0f);
            float* zleft = zc.data() + hssfsst::kCanonYcFrame;
            float* zright = zleft + 4 * 32 * 64 * 2;
            for (int fr = 0; fr < 1 + 64 + 63; ++fr) {
                const int m0 = (fr >= 1 && fr <= 64) ? 64 - (fr - 1) : 0;
                const int m1 = (fr >= 65) ? (fr - 65) + 64 : 127;
                double zr[128], zi[128];
                for (int k = 0; k < 128; ++k) {
                    double re = 0.0, im = 0.0;
                    for (int m = m0; m <= m1; ++m) {
                        const double ang = -2.0 * M_PI * static_cast<double>((k * m) % 128) / 128.0;
                        const double c = std::cos(ang), sn = std::sin(ang);
                        re += window[m] * c - dwb[m] * sn;
                        im += window[m] * sn + dwb[m] * c;
                    }
                    const double sg = (k & 1) ? -0.5 : 0.5;
                    zr[k] = sg * re * cs; zi[k] = sg * im * cs;
                }
                for (int gg = 0; gg < 4; ++gg)
                    for (int s16 = 0; s16 < 16; ++s16) {
                        const int ks[2] = {8 * s16 + gg, 8 * s16 + (gg ? 8 - gg : 4)};       // the lane group's two classes: za[s], zb[s]
                        for (int ab = 0; ab < 2; ++ab) {
                            const int ent = ab * 16 + s16;
                            const float vr = static_cast<float>(zr[ks[ab]]), vi = static_cast<float>(zi[ks[ab]]);
                            auto put = [&](float* e) { e[0] = vr; e[1] = vi; };
                            if (fr == 0) {
                                put(zc.data() + (gg * hssfsst::kCanonYcGroup + ent) * 2);
                                for (int r = 63; r < hssfsst::kCanonYcRight; ++r) put(zright + ((gg * 32 + ent) * hssfsst::kCanonYcRight + r) * 2);
                            } else if (fr <= 64) put(zleft + ((gg * 32 + ent) * 64 + (fr - 1)) * 2);
                            else put(zright + ((gg * 32 + ent) * hssfsst::kCanonYcRight + (fr - 65)) * 2);
                        }
                    }
            }
            p->canon_inv_c = static_cast<float>(std::ldexp(1.0, -sc));
            p->canon_r2s = static_cast<float>(static_cast<double>(p->r2scale) * cs * cs);
            e = hipMalloc(reinterpret_cast<void**>(&p->d_atab16), ht.size() * sizeof(unsigned short) + zc.size() * sizeof(float));
            if (e == hipSuccess) e = hipMemcpy(p->d_atab16, ht.data(), ht.size() * sizeof(unsigned short), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->d_atab16 + hssfsst::kCanonAtabFloats, zc.data(), zc.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (p->d_atab16) (void)hipFree(p->d_atab16);
                (void)hipFree(p->d_atab); (void)hipFree(p->d_ctab); (void)hipFree(p->d_wtab); delete p;
                return fail(HSSFSST_EHIP, "plan_create: f16 A-table upload: %s", hipGetErrorString(e));
            }
        }
    }
    *out = p;
    return 0;
}

int hssfsst_plan_destroy(hssfsst_plan* p)
{
    if (!p) return 0;
    DeviceGuard device_guard_(p->device);
    if (p->d_ctab) (void)hipFree(p->d_ctab);
    if (p->d_wtab) (void)hipFree(p->d_wtab);
    if (p->d_dtab) (void)hipFree(p->d_dtab);
    if (p->d_atab) (void)hipFree(p->d_atab);
    if (p->d_atab16) (void)hipFree(p->d_atab16);
    if (p->d_stream_arrive) (void)hipFree(p->d_stream_arrive);
    if (p->d_stream_pieces) (void)hipFree(p->d_stream_pieces);
    if (p->d_partials) (void)hipFree(p->d_partials);
    if (p->h_status) (void)hipHostFree(const_cast<unsigned*>(p->h_status));
    if (p->d_mail) (void)hipFree(p->d_mail);
    if (p->d_arrive) (void)hipFree(p->d_arrive);
    if (p->h_fallback) (void)hipHostFree(const_cast<unsigned*>(p->h_fallback));
    if (p->d_stats) (void)hipFree(p->d_stats);
    if (p->d_xstage) (void)hipFree(p->d_xstage);
    if (p->d_ostage) (void)hipFree(p->d_ostage);
    if (p->h_xpin) (void)hipHostFree(p->h_xpin);
    if (p->h_opin) (void)hipHostFree(p->h_opin);
    for (auto& b : p->pin_pool) if (b.h) (void)hipHostFree(b.h);
    if (p->d_starts) (void)hipFree(p->d_starts);
    if (p->d_frames) (void)hipFree(p->d_frames);
    for (auto& ev : p->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : p->sync_ev) if (ev) (void)hipEventDestroy(ev);
    if (p->aux) (void)hipStreamDestroy(p->aux);
    delete p;
    return 0;
}

int hssfsst_plan_info(const hssfsst_plan* p, int* nwin, int* nf, int* klo, int* K,
                      int* ofps, int* mode, int* device)
{
    if (!p) return fail(HSSFSST_EINVAL, "plan_info: plan is NULL");
    if (nwin) *nwin = p->nwin;
    if (nf) *nf = p->nf;
    if (klo) *klo = p->klo;
    if (K) *K = p->K;
    if (ofps) *ofps = out_floats_per_sample(p);
    if (mode) *mode = p->mode;
    if (device) *device = p->device;
    return 0;
}


int hssfsst_plan_last_exec_fused(const hssfsst_plan* p) { return (p && p->last_fused) ? p->last_zpath : 0; }

int hssfsst_plan_last_kernel(const hssfsst_plan* p, char* buf, int len)
{
    if (!p || !buf || len < 1) return fail(HSSFSST_EINVAL, "plan_last_kernel: bad argument");
    snprintf(buf, static_cast<size_t>(len), "%s", p->last_kernel);
    return 0;
}

// ---- hssfsst_allgather: ncclAllGather through dlopen (no link-time dependency on RCCL)
namespace {
struct RcclApi {
    void* handle = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
    bool tried = false;
    std::string why;                                     // why RCCL is not available (dlerror() answers once: kept)
};
RcclApi& rccl_api()
{
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!api.tried) {
        api.tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            const char* e = dlerror();
            if (e && api.why.empty()) api.why = e;
        }
        if (api.handle) api.why = "symbol ncclAllGather missing";
        if (api.handle) {
            api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(api.handle, "ncclAllGather"));
            api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(api.handle, "ncclGetErrorString"));
        }
    }
    return api;
}
}  // namespace

int hssfsst_allgather(const float* sendbuf, float* recvbuf, int64_t count, void* nccl_comm, void* stream, int timeout_ms)
{
    if (!sendbuf || !recvbuf || !nccl_comm || count < 0 || timeout_ms < 0) return fail(HSSFSST_EINVAL, "allgather: bad argument");
    if (count == 0) return 0;
    RcclApi& api = rccl_api();
    if (!api.all_gather) return fail(HSSFSST_EUNSUPPORTED, "allgather: RCCL (librccl.so.1: ncclAllGather) is not available: %s", api.why.empty() ? "dlopen failed" : api.why.c_str());
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr int kNcclFloat32 = 7;                      // ncclFloat (rccl.h)
    const int rc = api.all_gather(sendbuf, recvbuf, static_cast<size_t>(count), kNcclFloat32, nccl_comm, st);
    if (rc != 0) return fail(HSSFSST_EHIP, "allgather: ncclAllGather: %s", api.error_string ? api.error_string(rc) : "error");
    if (timeout_ms == 0) return 0;
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, st);
    const auto t0 = std::chrono::steady_clock::now();
    while (e == hipSuccess) {
        e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) break;
        e = hipSuccess;
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) {
            (void)hipEventDestroy(ev);
            return fail(HSSFSST_EHIP, "allgather: the collective did not complete within %d ms (a rank is missing?); it is still enqueued on the stream and owns both buffers", timeout_ms);
        }
        std::this_thread::yield();
    }
    (void)hipEventDestroy(ev);
    if (e != hipSuccess) return fail(HSSFSST_EHIP, "allgather: %s", hipGetErrorString(e));
    return 0;
}

int hssfsst_plan_fallbacks(hssfsst_plan* p)
{
    if (!p) return fail(HSSFSST_EINVAL, "plan_fallbacks: plan is NULL");
    if (p->h_fallback) {
        const unsigned now = *p->h_fallback;
        if (now != p->seen_fallback) { p->seen_fallback = now; ++p->fallbacks; note_team_outcome(p, true); }
    }
    return p->fallbacks;
}

int hssfsst_plan_set_zpath(hssfsst_plan* p, int zpath)
{
    if (!p || zpath < HSSFSST_ZPATH_AUTO || zpath > HSSFSST_ZPATH_TEAM) return fail(HSSFSST_EINVAL, "plan_set_zpath: bad argument");
    p->zpath_pref = zpath;
    return 0;
}

int hssfsst_plan_check(hssfsst_plan* p)
{
    if (!p) return fail(HSSFSST_EINVAL, "plan_check: plan is NULL");
    if (!p->d_status) return 0;
    DEVICE_SCOPE(p->device);
    HIP_TRY(hipDeviceSynchronize());                     // every exec of this plan has finished
    const unsigned code = *p->h_status;
    if (code != 0) {
        *p->h_status = 0u;
        return fail(HSSFSST_EHIP, "fused z-score: a wait inside the kernel gave up (code %u); results of that exec are invalid", code);
    }
    return 0;
}

int hssfsst_plan_set_timing(hssfsst_plan* p, int enable)
{
    if (!p) return fail(HSSFSST_EINVAL, "plan_set_timing: plan is NULL");
    p->timing_every = enable > 0 ? enable : 0;
    p->timing = 0;
    p->timing_seq = 0;
    p->ev_used = 0;
    p->ev_chunks.clear();
    return 0;
}

int hssfsst_plan_timing(hssfsst_plan* p, float ms_sum[2], int* nexec)
{
    if (!p || !ms_sum || !nexec) return fail(HSSFSST_EINVAL, "plan_timing: bad argument");
    ms_sum[0] = ms_sum[1] = 0.0f;
    *nexec = static_cast<int>(p->ev_chunks.size());
    if (p->ev_used == 0) return 0;
    DEVICE_SCOPE(p->device);
    HIP_TRY(hipEventSynchronize(p->ev[p->ev_used - 1]));
    size_t i = 0;
    for (int nc : p->ev_chunks) {
        if (nc < 0) {                                    // fused exec: (before, after) only
            float a = 0.0f;
            HIP_TRY(hipEventElapsedTime(&a, p->ev[i], p->ev[i + 1]));
            ms_sum[0] += a;
            i += 2;
            continue;
        }
        float core = 0.0f, total = 0.0f;
        for (int c = 0; c < nc; ++c) {
            float a = 0.0f;
            HIP_TRY(hipEventElapsedTime(&a, p->ev[i + 2 * c], p->ev[i + 2 * c + 1]));
            core += a;
        }
        HIP_TRY(hipEventElapsedTime(&total, p->ev[i], p->ev[i + 2 * nc]));
        ms_sum[0] += core;
        ms_sum[1] += total - core;
        i += 2 * static_cast<size_t>(nc) + 1;
    }
    return 0;
}

int hssfsst_exec(hssfsst_plan* p, const float* x, int64_t batch, int n, int x_on_device,
                 float* out, int out_on_device, void* stream)
{
    return hssfsst_exec_cols(p, x, batch, n, 0, n, x_on_device, out, out_on_device, stream);
}

int hssfsst_exec_cols(hssfsst_plan* p, const float* x, int64_t batch, int n, int col0, int ncols, int x_on_device,
                      float* out, int out_on_device, void* stream)
{
    return hssfsst_exec_frames(p, x, batch, n, static_cast<int64_t>(n), col0, ncols, x_on_device, out, out_on_device, stream);
}

// The one exec: `batch` signals of n samples, signal b at x + b * x_stride, or -- d_starts != nullptr (device array) --
// at x + d_starts[b] inside a buffer of x_len samples.  A frame list is first gathered into a dense [batch][n] staging
// buffer (fsst_gather_frames_kernel: 8 kB read + 8 kB written per frame, against 360 kB of output) and then takes the
// same kernels as a dense batch: the transform kernels sit at the 128-VGPR limit of their occupancy and a second
// addressing mode in them cost spilled registers.
// (pin_d != nullptr: `out` is a pinned buffer of the plan's pool and pin_d its device alias -- hssfsst_exec_pinned: the kernels store the
//  features there and nothing is copied; returns 1 when this exec is not one the kernels can write straight to host memory)
static int exec_impl(hssfsst_plan* p, const float* x, int64_t batch, int n, int64_t x_stride, const long long* d_starts,
                     size_t x_len, int col0, int ncols, int x_on_device, float* out, int out_on_device, void* stream, float* pin_d = nullptr)
{
    if (!p || !x || !out || batch < 0 || n < 1 || col0 < 0 || ncols < 1 || col0 > n - ncols || x_stride < 1)
        return fail(HSSFSST_EINVAL, "exec: bad argument (batch=%lld n=%d stride=%lld col0=%d ncols=%d)",
                    static_cast<long long>(batch), n, static_cast<long long>(x_stride), col0, ncols);
    if (batch == 0 || p->K == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEVICE_SCOPE(p->device);
    // a wait inside an EARLIER exec's kernel gave up (pinned status word, no synchronisation): that exec's features are
    // invalid and the caller of a device-output exec has not been told yet -- refuse until the plan is checked
    if (p->h_status && *p->h_status != 0u) {
        const unsigned code = *p->h_status;
        *p->h_status = 0u;
        return fail(HSSFSST_EHIP, "exec: a wait inside a previous exec's z-score kernel gave up (code %u); the results of that "
                    "exec are invalid", code);
    }
    const int ofps = out_floats_per_sample(p);
    const bool use128 = (p->d_atab != nullptr);
    // statistics partials per signal: one per 16-frame group (MFMA kernel) / per 64-frame tile (generic kernel)
    const int fpp = (use128 || p->dft) ? 16 : kTile;
    const int nblk = (ncols + fpp - 1) / fpp;
    const long long nblocks = static_cast<long long>(batch) * nblk;
    if (nblocks > 0x7fffffffLL) return fail(HSSFSST_EINVAL, "exec: batch*tiles = %lld exceeds the grid limit; split the batch", nblocks);
    if (static_cast<long long>(n) * 2 * p->nf >= 0x7fffffffLL) return fail(HSSFSST_EINVAL, "exec: signal too long (n = %d)", n);
    // input extent: `batch` signals of n samples whose starts are x_stride apart (they may overlap)
    const size_t nx = d_starts ? x_len : static_cast<size_t>(batch > 0 ? batch - 1 : 0) * static_cast<size_t>(x_stride) + n;
    const size_t no = static_cast<size_t>(batch) * ncols * ofps;

    const float* dx = x;
    float* dout = out;
    int rc;
    // Small host-to-host execs -- the reference's dataset loop calls the transform once per 2000-sample frame with CPU tensors
    // (/root/reference/hss/datasets/heart_sounds.py:166-168,199-201) -- do not go through hipMemcpyAsync from / to pageable memory
    // (two staged copies by the runtime, ~0.03 ms of a 0.068 ms call): the samples are copied into a pinned, device-mapped buffer
    // that the kernels read in place, and where the output is written exactly once (every mode but a STACK whose z-score is a
    // second pass over the features) the kernels store it into a pinned buffer too -- the mechanism of hssfsst_stream_step.
    const bool tiny_in = !x_on_device && nx <= (static_cast<size_t>(1) << 16);
    // (STACK: only where the team kernel will take the exec -- its features are written once; a z-score that is a second pass would
    //  read and rewrite them in place over PCIe: signals of more than 128 groups keep the device staging buffer + one copy)
    const bool tiny_out = tiny_in && !out_on_device && no <= (static_cast<size_t>(1) << 21) &&
                          (p->mode != HSSFSST_MODE_STACK || (plan_is_canon(p) && (col0 & 15) == 0 && !debug_switches().no_team &&
                                                             p->zpath_pref != HSSFSST_ZPATH_ONE_CU && p->zpath_pref != HSSFSST_ZPATH_TWO_LAUNCH &&
                                                             (p->team_pause == 0 || p->zpath_pref == HSSFSST_ZPATH_TEAM) &&
                                                             (ncols + 15) / 16 <= hssfsst::kFusedMaxGroups));
    auto pin = [&](float** h, float** d, size_t* cap, size_t need) -> int {
        if (*cap >= need) return 0;
        if (*h) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipHostFree(*h)); *h = nullptr; *d = nullptr; *cap = 0; }
        size_t c = 1 << 12;
        while (c < need) c *= 2;
        void* hp = nullptr; void* dp = nullptr;
        HIP_TRY(hipHostMalloc(&hp, c * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
        *h = static_cast<float*>(hp); *d = static_cast<float*>(dp); *cap = c;
        return 0;
    };
    if (tiny_in) {
        if ((rc = pin(&p->h_xpin, &p->d_xpin, &p->xpin_cap, nx)) != 0) return rc;
        std::memcpy(p->h_xpin, x, nx * sizeof(float));
        dx = p->d_xpin;
    } else if (!x_on_device) {
        if ((rc = grow(reinterpret_cast<void**>(&p->d_xstage), &p->xstage_cap, nx, sizeof(float))) != 0) return rc;
        HIP_TRY(hipMemcpyAsync(p->d_xstage, x, nx * sizeof(float), hipMemcpyHostToDevice, st));
        dx = p->d_xstage;
    }
    if (d_starts) {
        const size_t nd = static_cast<size_t>(batch) * n;
        if ((rc = grow(reinterpret_cast<void**>(&p->d_frames), &p->frames_cap, nd, sizeof(float))) != 0) return rc;
        const long long quads = (static_cast<long long>(n) + 3) / 4;
        long long blocks = (static_cast<long long>(batch) * quads + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(hssfsst::fsst_gather_frames_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st,
                           dx, d_starts, p->d_frames, static_cast<long long>(batch), n);
        HIP_TRY(hipGetLastError());
        dx = p->d_frames;
        x_stride = n;
    }
    if (pin_d && !tiny_out) return 1;                    // (not an exec whose features are written once: the caller takes the copying call)
    p->defer_fallback = false;
    if (tiny_out) {
        if (pin_d) dout = pin_d;
        else {
            if ((rc = pin(&p->h_opin, &p->d_opin, &p->opin_cap, no)) != 0) return rc;
            dout = p->d_opin;
        }
        p->defer_fallback = p->zpath_pref != HSSFSST_ZPATH_ONE_CU && !d_starts;
        p->deferred_launch = 0u;
    } else if (!out_on_device) {
        if ((rc = grow(reinterpret_cast<void**>(&p->d_ostage), &p->ostage_cap, no, sizeof(float))) != 0) return rc;
        dout = p->d_ostage;
    }
    if (p->mode == HSSFSST_MODE_STACK)
    {
        if ((rc = grow(reinterpret_cast<void**>(&p->d_partials), &p->partials_cap, static_cast<size_t>(nblocks) * hssfsst::kPartFloats, sizeof(float))) != 0) return rc;
        if ((rc = grow(reinterpret_cast<void**>(&p->d_stats), &p->stats_cap, static_cast<size_t>(batch) * 4, sizeof(float))) != 0) return rc;
    }

    auto next_event = [&](hipEvent_t* out_ev) -> int { return plan_next_event(p, out_ev); };
    int timed_chunks = 0;
    p->timing_closed = false;
    p->timing = (p->timing_every > 0 && (p->timing_seq++ % static_cast<unsigned>(p->timing_every)) == 0u) ? 1 : 0;
    // STACK: core (FP32-issue-bound) then the z-score sweep (HBM-bound, in place).  An optional
    // pipeline (HSSFSST_CHUNKS=k) cuts the batch into k chunks and runs the sweep of chunk i on a side
    // stream while the core of chunk i+1 runs on the caller's stream.  Measured on MI355X
    // (tools/chunk_sweep.sh): the overlap LOSES -- the saturating sweep back-pressures the core's own
    // stores (core 0.25 -> 0.32-0.41 ms per 1024 windows) -- and keeping a chunk inside the 256 MiB
    // Infinity Cache does not speed the sweep up either, so the default is k = 1.
    // Also measured and rejected (git history): fusing the z-score into the
    // core launch -- "last ticket normalises the signal" (write-through stores + one agent acquire:
    // 0.62 ms; with an L2 write-back release per block: 0.99 ms) and "blocks of a signal wait for each
    // other, then normalise their own tiles" (bounded spin + fix-up kernel: 0.58 ms) -- both
    // bit-identical to, and slower than, the two-pass 0.37 ms per 1024 windows.
    const int64_t per = static_cast<int64_t>(ncols) * ofps;
    int64_t nchunks = 1;
    if (p->mode == HSSFSST_MODE_STACK) {
        if (debug_switches().chunks > 0) nchunks = debug_switches().chunks;
        if (nchunks > batch) nchunks = batch;
    }
    const int64_t chunk = (batch + nchunks - 1) / nchunks;
    const bool piped = nchunks > 1;
    // a host-output exec of one team launch: the launch's last wave says "done" in pinned host memory and this call waits for that word
    // instead of synchronising the stream (below)
    p->flag_done = tiny_out && p->defer_fallback && !p->timing && nchunks == 1;
    p->flag_launch = 0u;
    if (piped) {
        if (!p->aux) HIP_TRY(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
        while (static_cast<int64_t>(p->sync_ev.size()) < nchunks + 1) {
            hipEvent_t e2 = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            p->sync_ev.push_back(e2);
        }
    }
    int ci = 0;
    for (int64_t c0 = 0; c0 < batch; c0 += chunk, ++ci) {
        const int64_t cb = (batch - c0 < chunk) ? batch - c0 : chunk;
        const float* cx = dx + c0 * x_stride;
        float* cout = dout + c0 * per;
        hssfsst::CoreParams cp;
        cp.x = cx; cp.out = cout; cp.partials = p->d_partials ? p->d_partials + c0 * nblk * hssfsst::kPartFloats : nullptr; cp.ctab = p->d_ctab;
        cp.n = n; cp.klo = p->klo; cp.K = p->K; cp.mode = p->mode; cp.nblk = nblk; cp.col0 = col0; cp.ncols = ncols; cp.xstride = x_stride;
        cp.wtab = p->d_wtab; cp.twtab = p->d_wtab + 2 * p->nwin; cp.r2scale = p->r2scale;
        const long long cblocks = static_cast<long long>(cb) * nblk;
        hipEvent_t evt = nullptr;
        if (p->timing) { if ((rc = next_event(&evt)) != 0) return rc; HIP_TRY(hipEventRecord(evt, st)); }
        bool did_fuse = false;
        const bool no_fused = debug_switches().no_fused;  // A/B and bit-equality tests
        if (p->dft) {
            hssfsst::DftParams dp{};
            dp.x = cx; dp.out = cout; dp.partials = cp.partials; dp.atab = p->d_dtab;
            dp.wtab = p->d_wtab; dp.twtab = p->d_wtab + 2 * p->nwin;
            dp.n = n; dp.nwin = p->nwin; dp.nf = p->nf; dp.klo = p->klo; dp.K = p->K; dp.mode = p->mode; dp.col0 = col0; dp.ncols = ncols;
            dp.nk4 = (p->nwin + 3) / 4; dp.nblk4 = (p->nf + 3) / 4;
            dp.nitems = static_cast<long long>(cb) * nblk; dp.xstride = x_stride; dp.r2scale = p->r2scale;
            // groups per work item: 4 when four planes fit the LDS of a wave (each A-operand load then feeds four MFMAs),
            // else 2, else 1; then as many waves per block as fit (at most 8)
            // largest tile that still leaves >= 16 waves resident per CU (the MFMA chains are dependent: latency is hidden
            // by waves, not by the tile), else whatever keeps the most waves (measured: nwin 100 is fastest with small tiles)
            int G = 1, best_waves = -1;
            for (int cand = 4; cand >= 1; cand >>= 1) {
                const size_t pw = static_cast<size_t>(hssfsst::dft_wave_lds_floats(dp.nk4, p->K, cand)) * sizeof(float);
                int w = static_cast<int>(static_cast<size_t>(kMaxLdsBytes) / pw);
                if (w > 8) w = 8;
                if (w < 1) continue;
                int per_cu = static_cast<int>(static_cast<size_t>(kMaxLdsBytes) / (pw * w)) * w;
                if (per_cu > 32) per_cu = 32;
                if (per_cu >= 16) { G = cand; best_waves = per_cu; break; }
                if (per_cu > best_waves) { G = cand; best_waves = per_cu; }
            }
            if (ncols <= 16) G = 1;
            const size_t per_wave = static_cast<size_t>(hssfsst::dft_wave_lds_floats(dp.nk4, p->K, G)) * sizeof(float);
            int waves = static_cast<int>(static_cast<size_t>(kMaxLdsBytes) / per_wave);
            if (waves > 8) waves = 8;
            if (waves < 1) return fail(HSSFSST_EUNSUPPORTED, "exec: LDS request %zu B per wave exceeds 160 KiB", per_wave);
            const int ntiles = (ncols + 16 * G - 1) / (16 * G);
            dp.nitems = static_cast<long long>(cb) * ntiles;
            long long blocks = (dp.nitems + waves - 1) / waves;
            if (blocks > 256 * 64) blocks = 256 * 64;                       // grid-stride beyond that
            auto launch = [&](auto kern) -> int {
                static std::atomic<unsigned long long> lds_ok{0};
                if (int r2 = allow_full_lds(kern, p->device, lds_ok)) return r2;
                name_kernel(p, waves, blocks, "fsst_dft_kernel<%d>", G);
                hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(64 * waves), per_wave * waves, st, dp);
                return (hipGetLastError() == hipSuccess) ? 0 : fail(HSSFSST_EHIP, "exec: fsst_dft_kernel launch failed");
            };
            rc = (G == 4) ? launch(hssfsst::fsst_dft_kernel<4>) : (G == 2) ? launch(hssfsst::fsst_dft_kernel<2>) : launch(hssfsst::fsst_dft_kernel<1>);
        } else if (use128) {
            rc = launch_core128(p, cx, x_stride, cout, cp.partials, n, col0, ncols, cb, st, !no_fused && !piped && p->zpath_pref != HSSFSST_ZPATH_TWO_LAUNCH, &did_fuse);
        } else switch (p->R) {
            case 1: rc = launch_core<1>(p, cp, cblocks, st); break;
            case 2: rc = launch_core<2>(p, cp, cblocks, st); break;
#ifndef HSS_DEV_ONLY128
            case 4: rc = launch_core<4>(p, cp, cblocks, st); break;
            case 8: rc = launch_core<8>(p, cp, cblocks, st); break;
            case 16: rc = launch_core<16>(p, cp, cblocks, st); break;
#endif
            default: rc = fail(HSSFSST_EUNSUPPORTED, "exec: unsupported radix %d", p->R);
        }
        if (rc != 0) return rc;
        if (p->timing) {
            if (p->timing_closed) p->timing_closed = false;
            else { if ((rc = next_event(&evt)) != 0) return rc; HIP_TRY(hipEventRecord(evt, st)); }
            ++timed_chunks;
        }
        const unsigned* gate = p->gate;                  // non-null: a team launch went first; what follows is its gated fallback
        const unsigned gate_val = p->gate_val;
        p->gate = nullptr;
        p->last_fused = (did_fuse || gate != nullptr) ? 1 : 0;
        if (p->mode == HSSFSST_MODE_STACK && !did_fuse) {
            hipStream_t zs = st;
            if (piped) {
                HIP_TRY(hipEventRecord(p->sync_ev[ci], st));
                HIP_TRY(hipStreamWaitEvent(p->aux, p->sync_ev[ci], 0));
                zs = p->aux;
            }
            float4* cstats = reinterpret_cast<float4*>(p->d_stats) + c0;
            const int zgrid_env = debug_switches().zgrid;
            const bool split_stats = debug_switches().split_stats;   // A/B: separate statistics launch
            int64_t zgrid = zgrid_env > 0 ? zgrid_env : (piped ? 512 : 4096);
            // small batches: several blocks per signal, else one block per signal would leave most CUs idle
            int slices = 1;
            const int zslices_env = debug_switches().zslices;
            if (zslices_env > 0) {
                slices = zslices_env;
                zgrid = cb * slices;
            } else if (zgrid_env <= 0 && !piped && cb < 1024) {
                slices = static_cast<int>(1024 / cb);
                if (slices > 32) slices = 32;
            }
            if (zgrid > cb * slices) zgrid = cb * slices;
            // big batches, a block per signal: it reduces the signal's partials itself (no separate statistics
            // launch, 4-7 us per step); otherwise a tiny kernel does all reductions at once
            const bool fused = !split_stats && slices == 1 && zgrid == cb && cb >= 512;
            if (!fused)
                hipLaunchKernelGGL(hssfsst::fsst_stats_kernel, dim3(static_cast<unsigned>(cb)), dim3(64), 0, zs,
                                   cp.partials, cstats, nblk, fpp, ncols, p->K, gate, gate_val);
            hipLaunchKernelGGL(hssfsst::fsst_normalize_kernel, dim3(static_cast<unsigned>(zgrid)), dim3(256), 0, zs,
                               cout, cstats, fused ? cp.partials : nullptr, nblk, fpp, ncols, p->K, static_cast<int>(cb), slices,
                               gate, gate_val);
            HIP_TRY(hipGetLastError());
        }
    }
    if (piped && p->mode == HSSFSST_MODE_STACK) {          // the caller's stream owns the result again
        HIP_TRY(hipEventRecord(p->sync_ev[nchunks], p->aux));
        HIP_TRY(hipStreamWaitEvent(st, p->sync_ev[nchunks], 0));
    }
    if (p->timing) {
        if (p->last_fused && timed_chunks == 1) {
            p->ev_chunks.push_back(-1);                  // one kernel did everything: its two events are the whole exec
        } else {
            hipEvent_t evt = nullptr;
            if ((rc = next_event(&evt)) != 0) return rc;
            HIP_TRY(hipEventRecord(evt, st));
            p->ev_chunks.push_back(timed_chunks);
        }
    }
    if (tiny_out) {
        // The team launch of this exec, if it was asked to (flag_done): its last wave stores the launch's identity to h_fallback[2] behind a
        // system-scope release of every wave's stores -- the features are in pinned host memory by then.  Waiting for that word instead of
        // the stream's completion signal spares the end-of-kernel cache flush and the signal's way to the host: 6 us of a 35 us call
        // (tools/sync_latency.hip).  A launch that gave itself up never stores it: the give-up word ends the wait, as does a bound.
        bool seen = false;
        const unsigned fl = (p->flag_done && p->flag_launch != 0u && p->flag_launch == p->deferred_launch && p->deferred_first == p->deferred_launch)
                            ? p->flag_launch : 0u;
        p->flag_done = false;
        p->flag_launch = 0u;
        if (fl != 0u && p->h_fallback) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0;; ++it) {
                if (p->h_fallback[2] == fl) { seen = true; break; }
                if (p->h_fallback[0] == fl) break;                                   // given up: the usual way below
                if ((it & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;
                __builtin_ia32_pause();
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(st));
            if (fl != 0u) {                               // (the blocks of a given-up launch did not all count themselves in: start the count over)
                HIP_TRY(hipMemsetAsync(p->d_arrive + 2, 0, sizeof(unsigned), st));
                p->done_total = 0;
            }
        }
        p->defer_fallback = false;
        if (p->deferred_launch != 0u) {
            // no gated kernels were queued behind this exec's team launch: if that launch gave itself up (pinned word, written with
            // system scope before the kernel ended), the exec is computed now by the kernels that would have been queued
            const unsigned dl = p->deferred_launch, df = p->deferred_first;
            p->deferred_launch = 0u;
            const unsigned gu = p->h_fallback ? p->h_fallback[0] : 0u;      // (launch identities count up; 0 is never one)
            const bool gave = gu != 0u && (df <= dl ? (gu >= df && gu <= dl) : (gu >= df || gu <= dl));
            note_team_outcome(p, gave);
            if (gave) {
                const int keep = p->zpath_pref;
                p->zpath_pref = HSSFSST_ZPATH_ONE_CU;
                rc = exec_impl(p, x, batch, n, x_stride, d_starts, x_len, col0, ncols, x_on_device, out, out_on_device, stream);      // (copies into `out`: a pool buffer is host memory too)
                p->zpath_pref = keep;
                return rc;
            }
        }
        if (seen) {                                       // (the status word is pinned host memory too: written before the wave that wrote it counted itself in)
            if (p->h_status && *p->h_status != 0u) {
                const unsigned code = *p->h_status;
                *p->h_status = 0u;
                return fail(HSSFSST_EHIP, "fused z-score: a wait inside the kernel gave up (code %u); results of that exec are invalid", code);
            }
        } else if (p->d_status && (rc = hssfsst_plan_check(p)) != 0) return rc;
        if (!pin_d) std::memcpy(out, p->h_opin, no * sizeof(float));
    } else if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out, dout, no * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (p->d_status && (rc = hssfsst_plan_check(p)) != 0) return rc;
    } else if (!x_on_device) {
        HIP_TRY(hipStreamSynchronize(st));   // the host source may be reused by the caller
    }
    return 0;
}

int hssfsst_exec_frames(hssfsst_plan* p, const float* x, int64_t batch, int n, int64_t x_stride, int col0, int ncols,
                        int x_on_device, float* out, int out_on_device, void* stream)
{
    return exec_impl(p, x, batch, n, x_stride, nullptr, 0, col0, ncols, x_on_device, out, out_on_device, stream);
}

// The dataset loop's call without the copy into the caller's tensor: the kernels store the features into a pinned, device-mapped buffer
// of the plan's pool and the caller is LENT that buffer (hssfsst.h).
constexpr size_t kPinPoolMax = 64;
int hssfsst_exec_pinned(hssfsst_plan* p, const float* x, int n, float** out)
{
    if (!p || !x || !out || n < 1) return fail(HSSFSST_EINVAL, "exec_pinned: bad argument");
    *out = nullptr;
    if (p->K == 0) return 1;
    DEVICE_SCOPE(p->device);
    const size_t no = static_cast<size_t>(n) * out_floats_per_sample(p);
    hssfsst_plan::PinBuf* b = nullptr;
    for (auto& c : p->pin_pool) if (!c.used && c.cap >= no) { b = &c; break; }
    if (!b) {
        for (auto& c : p->pin_pool) if (!c.used) { b = &c; break; }       // (a free one that is too small is replaced)
        if (!b) {
            if (p->pin_pool.size() >= kPinPoolMax) return 1;               // every buffer is still in the caller's hands: take the copying call
            p->pin_pool.push_back({nullptr, nullptr, 0, false});
            b = &p->pin_pool.back();
        }
        if (b->h) { HIP_TRY(hipHostFree(b->h)); b->h = nullptr; b->d = nullptr; b->cap = 0; }
        size_t c = 1 << 12;
        while (c < no) c *= 2;
        void* hp = nullptr; void* dp = nullptr;
        HIP_TRY(hipHostMalloc(&hp, c * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&dp, hp, 0));
        b->h = static_cast<float*>(hp); b->d = static_cast<float*>(dp); b->cap = c;
    }
    const int rc = exec_impl(p, x, 1, n, static_cast<int64_t>(n), nullptr, 0, 0, n, 0, b->h, 0, nullptr, b->d);
    if (rc != 0) return rc;
    b->used = true;
    *out = b->h;
    return 0;
}

int hssfsst_pinned_release(hssfsst_plan* p, float* buf)
{
    if (!p || !buf) return fail(HSSFSST_EINVAL, "pinned_release: bad argument");
    for (auto& c : p->pin_pool) if (c.h == buf) { c.used = false; return 0; }
    return fail(HSSFSST_EINVAL, "pinned_release: not a buffer of this plan's pool");
}

int hssfsst_exec_list(hssfsst_plan* p, const float* x, int64_t x_len, const int64_t* starts, int starts_on_device,
                      int64_t batch, int n, int x_on_device, float* out, int out_on_device, void* stream)
{
    if (!p || !x || !starts || !out || batch < 0 || n < 1 || x_len < n)
        return fail(HSSFSST_EINVAL, "exec_list: bad argument (batch=%lld n=%d x_len=%lld)", static_cast<long long>(batch), n,
                    static_cast<long long>(x_len));
    if (batch == 0 || p->K == 0) return 0;
    static_assert(sizeof(long long) == sizeof(int64_t), "frame starts are 64-bit");
    const long long* d_starts = reinterpret_cast<const long long*>(starts);
    if (!starts_on_device) {
        for (int64_t b = 0; b < batch; ++b)
            if (starts[b] < 0 || starts[b] > x_len - n)
                return fail(HSSFSST_EINVAL, "exec_list: frame %lld starts at %lld, outside [0, %lld]", static_cast<long long>(b),
                            static_cast<long long>(starts[b]), static_cast<long long>(x_len - n));
        DEVICE_SCOPE(p->device);
        int rc;
        if ((rc = grow(reinterpret_cast<void**>(&p->d_starts), &p->starts_cap, static_cast<size_t>(batch), sizeof(long long))) != 0) return rc;
        HIP_TRY(hipMemcpyAsync(p->d_starts, starts, static_cast<size_t>(batch) * sizeof(long long), hipMemcpyHostToDevice,
                               static_cast<hipStream_t>(stream)));
        d_starts = p->d_starts;
    }
    return exec_impl(p, x, batch, n, 1, d_starts, static_cast<size_t>(x_len), 0, n, x_on_device, out, out_on_device, stream);
}

int hssfsst_moments_merge(hssfsst_plan* p, const float* feats, int64_t batch, int n, double* state, void* stream)
{
    if (!p || !feats || !state || batch < 0 || n < 1) return fail(HSSFSST_EINVAL, "moments_merge: bad argument");
    if (batch == 0 || p->K == 0) return 0;
    if (batch > 0x7fffffffLL || static_cast<long long>(n) * 2 * p->K >= 0x7fffffffLL) return fail(HSSFSST_EINVAL, "moments_merge: too large");
    DEVICE_SCOPE(p->device);
    hipLaunchKernelGGL(hssfsst::fsst_moments_merge_kernel, dim3(static_cast<unsigned>(batch)), dim3(hssfsst::kMomThreads), 0,
                       static_cast<hipStream_t>(stream), feats, state, n, p->K);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hssfsst_resample(const double* x, int64_t n, int64_t num, double* y)
{
    if (!x || !y || n < 1 || num < 1) return fail(HSSFSST_EINVAL, "hssfsst_resample: bad argument (n=%lld num=%lld)",
                                                   static_cast<long long>(n), static_cast<long long>(num));
    try {
        if (!hssfsst::fourier_resample(x, n, num, y)) return fail(HSSFSST_EINVAL, "hssfsst_resample: bad argument");
    } catch (const std::bad_alloc&) {
        return fail(HSSFSST_ENOMEM, "hssfsst_resample: out of host memory");
    }
    return 0;
}

int64_t hssfsst_pack_recordings(const float* const* ptrs, const int64_t* lens, int64_t count, int stride, int n,
                                float* stage, int64_t stage_cap, int64_t* starts, int64_t starts_cap, int threads)
{
    if (!ptrs || !lens || !stage || !starts || count < 0 || stride < 1 || n < 1)
        return fail(HSSFSST_EINVAL, "pack_recordings: bad argument");
    std::vector<int64_t> pos(static_cast<size_t>(count) + 1, 0);
    int64_t nf = 0;
    for (int64_t i = 0; i < count; ++i) {
        const int64_t T = lens[i];
        if (T < 0 || !ptrs[i]) return fail(HSSFSST_EINVAL, "pack_recordings: recording %lld is NULL or negative", static_cast<long long>(i));
        pos[i + 1] = pos[i] + T;
        // frame_signal: L = floor((T - n) / stride) frames, one fewer than fit; L <= 0 -> the single frame x[:n]
        int64_t L = (T - n >= 0) ? (T - n) / stride : -1;
        if (L <= 0) L = 1;
        if (nf + L > starts_cap) return fail(HSSFSST_EINVAL, "pack_recordings: more than %lld frames", static_cast<long long>(starts_cap));
        for (int64_t k = 0; k < L; ++k) starts[nf + k] = pos[i] + k * stride;
        nf += L;
    }
    const int64_t total = pos[count];
    if (total > stage_cap) return fail(HSSFSST_EINVAL, "pack_recordings: %lld samples exceed the staging capacity %lld",
                                       static_cast<long long>(total), static_cast<long long>(stage_cap));
    int nt = threads > 0 ? threads : static_cast<int>(total / (1 << 20)) + 1;     // one thread per 4 MB of float32
    if (nt > 8) nt = 8;
    if (nt > count) nt = static_cast<int>(count > 0 ? count : 1);
    auto work = [&](int t) {
        // thread t copies the recordings whose first sample falls into its share of the staging buffer
        const int64_t lo = total * t / nt, hi = total * (t + 1) / nt;
        for (int64_t i = 0; i < count; ++i)
            if (pos[i] >= lo && pos[i] < hi && lens[i] > 0) std::memcpy(stage + pos[i], ptrs[i], static_cast<size_t>(lens[i]) * sizeof(float));
    };
    if (nt <= 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    return nf;
}

int64_t hssfsst_parse_signal_csv(const char* text, int64_t len, float* signals, int64_t* labels, int64_t cap)
{
    if (!text || len < 0 || cap < 0 || (cap > 0 && (!signals || !labels))) return fail(HSSFSST_EINVAL, "parse_signal_csv: bad argument");
    const char* p = text;
    const char* end = text + len;
    while (p < end && *p != '\n') ++p;                  // skiprows=1
    if (p < end) ++p;
    int64_t rows = 0;
    while (p < end) {
        const char* eol = p;
        while (eol < end && *eol != '\n') ++eol;
        const char* q = p;
        while (q < eol && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
        if (q < eol) {                                  // non-empty line
            char* stop = nullptr;
            const double sig = std::strtod(q, &stop);
            if (stop == q || stop >= eol || *stop != ',')
                return fail(HSSFSST_EINVAL, "parse_signal_csv: malformed row %lld", static_cast<long long>(rows + 2));
            const char* l0 = stop + 1;
            const double lab = std::strtod(l0, &stop);
            if (stop == l0) return fail(HSSFSST_EINVAL, "parse_signal_csv: malformed label in row %lld", static_cast<long long>(rows + 2));
            if (rows < cap) { signals[rows] = static_cast<float>(sig); labels[rows] = static_cast<int64_t>(lab); }
            ++rows;
        }
        p = (eol < end) ? eol + 1 : end;
    }
    return rows;
}

int hssfsst_normalize_running(hssfsst_plan* p, float* feats, int64_t batch, int n, const double* state, void* stream)
{
    if (!p || !feats || !state || batch < 0 || n < 1) return fail(HSSFSST_EINVAL, "normalize_running: bad argument");
    if (batch == 0 || p->K == 0) return 0;
    if (batch > 0x7fffffffLL || static_cast<long long>(n) * 2 * p->K >= 0x7fffffffLL) return fail(HSSFSST_EINVAL, "normalize_running: too large");
    DEVICE_SCOPE(p->device);
    int rc;
    if ((rc = grow(reinterpret_cast<void**>(&p->d_stats), &p->stats_cap, static_cast<size_t>(batch) * 4, sizeof(float))) != 0) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float4* stats = reinterpret_cast<float4*>(p->d_stats);
    hipLaunchKernelGGL(hssfsst::fsst_stats_from_state_kernel, dim3(static_cast<unsigned>((batch + 63) / 64)), dim3(64), 0, st,
                       state, stats, static_cast<int>(batch));
    const int64_t zgrid = batch < 4096 ? batch : 4096;
    hipLaunchKernelGGL(hssfsst::fsst_normalize_kernel, dim3(static_cast<unsigned>(zgrid)), dim3(256), 0, st,
                       feats, stats, static_cast<const float*>(nullptr), 0, 0, n, p->K, static_cast<int>(batch), 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int hssfsst_stream_step(hssfsst_plan* p, float* tape, int64_t tape_len, int64_t pos, const float* x_new, int64_t x_stride,
                        int x_on_device, int channels, int chunk, float* out, double* state, float* out_host,
                        void* stream)
{
    if (!p || !tape || !x_new || !out || channels < 1 || chunk < 1 || x_stride < chunk)
        return fail(HSSFSST_EINVAL, "stream_step: bad argument");
    if (p->mode != HSSFSST_MODE_STACK_UNNORM) return fail(HSSFSST_EINVAL, "stream_step: the plan must be STACK_UNNORM");
    const int64_t hist = p->nwin - 1;
    if (pos < hist || pos + chunk > tape_len)
        return fail(HSSFSST_EINVAL, "stream_step: pos %lld outside [nwin - 1, tape_len - chunk] (tape_len %lld, chunk %d)",
                    static_cast<long long>(pos), static_cast<long long>(tape_len), chunk);
    if (hist + chunk > 0x7fffffffLL || static_cast<long long>(chunk) * 2 * p->K >= 0x7fffffffLL)
        return fail(HSSFSST_EINVAL, "stream_step: chunk too large");
    if (p->K == 0) return 0;
    if (p->h_status && *p->h_status != 0u) {             // an earlier step's wait between blocks gave up (see hssfsst_plan_check)
        const unsigned code = *p->h_status;
        *p->h_status = 0u;
        // (a step that gave up may have left its channels' arrival counters short of a full round: later steps would never
        //  normalise -- start them from zero again)
        if (p->d_stream_arrive && p->stream_arrive_cap > 0) {
            DEVICE_SCOPE(p->device);
            (void)hipMemsetAsync(p->d_stream_arrive, 0, static_cast<size_t>(p->stream_arrive_cap) * sizeof(unsigned), static_cast<hipStream_t>(stream));
        }
        return fail(HSSFSST_EHIP, "stream_step: an earlier step gave up waiting inside its launch (code %u); its output is invalid", code);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    DEVICE_SCOPE(p->device);
    // one launch for the whole step where the transform is the wide-store MFMA kernel in one-group chunks (nwin 256 / 512, an even
    // band of <= 24 rows: BASELINE config 5); host samples are copied into the tape first and the kernel reads them there
    const bool one_launch = p->d_atab != nullptr && p->rq == 16 && (p->K & 1) == 0 && p->K <= 24 && !debug_switches().no_stream_fuse &&
                            (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                            static_cast<long long>(channels) * ((chunk + 15) / 16) < 0x7fffffffLL;
    int launched = 0;
    float* mirror = nullptr;
    if (one_launch) {
        // host samples in PINNED memory are read by the kernel where they lie (32 KiB per step over the link: no copy operation
        // in front of the launch); pageable memory is copied into the tape first and the kernel reads it there
        const float* xd = x_on_device ? x_new : nullptr;
        if (!x_on_device) {
            // (asked at every step: a cached answer would outlive the caller's buffer)
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, const_cast<float*>(x_new), 0) == hipSuccess && dp) xd = static_cast<const float*>(dp);
            else (void)hipGetLastError();
            if (!xd)
                HIP_TRY(hipMemcpy2DAsync(tape + pos, static_cast<size_t>(tape_len) * sizeof(float), x_new, static_cast<size_t>(x_stride) * sizeof(float),
                                         static_cast<size_t>(chunk) * sizeof(float), static_cast<size_t>(channels), hipMemcpyHostToDevice, st));
        }
        // a pinned host destination is written by the kernel itself (the last block of a channel stores the normalised chunk to
        // both places): no copy operation behind the launch either
        if (out_host && (reinterpret_cast<uintptr_t>(out_host) & 15) == 0) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, out_host, 0) == hipSuccess && dp) mirror = static_cast<float*>(dp);
            else (void)hipGetLastError();
        }
        // (wave pairs: the step's latency is one group's; regions of a block = 2)
        if (debug_switches().no_pair)
            launched = (p->nt == 32) ? launch_stream<32, 16, 4, false>(p, tape + (pos - hist), tape_len, xd, x_stride, channels, chunk, out, state, mirror, st)
                                     : launch_stream<16, 16, 4, false>(p, tape + (pos - hist), tape_len, xd, x_stride, channels, chunk, out, state, mirror, st);
        else
            launched = (p->nt == 32) ? launch_stream<32, 16, 4, true>(p, tape + (pos - hist), tape_len, xd, x_stride, channels, chunk, out, state, mirror, st)
                                     : launch_stream<16, 16, 4, true>(p, tape + (pos - hist), tape_len, xd, x_stride, channels, chunk, out, state, mirror, st);
        if (launched < 0) return launched;
        if (launched == 0 && xd)                         // (not this kernel's shape after all: the chunk goes into the tape by a copy)
            HIP_TRY(hipMemcpy2DAsync(tape + pos, static_cast<size_t>(tape_len) * sizeof(float), x_new, static_cast<size_t>(x_stride) * sizeof(float),
                                     static_cast<size_t>(chunk) * sizeof(float), static_cast<size_t>(channels),
                                     x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    } else {
        HIP_TRY(hipMemcpy2DAsync(tape + pos, static_cast<size_t>(tape_len) * sizeof(float), x_new, static_cast<size_t>(x_stride) * sizeof(float),
                                 static_cast<size_t>(chunk) * sizeof(float), static_cast<size_t>(channels),
                                 x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    if (launched == 0) {
        // frame j of the chunk = samples [pos - hist + j, pos - hist + j + nwin): column nwin/2 + j of the zero-padded
        // transform of the last hist + chunk samples
        int rc = hssfsst_exec_frames(p, tape + (pos - hist), channels, static_cast<int>(hist + chunk), tape_len, p->nwin / 2, chunk, 1, out, 1, stream);
        if (rc != 0) return rc;
        if (state) {
            hipLaunchKernelGGL(hssfsst::fsst_stream_finish_kernel, dim3(static_cast<unsigned>(channels)), dim3(hssfsst::kMomThreads), 0, st, out, state, chunk, p->K);
            HIP_TRY(hipGetLastError());
        }
    }
    if (out_host && launched == 1 && mirror) {
        HIP_TRY(hipStreamSynchronize(st));
    } else if (out_host) {
        HIP_TRY(hipMemcpyAsync(out_host, out, static_cast<size_t>(channels) * chunk * 2 * p->K * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
