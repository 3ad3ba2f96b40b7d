/*
 * hssfsst.h -- C ABI of libhssfsst.so: the MI355X (gfx950) Fourier-synchrosqueezed-transform
 * feature path.  This is the drop-in boundary for the ONE native dependency of the reference's hot
 * path: the CPython extension `ssq` (-> libssq -> FFTW) that
 *     /root/reference/hss/transforms/synchrosqueeze.py:48   s, f, t = ssq.fsst(x.numpy(), fs, window)
 * binds, plus the tensor epilogue of the same file (:50-111) which the library fuses on the device.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative HSSFSST_E* code and
 *     never throws; hssfsst_last_error() returns a thread-local message for the last failure;
 *   - the caller owns every data buffer; a plan is an opaque handle owned by the library;
 *   - there is NO CPU compute path in this library: creating a plan needs a HIP device and fails
 *     with HSSFSST_ENODEVICE otherwise (the CPU restatement lives in oracle/, test-only);
 *   - HIP is touched for the first time inside hssfsst_plan_create(), in the calling process
 *     (fork-safe for DataLoader workers: create the plan after the fork);
 *   - a plan is SINGLE-STREAM: its execs must be queued on one stream at a time (scratch buffers, the team kernel's arrival
 *     numbers and mailboxes belong to the plan); use one plan per stream.
 *   - hssfsst_exec() enqueues on `stream` (a hipStream_t, NULL = default stream) and returns
 *     without synchronising when both buffers are device buffers; with a host buffer on either
 *     side it stages through the plan's device scratch and synchronises before returning;
 *   - one plan may be used from one host thread at a time.
 */
#ifndef HSSFSST_H
#define HSSFSST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSSFSST_VERSION 208

/* status codes */
#define HSSFSST_OK 0
#define HSSFSST_EINVAL (-1)    /* bad argument (NULL, non-positive size, unknown mode ...)        */
#define HSSFSST_ENODEVICE (-2) /* no usable HIP device / runtime                                  */
#define HSSFSST_EUNSUPPORTED (-3) /* configuration beyond the kernels' LDS budget (huge window x wide band)  */
#define HSSFSST_ENOMEM (-4)    /* host or device allocation failed                                */
#define HSSFSST_EHIP (-5)      /* a HIP call failed (message in hssfsst_last_error)               */

/* output modes == the branches of FSST.__call__ (synchrosqueeze.py:59-65) */
#define HSSFSST_MODE_RAW 0   /* :65     complex64 (K, n), frequency-major, interleaved re/im      */
#define HSSFSST_MODE_ABS 1   /* :59-60  float32 (n, K)   = abs(s).t()                             */
#define HSSFSST_MODE_STACK 2 /* :62-63  float32 (n, 2K)  = z-scored [real | imag], transposed      */
#define HSSFSST_MODE_STACK_UNNORM 3 /* extension (streaming): as STACK but without the z-score    */

typedef struct hssfsst_plan hssfsst_plan;

/* Replaces FSST.__init__ (synchrosqueeze.py:13-35) + everything of ssq.fsst that depends only on
 * (fs, window): derivative window (not-a-knot spline), class-folded twiddle tables, the kept band
 * of _truncate_frequencies (synchrosqueeze.py:91-111; inclusive bounds compared in float32).
 *   device     HIP device ordinal (>= 0)
 *   window     nwin doubles (the analysis window; nfft = nwin).  Any length, odd ones included, as the reference
 *              (synchrosqueeze.py:48): 32 / 64 / 128 / 256 / 512 run the radix kernels, every other length the
 *              any-length kernel (windowed DFT on the fp32 matrix pipe, csrc/fsst_dft.hpp)
 *   has_band   0: keep all nwin/2+1 rows; 1: keep rows with f_lo <= k*fs/nwin <= f_hi
 *   mode       HSSFSST_MODE_* */
int hssfsst_plan_create(hssfsst_plan** out, int device, int nwin, const double* window, double fs,
                        int has_band, double f_lo, double f_hi, int mode);
int hssfsst_plan_destroy(hssfsst_plan* plan);

/* Plan geometry: nf = nwin/2+1 one-sided rows, klo = first kept row, K = number of kept rows,
 * out_floats_per_sample = floats written per input sample (RAW 2K, ABS K, STACK 2K).
 * Any output pointer may be NULL. */
int hssfsst_plan_info(const hssfsst_plan* plan, int* nwin, int* nf, int* klo, int* K,
                      int* out_floats_per_sample, int* mode, int* device);

/* Replaces the call ssq.fsst(x, fs, window) (synchrosqueeze.py:48) AND the epilogue :50-65 for
 * `batch` independent signals of `n` samples each (x: float32 [batch][n], contiguous).
 * out: float32, batch * n * out_floats_per_sample elements, laid out per mode (see HSSFSST_MODE_*),
 * each signal's block contiguous.  x_on_device / out_on_device: 1 = device pointer on the plan's
 * device, 0 = host pointer.  stream: hipStream_t or NULL.
 * A call with a host `out` returns when the features are in `out`.  For small execs (the dataset loop's one frame per call) it learns that
 * from a word the kernel's last block stores to pinned host memory behind a system-scope release of all stores, not from the stream:
 * `stream` may still be retiring that launch for a few microseconds after the return (later work on it queues behind, as always). */
int hssfsst_exec(hssfsst_plan* plan, const float* x, int64_t batch, int n, int x_on_device,
                 float* out, int out_on_device, void* stream);

/* Streaming / windowed variant: computes only the output columns (frame centres) col0 .. col0+ncols-1
 * of every signal; `out` then holds batch * ncols * out_floats_per_sample floats and the STACK
 * statistics run over those columns.  hssfsst_exec(...) == hssfsst_exec_cols(..., 0, n, ...).
 * With col0 >= nwin/2 and col0 + ncols <= n - nwin/2 + 1 no frame touches the zero padding, which is
 * what a rolling transform over a ring buffer needs (SURVEY section 8f row 3 / BASELINE config 5). */
int hssfsst_exec_cols(hssfsst_plan* plan, const float* x, int64_t batch, int n, int col0, int ncols,
                      int x_on_device, float* out, int out_on_device, void* stream);

/* Framed variant: the `batch` signals are views of ONE buffer whose starts are x_stride samples apart
 * (x_stride < n: overlapping frames).  Replaces the reference's framing + per-frame transform loop
 * (hss/utils/preprocess.py:40-52 frame_signal -> hss/datasets/heart_sounds.py:166-168): a recording is uploaded once
 * and its stride-1000 / length-2000 frames are read in place.  x spans (batch-1)*x_stride + n floats.
 * hssfsst_exec_cols(...) == hssfsst_exec_frames(..., x_stride = n, ...). */
int hssfsst_exec_frames(hssfsst_plan* plan, const float* x, int64_t batch, int n, int64_t x_stride, int col0, int ncols,
                        int x_on_device, float* out, int out_on_device, void* stream);

/* The reference's dataset loop calls the transform once per 2000-sample CPU frame (hss/datasets/heart_sounds.py:166-168,199-201).
 * hssfsst_exec_pinned is hssfsst_exec for ONE host signal of n samples whose result is not copied: the kernels store the features
 * straight into a pinned, device-mapped buffer of the plan's pool (at most 64 buffers) and *out is LENT that buffer -- n x
 * floats-per-sample floats in the mode's layout, valid until hssfsst_pinned_release(plan, *out) or the plan's destruction.
 * Returns 0; 1 (no error, *out = NULL) when every pool buffer is still lent out or the exec is not one whose features are written
 * exactly once (then call hssfsst_exec); < 0 on error.  The Python class hands the buffer out as the returned tensor's storage and
 * releases it when that tensor dies; a caller that keeps every result (the in-memory dataset) falls back to the copying call
 * after 64 frames. */
int hssfsst_exec_pinned(hssfsst_plan* plan, const float* x, int n, float** out);
int hssfsst_pinned_release(hssfsst_plan* plan, float* buf);

/* The same transform for a LIST of frames of one buffer: signal b = x[starts[b] .. starts[b] + n), 0 <= starts[b] <=
 * x_len - n (checked when `starts` is host memory; a device array is trusted).  This is the batched form of the
 * dataset loop hss/datasets/heart_sounds.py:155-169 over MANY recordings: the recordings sit back to back in `x`
 * (one upload), `starts` lists every frame of every recording (hss/utils/preprocess.py:40-52), one call transforms
 * them all: the frames are gathered into a dense batch on the device (8 kB per 2000-sample frame against 360 kB of
 * features) and take the kernels of hssfsst_exec.  out: [batch][n][...] as hssfsst_exec. */
int hssfsst_exec_list(hssfsst_plan* plan, const float* x, int64_t x_len, const int64_t* starts, int starts_on_device,
                      int64_t batch, int n, int x_on_device, float* out, int out_on_device, void* stream);

/* Device-side health of the plan's asynchronous work.  The single-launch z-score kernels contain waits on other
 * waves (of the same CU: the one-CU-per-signal kernel; of other CUs of a team: the team kernel, see hssfsst_plan_fallbacks).
 * Every wait of the one-CU-per-signal kernel is bounded (2 s) and a give-up is recorded in a status word in pinned host
 * memory instead of hanging the GPU.  The library looks at that
 * word without synchronising at the start of EVERY exec of the plan and returns HSSFSST_EHIP if an earlier exec's wait
 * gave up (that exec's features are invalid), so a caller of device-output execs learns of it at the next call at the
 * latest; host-output execs check before they return; this function waits for the device and checks now. */
int hssfsst_plan_check(hssfsst_plan* plan);

/* Which path the plan's last STACK exec took: 0 = two launches (transform, then statistics + z-score sweep: long
 * signals, other window lengths, odd or wide bands), 1 = the one-CU-per-signal kernel (full batches of 961..2048-sample
 * signals; only when preferred or for bands the team kernel does not take), 2 = the team kernel (the reference's band,
 * signals up to 2048 samples, any batch: features stay in registers until the signal's statistics arrive from the team).
 * All three give bit-identical results. */
int hssfsst_plan_last_exec_fused(const hssfsst_plan* plan);

/* The dispatch made observable: the transform kernel the plan's last exec ran, as
 * "<instantiation> [<waves> waves/block, grid <blocks>]" (NUL-terminated, truncated to len).  The kernel a given
 * (window length, band, mode, n, batch) takes is otherwise only visible in a kernel trace. */
int hssfsst_plan_last_kernel(const hssfsst_plan* plan, char* buf, int len);

/* Preference among those paths for the plan's following STACK execs: HSSFSST_ZPATH_AUTO (default: the fastest that
 * applies), HSSFSST_ZPATH_TWO_LAUNCH, HSSFSST_ZPATH_ONE_CU (else two launches), HSSFSST_ZPATH_TEAM (else two launches).  A
 * path that does not apply to a shape (see above) is never forced.  Results do not depend on the choice. */
#define HSSFSST_ZPATH_AUTO 0
#define HSSFSST_ZPATH_TWO_LAUNCH 1
#define HSSFSST_ZPATH_ONE_CU 2
#define HSSFSST_ZPATH_TEAM 3
int hssfsst_plan_set_zpath(hssfsst_plan* plan, int zpath);

/* The team kernel's blocks wait for each other; when they are kept apart (other processes' kernels on the same GPU, two launches
 * of one plan on different streams) a wait runs out of time (0.5 ms), the launch gives itself up and the exec is computed by the
 * kernels the library queues behind every team launch, gated on exactly that event: same result, no error.  This returns how many
 * distinct team launches of the plan were seen to have fallen back so far (sampled whenever it is called: call it after a
 * synchronisation).  (Signals that ride on an offset no longer count here: since round 5 the team kernel computes them itself.)
 * A give-up costs the launch its wait bound (0.5 ms) before the gated kernels run: when the host sees four give-ups of a plan in a
 * row (host-output execs look after their own synchronisation, this call looks too) the plan's next 256 execs that would take the
 * team kernel take the one-CU-per-signal / two-launch kernels straight away, then the team kernel is tried again
 * (profiles/r06_stress.txt: DataLoader workers sharing one GPU, /root/reference/main.py:202-218). */
int hssfsst_plan_fallbacks(hssfsst_plan* plan);

/* Multi-GPU reassembly of the feature batch -- the one exchange of the path (BASELINE north_star: "an RCCL all-gather over xGMI only
 * to reassemble the feature batch for the LSTM"; the reference itself has no distributed code, main.py:224-230) -- without torch:
 * every rank contributes `count` floats at sendbuf and receives world x count floats, in rank order, at recvbuf (device pointers;
 * sendbuf may be recvbuf + rank x count: in place).  comm: the caller's ncclComm_t (one process per GPU).  RCCL is loaded with
 * dlopen("librccl.so.1") on first use: the library does not link it, and a host without it gets HSSFSST_EUNSUPPORTED here only.
 * timeout_ms > 0: the call waits for the collective on `stream`, at most that long (HSSFSST_EHIP when exceeded: a peer is missing
 * -- every wait of this library is bounded; the collective is then STILL ENQUEUED on `stream` and still owns both buffers: the
 * caller must not reuse or free them before the stream has drained or the communicator was aborted); 0: returns after enqueueing.
 * The call uses the CURRENT HIP device of the calling thread (the communicator's), it does not switch devices. */
int hssfsst_allgather(const float* sendbuf, float* recvbuf, int64_t count, void* nccl_comm, void* stream, int timeout_ms);

/* Per-kernel HIP-event timing on the exec stream (bench.py's roofline leg).  While enabled, every
 * hssfsst_exec records events around each of its core-kernel launches (a STACK exec runs the batch in
 * cache-sized chunks: core, z-score, core, z-score ...) WITHOUT synchronising; enabling resets the
 * record.  enable = n > 1 times every n-th exec only (an event is a packet on the stream: two per exec cost a
 * 0.18 ms exec ~3 us).  hssfsst_plan_timing() synchronises once and returns, over all TIMED execs since enabling:
 * ms_sum[0] = total synchrosqueeze-core kernel time, ms_sum[1] = rest of the exec (z-score kernels,
 * 0 unless STACK), both in milliseconds, and *nexec = number of execs recorded. */
int hssfsst_plan_set_timing(hssfsst_plan* plan, int enable);
int hssfsst_plan_timing(hssfsst_plan* plan, float ms_sum[2], int* nexec);

/* Host helper, no device needed: derivative window of ssq.fsst's IF estimator
 * (dtwin: not-a-knot cubic spline through (1..n, w), analytic derivative at the knots, * fs/2pi). */
int hssfsst_dtwin(const double* window, int nwin, double fs, double* dwindow);

/* Host helper, no device needed: kept band of _truncate_frequencies (synchrosqueeze.py:91-111). */
int hssfsst_band(int nwin, double fs, double f_lo, double f_hi, int* klo, int* K);

/* hss.moments (hss/moments/__init__.py:1-36): scalar running mean and Welford M2 update. */
double hssfsst_update_mean(double m, double x, int64_t k);
double hssfsst_update_variance(double x, double m, double var, int64_t k);

/* Device-side counterpart of hss.moments for feature batches: merges, per signal, the running
 * (count, mean, M2) of the real and imaginary feature blocks with the statistics of a new STACK-less
 * chunk (Chan's pairwise form of the update above).  state: float64 [batch][6] =
 * {count_re, mean_re, M2_re, count_im, mean_im, M2_im} on the device.  feats: float32
 * [batch][n][2K] un-normalised [real | imag] features on the device. */
int hssfsst_moments_merge(hssfsst_plan* plan, const float* feats, int64_t batch, int n,
                          double* state, void* stream);

/* Streaming z-score: normalises un-normalised features IN PLACE with the running statistics in
 * `state` (as maintained by hssfsst_moments_merge): (v - mean) / sqrt(M2 / (count - 1)) per block. */
int hssfsst_normalize_running(hssfsst_plan* plan, float* feats, int64_t batch, int n,
                              const double* state, void* stream);

/* One step of the rolling transform (BASELINE config 5: C channels, `chunk` new samples per channel per step; no
 * counterpart in the reference, built from the same path -- see heart_sounds_segmentation_amd/streaming.py).
 * `tape`: float32 [channels][tape_len] on the device, the caller's sample history; the step
 *   1. copies x_new ([channels] rows of `chunk` samples, x_stride >= chunk samples apart; device memory, or host
 *      memory when x_on_device = 0 -- pinned for an asynchronous copy) to tape[:, pos .. pos + chunk),
 *   2. transforms the `chunk` frames that end at the new samples (frames tape[:, pos - (nwin-1) + j ..], hop 1: no
 *      frame touches padding) into `out` = float32 [channels][chunk][2K] on the device (plan mode STACK_UNNORM),
 *   3. state != NULL: merges the chunk into the running moments `state` (float64 [channels][6], as
 *      hssfsst_moments_merge) and normalises `out` with the UPDATED moments (as hssfsst_normalize_running) in one
 *      launch -- bit-identical to calling the two,
 *   4. out_host != NULL: copies `out` to out_host (pinned host memory) and waits for the stream: when the call
 *      returns the features of this step are on the host (the latency BASELINE config 5 asks for).
 * Requires nwin - 1 <= pos and pos + chunk <= tape_len; the caller moves the history back to the start of the
 * tape when it is full.
 * For window lengths 256 / 512 with an even band of <= 24 rows (config 5) steps 1-4 are ONE kernel launch: x_new in device or
 * PINNED host memory is read by the kernel where it lies (it must stay untouched until the step has run, as for the copy), a
 * pinned, 16-byte aligned out_host is written by the kernel itself; pageable memory is copied.  Same results either way. */
int hssfsst_stream_step(hssfsst_plan* plan, float* tape, int64_t tape_len, int64_t pos, const float* x_new,
                        int64_t x_stride, int x_on_device, int channels, int chunk, float* out, double* state,
                        float* out_host, void* stream);

/* Host helper, no device needed: parser for the corpus files read by DavidSpringerHSS._load_file
 * (hss/datasets/heart_sounds.py:193-197: pd.read_csv(skiprows=1, names=["Signals", "Labels"])):
 * text = the whole file; the first line is skipped; every following non-empty line is
 * "<float>,<number>" (second column read as a number and truncated to int64, as pandas+torch do for
 * integral label columns).  Fills at most `cap` rows; returns the number of data rows in the file
 * (call once with cap = 0 to size the buffers) or a negative status on a malformed line. */
int64_t hssfsst_parse_signal_csv(const char* text, int64_t len, float* signals, int64_t* labels, int64_t cap);

/* Host helper of the batched dataset builder (/root/reference/hss/datasets/heart_sounds.py:155-169 +
 * hss/utils/preprocess.py:30-58, many recordings per call): copies `count` host recordings (float32, contiguous, lens[i]
 * samples at ptrs[i]) back to back into `stage` and writes the start, inside `stage`, of every frame frame_signal would
 * emit for them -- L = floor((T - n) / stride) frames i * stride, or ONE frame at 0 when L <= 0 (the caller has already
 * dropped recordings shorter than n) -- into `starts` (capacity starts_cap).  Uses up to `threads` host threads for the
 * copies (<= 0: one per 4 MB, at most 8).  Returns the number of frames written, or a negative status. */
int64_t hssfsst_pack_recordings(const float* const* ptrs, const int64_t* lens, int64_t count, int stride, int n,
                                float* stage, int64_t stage_cap, int64_t* starts, int64_t starts_cap, int threads);

/* Host helper, no device needed: replaces Resample.__call__ (hss/transforms/resample.py:13-21), i.e.
 * scipy.signal.resample(x, num) for a real 1-D sequence (Fourier method, window=None): y[0..num) from x[0..n).
 * Used by the dataset for the label path (hss/datasets/heart_sounds.py:202-207: round(Resample(y)) - 1). */
int hssfsst_resample(const double* x, int64_t n, int64_t num, double* y);

int hssfsst_device_count(void);
int hssfsst_version(void);
const char* hssfsst_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HSSFSST_H */
