#!/usr/bin/env python3
"""Side measurements for the BASELINE.json configs that are NOT the bench line (C1, C3 stand-in on one
GPU, C4 end-to-end, C5 streaming) and the PCIe-inclusive single-window rate.  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heart_sounds_segmentation_amd import FSST, synth
from heart_sounds_segmentation_amd.consumer import SegmenterHead, segment
from heart_sounds_segmentation_amd.corpus import build_features
from heart_sounds_segmentation_amd.streaming import StreamingFSST

def timed(fn, reps, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

w = synth.kaiser_window(128, 0.5)
tf = FSST(1000, w, truncate_freq=(25, 200), stack=True)
res = {}
# C1: one recording through the drop-in call (CPU tensor in/out), whole recording and framed
rec = torch.from_numpy(synth.recording(35500))
res["C1_whole_recording_ms"] = round(timed(lambda: tf(rec), 10) * 1e3, 3)
res["C1_33_frames_batched_ms"] = round(timed(lambda: build_features([(rec, None)], tf), 10) * 1e3, 3)
# PCIe-inclusive single-window calls, as the unchanged dataset loop issues them (heart_sounds.py:166-168)
fr = torch.from_numpy(synth.pcg_windows(1, 2000)[0]).reshape(2000, 1)
dt = min(timed(lambda: tf(fr), 200, 10) for _ in range(3))      # best of 3 rounds of 200 calls
res["pcie_inclusive_single_window"] = {"ms_per_call": round(dt * 1e3, 4), "windows_per_s": round(1 / dt, 1)}
# C3 stand-in on ONE GPU: 792 recordings x 35.5 k samples -> 26 136 windows
R = torch.from_numpy(synth.pcg_windows(8, 35500, seed=5)).cuda()
from heart_sounds_segmentation_amd.framing import frame_batch, frame_starts
def corpus_per_recording():
    for r in range(792):
        tf.batch(frame_batch(R[r % 8], 1000, 2000))
dt1 = timed(corpus_per_recording, 2, 1)
# the same corpus as groups of 124 recordings (4092 windows) laid back to back: one hssfsst_exec_list call per group
GROUP = 124
big = torch.cat([R[r % 8] for r in range(GROUP)])
st0 = torch.from_numpy(frame_starts(35500, 1000, 2000)[0])
starts = torch.cat([st0 + 35500 * r for r in range(GROUP)]).cuda()
outg = torch.empty((GROUP * 33, 2000, 44), dtype=torch.float32, device="cuda")
def corpus_grouped():
    done = 0
    while done < 792:
        g = min(GROUP, 792 - done)
        tf.frames(big[: g * 35500], starts[: g * 33], 2000, out=outg[: g * 33])
        done += g
dt2 = timed(corpus_grouped, 3, 1)
# host recordings -> features on the device through the product's builder (upload of every group included)
recs_host = [(torch.from_numpy(synth.recording(35500, seed=100 + (r % 8))), None) for r in range(792)]
t0 = time.perf_counter(); items = build_features(recs_host, tf, keep_on_device=True); torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
assert len(items) == 792 * 33
del items
res["C3_standin_1gpu"] = {"recordings": 792, "windows": 792 * 33,
                          "per_recording_launches": {"seconds": round(dt1, 4), "windows_per_s": round(792 * 33 / dt1, 1),
                                                     "note": "one launch per recording (33 windows): launch-bound"},
                          "grouped_frame_lists": {"seconds": round(dt2, 4), "windows_per_s": round(792 * 33 / dt2, 1),
                                                  "note": "7 hssfsst_exec_list calls of <= 4092 windows, device-resident"},
                          "build_features_host_to_device": {"seconds": round(dt3, 4), "windows_per_s": round(792 * 33 / dt3, 1),
                                                            "note": "corpus.build_features: host recordings -> device features, "
                                                                    "Python list of 26 136 items included"}}
# C4: 50 windows -> FSST -> BiLSTM(44 -> 2x240 -> 2x240 -> 4) inference
head = SegmenterHead(44, 240, 50).cuda().eval()
X50 = torch.from_numpy(synth.pcg_windows(50, 2000, seed=9)).cuda()
with torch.no_grad():
    y = segment(tf, head, X50)
    assert y.shape == (50, 2000, 4) and torch.isfinite(y).all()
    e2e = timed(lambda: segment(tf, head, X50), 20, 3)
    fonly = timed(lambda: tf.batch(X50), 50, 5)
res["C4_end_to_end_batch50"] = {"ms": round(e2e * 1e3, 3), "windows_per_s": round(50 / e2e, 1), "fsst_only_ms": round(fonly * 1e3, 4)}
# C5: 64 channels x 4 kHz, 128 new samples per step, nwin 512 (same 128 ms window, same 22 bins)
from scipy.signal import get_window
w512 = get_window(("kaiser", 0.5), 512, fftbins=False)
st = StreamingFSST(64, 4000, w512, truncate_freq=(25, 200), chunk=128)
# (a continuous stream cut into chunks, as bench.py --config c5 does: feeding ONE chunk over and over is a signal of period
#  128 samples -- harmonics exactly on every fourth bin, every other cell numerical zero, the float64 tie path's worst case)
xall = torch.from_numpy(synth.pcg_windows(64, 128 * 64, fs=4000, seed=2)).cuda()
_i = [0]
def _next():
    _i[0] += 1
    return xall[:, (_i[0] % 64) * 128:(_i[0] % 64 + 1) * 128]
dt = timed(lambda: st.step(_next()), 200, 10)
res["C5_streaming_64ch_4kHz_nwin512"] = {"ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1 / dt, 1),
                                          "realtime_factor": round((128 / 4000) / dt, 1), "lookahead_ms": round(255 / 4000 * 1e3, 2)}
st2 = StreamingFSST(64, 4000, get_window(("kaiser", 0.5), 128, fftbins=False), truncate_freq=(25, 200), chunk=128)
dt = timed(lambda: st2.step(_next()), 200, 10)
res["C5_streaming_64ch_4kHz_nwin128"] = {"ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1 / dt, 1)}
print(json.dumps(res))
