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
# C3 stand-in on ONE GPU: 792 recordings x 35.5 k samples -> 26 136 windows, device-resident
R = torch.from_numpy(synth.pcg_windows(8, 35500, seed=5)).cuda()
from heart_sounds_segmentation_amd.framing import frame_batch
def corpus():
    for r in range(792):
        tf.batch(frame_batch(R[r % 8], 1000, 2000))
dt = timed(corpus, 2, 1)
res["C3_standin_1gpu"] = {"recordings": 792, "windows": 792 * 33, "seconds": round(dt, 4), "windows_per_s": round(792 * 33 / dt, 1),
                          "note": "one launch per recording (33 windows): launch-bound; bench.py batches 1024"}
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
xs = torch.from_numpy(synth.pcg_windows(64, 128, fs=4000, seed=2)).cuda()
dt = timed(lambda: st.step(xs), 200, 10)
res["C5_streaming_64ch_4kHz_nwin512"] = {"ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1 / dt, 1),
                                          "realtime_factor": round((128 / 4000) / dt, 1), "lookahead_ms": round(255 / 4000 * 1e3, 2)}
st2 = StreamingFSST(64, 4000, get_window(("kaiser", 0.5), 128, fftbins=False), truncate_freq=(25, 200), chunk=128)
dt = timed(lambda: st2.step(xs), 200, 10)
res["C5_streaming_64ch_4kHz_nwin128"] = {"ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1 / dt, 1)}
print(json.dumps(res))
