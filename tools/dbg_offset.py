import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import oracle
from heart_sounds_segmentation_amd import FSST
from scipy.signal import get_window
fs, n = 1000.0, 1536
t = np.arange(n) / fs
x = (100.0 + np.cos(2 * np.pi * 80.0 * t)).astype(np.float32)[None]
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w = get_window("hann", nwin, fftbins=False)
tf = FSST(fs, w, truncate_freq=None, stack=True)
got = tf.batch(torch.from_numpy(x).cuda()).cpu().numpy()[0]
print("kernel:", tf.last_kernel())
ref = oracle.features(x, fs, w, None, "stack")[0]
K = got.shape[1] // 2
d = got - ref
print("max|ref|", np.abs(ref).max(), "rel L2", np.linalg.norm(d) / np.linalg.norm(ref))
for name, sl in (("re", slice(0, K)), ("im", slice(K, 2 * K))):
    dd = d[:, sl]
    i = np.unravel_index(np.abs(dd).argmax(), dd.shape)
    print(name, "max err", np.abs(dd).max(), "at col", i[0], "row", i[1], "ref there", ref[:, sl][i], "block max", np.abs(ref[:, sl]).max(), "L2", np.linalg.norm(dd) / np.linalg.norm(ref[:, sl]))
    # error by row
    er = np.abs(dd).max(axis=0)
    print("   worst rows", np.argsort(er)[-5:], er[np.argsort(er)[-5:]])
    ec = np.abs(dd).max(axis=1)
    print("   worst cols", np.argsort(ec)[-5:], ec[np.argsort(ec)[-5:]])
raw = tf.unnormalized(torch.from_numpy(x).cuda()).cpu().numpy()[0]
tfr = FSST(fs, w, truncate_freq=None)
rr = oracle.features(x, fs, w, None, "raw")[0]   # (K, n) complex
un = np.concatenate([rr.real.T, rr.imag.T], axis=1)
du = raw - un
print("unnormalised: max err re", np.abs(du[:, :K]).max(), "im", np.abs(du[:, K:]).max(), "max |im|", np.abs(un[:, K:]).max(), "std im", un[:, K:].std(), "std re", un[:, :K].std())
i = np.unravel_index(np.abs(du[:, K:]).argmax(), du[:, K:].shape); print("  im worst at col", i[0], "row", i[1], raw[:, K:][i], un[:, K:][i])
