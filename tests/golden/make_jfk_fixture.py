"""Makes tests/golden/jfk_16k_mono.flac: the reference's real-speech fixture (/root/reference/test/jfk.flac, 44.1 kHz stereo
24-bit, 11 s) exactly as this package's loader hands it to the hot path -- decoded by libswx's FLAC decoder (MD5 signature
verified), mixed to mono, resampled to 16 kHz, rounded to the s16 grid (stable_ts_amd.audio_io.load_audio) -- re-encoded
losslessly by the test encoder (tests/flac_encoder.py; FIXED predictors, Rice partitions; ~190 KB).  The GPU box has no
/root/reference: GPU tests and `bench.py --host-audio` read this file.  tests/test_flac_cpu.py checks that loading the two
files gives identical samples.   usage (in the build container): python tests/golden/make_jfk_fixture.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import flac_encoder as fe                      # noqa: E402
from stable_ts_amd import audio_io             # noqa: E402

src = "/root/reference/test/jfk.flac"
y = audio_io.load_audio(src)                   # f32 on the s16 grid, 16 kHz mono
pcm = np.round(y.astype(np.float64) * 32768.0).astype(np.int64)[:, None]
assert np.array_equal((pcm[:, 0] / 32768.0).astype(np.float32), y)
BS = 4096
blocks, specs = [], []
for a in range(0, len(pcm), BS):
    x = pcm[a:a + BS, 0]
    best = None
    for order in range(5):
        if order >= len(x):
            break
        r = np.diff(x, n=order)
        cost = int(np.abs(r).sum())
        if best is None or cost < best[0]:
            best = (cost, order)
    bs = len(x)
    porder = 4 if bs == BS else 0
    blocks.append(bs)
    specs.append([dict(kind="fixed", order=best[1], porder=porder)])
data = fe.encode(pcm, 16000, 16, blocks, specs)
out = os.path.join(HERE, "jfk_16k_mono.flac")
with open(out, "wb") as f:
    f.write(data)
z = audio_io.load_audio(out)
assert np.array_equal(z, y), "round trip"
print(out, len(data), "bytes;", len(y), "samples")
