"""CPU check of the RAGGED log-mel variant's index logic (csrc/swx_mel.hip, swx_log_mel_ragged).

The kernel cannot run here, so its addressing is restated literally in numpy -- per frame t and tap n the sample index
``s = 160 t + n - 200`` reflected about 0 and about n_total, zero outside [0, n_valid); frames below n_total/160 count,
the per-item (or whole-batch) max runs over those frames including the ones past 3000, frames past n_total/160 are 0.0 --
and compared with what upstream computes for refine (alignment.py:660-661, no padding) and locate (alignment.py:924-925,
padding 201): ``pad_or_trim(log_mel_spectrogram(segment, padding=p), 3000)`` on the oracle.  The arithmetic (f64 DFT of
the f32 windowed frame, |.|^2 in f32, f64 filterbank sum) follows the kernel too.  The hardware parity check of the
kernel itself is tests/hw_checks/mel_ragged_check.py.
"""
import numpy as np
import pytest
import torch

from oracle.whisper.audio import N_FRAMES, log_mel_spectrogram, mel_filters, pad_or_trim

HOP, NFFT, FB, BLOCKS = 160, 400, 8, N_FRAMES // 8 + 1


def kernel_restatement(segments, padding, n_mels, per_item_max=True):
    hann = torch.hann_window(NFFT).numpy().astype(np.float32)
    filt = mel_filters("cpu", n_mels).numpy().astype(np.float64)
    B = len(segments)
    out = np.zeros((B, n_mels, N_FRAMES), np.float32)
    gmax = np.full(B, -np.inf, np.float32)
    lens = []
    for b, seg in enumerate(segments):
        x = np.zeros(480000, np.float32)
        n_valid = len(seg)
        x[:n_valid] = seg
        n_total = n_valid + padding
        assert 200 < n_total and n_total // HOP <= 3008 and n_valid <= 480000      # swx_log_mel_ragged's range check
        n_frames = n_total // HOP
        lens.append(n_frames)
        for blk in range(BLOCKS):
            t0 = blk * FB
            if t0 >= n_frames:
                continue
            t = t0 + np.arange(FB)[:, None]
            s = t * HOP + np.arange(NFFT)[None, :] - NFFT // 2
            s = np.where(s < 0, -s, s)
            s = np.where(s >= n_total, 2 * (n_total - 1) - s, s)
            ok = (s >= 0) & (s < n_valid)
            xw = np.where(ok, x[np.clip(s, 0, 479999)] * hann[None, :], np.float32(0)).astype(np.float32)
            spec = np.fft.rfft(xw.astype(np.float64), axis=-1)
            re, im = spec.real.astype(np.float32), spec.imag.astype(np.float32)
            mag = np.sqrt(re * re + im * im, dtype=np.float32)
            pw = (mag * mag).astype(np.float32)
            v = np.log10(np.maximum((pw.astype(np.float64) @ filt.T).astype(np.float32), np.float32(1e-10))).astype(np.float32)
            for f in range(FB):
                if t0 + f >= n_frames:
                    continue
                if t0 + f < N_FRAMES:
                    out[b, :, t0 + f] = v[f]
                gmax[b] = max(gmax[b], v[f].max())
    for b in range(B):
        mx = gmax[b] if per_item_max else gmax.max()
        body = (np.maximum(out[b], mx - np.float32(8.0)) + np.float32(4.0)) / np.float32(4.0)
        keep = np.arange(N_FRAMES)[None, :] < lens[b]
        out[b] = np.where(keep, body, np.float32(0))
    return out


def synth(n, seed):
    g = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * (180 + 40 * seed) * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * g.standard_normal(n)
    return x.astype(np.float32)


@pytest.mark.parametrize("n,padding", [(480000, 201), (480000, 0), (479999, 201), (163217, 201), (163217, 0), (8000, 0),
                                       (1601, 201), (320, 0), (201, 0), (479840, 201), (16000 * 7, 201)])
def test_ragged_restatement_matches_upstream_frames(n, padding):
    seg = synth(n, seed=n % 7)
    want = pad_or_trim(log_mel_spectrogram(torch.from_numpy(seg), 80, padding=padding), N_FRAMES).numpy()
    got = kernel_restatement([seg], padding, 80)[0]
    n_frames = (n + padding) // HOP
    assert np.array_equal(got[:, n_frames:], want[:, n_frames:])            # the 0.0 fill, exactly
    assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()


def test_ragged_max_includes_frames_cut_by_pad_or_trim():
    # a click in the last 100 samples of a full chunk lands mostly in frame 3000, which pad_or_trim cuts after the
    # clamp floor was taken from it: the floor of the quiet body must follow that frame
    seg = 1e-4 * synth(480000, 3)
    seg[-60:] = 0.9
    want = pad_or_trim(log_mel_spectrogram(torch.from_numpy(seg), 80, padding=201), N_FRAMES).numpy()
    got = kernel_restatement([seg], 201, 80)[0]
    assert np.abs(got - want).max() < 2e-4
    # had the floor come from the kept frames only it would sit 2.0 below their max; the cut frame 3000 raised it
    assert want.min() > want.max() - 2.0 + 0.1


def test_ragged_batch_max_is_upstreams_batched_call():
    a, b = synth(52000, 1), 0.01 * synth(52000, 2)
    want = pad_or_trim(log_mel_spectrogram(torch.from_numpy(np.stack([a, b])), 128), N_FRAMES).numpy()
    got = kernel_restatement([a, b], 0, 128, per_item_max=False)
    assert np.abs(got - want).max() < 2e-4
    per_item = kernel_restatement([a, b], 0, 128, per_item_max=True)
    assert np.abs(per_item[1] - want[1]).max() > 0.05                       # the quirk is observable


def test_ragged_range_check_mirrors_torch_stft():
    with pytest.raises(RuntimeError):
        log_mel_spectrogram(torch.zeros(200), 80)                           # reflect pad needs more than n_fft/2 samples
    with pytest.raises(AssertionError):
        kernel_restatement([np.zeros(200, np.float32)], 0, 80)
