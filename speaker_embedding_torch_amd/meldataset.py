"""wav -> log-mel front-end on the MI355X, same call surface as the reference's meldataset.py:73-96.

`mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False)`: `y` is a float
tensor [Batch, Samples] on the GPU; the result is the log-mel [Batch, num_mels, Frames] float32 on the GPU, produced
by `ge2e_mel_spectrogram` (framing + Hann window, real DFT as an fp32 MFMA contraction, magnitude, mel filterbank,
log) -- nothing is computed on the host except the filterbank table, built once per geometry.  `win_size` must equal
`n_fft` (the reference's configuration, Hyper_Parameters.yaml Sound.Frame_Length == N_FFT) and `center` must be False
(how the reference calls it).  `load_wav` reads PCM through scipy.io.wavfile exactly as meldataset.py:38-40.

The reference takes its filterbank from librosa.filters.mel (Slaney scale, Slaney normalisation); librosa is not a
dependency here, the same construction is written out below.
"""
import math
import threading

import numpy as np
import torch

from . import _lib

MAX_WAV_VALUE = 32768.0
_basis_cache = {}
_lock = threading.Lock()
_handle = {}


def load_wav(full_path):
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return data, sampling_rate


def _slaney_hz(mel):
    return 200.0 / 3 * mel if mel < 15.0 else 1000.0 * math.exp(math.log(6.4) / 27.0 * (mel - 15.0))


def _slaney_mel(hz):
    return hz / (200.0 / 3) if hz < 1000.0 else 15.0 + math.log(hz / 1000.0) / (math.log(6.4) / 27.0)


def slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin=0.0, fmax=None):
    """[num_mels, n_fft // 2 + 1] float32 -- librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults."""
    fmax = sampling_rate / 2.0 if fmax is None else float(fmax)
    lo, hi = _slaney_mel(float(fmin)), _slaney_mel(fmax)
    edges = [_slaney_hz(lo + (hi - lo) * i / (num_mels + 1)) for i in range(num_mels + 2)]
    bins = n_fft // 2 + 1
    freqs = [sampling_rate / 2.0 * k / (bins - 1) for k in range(bins)]
    basis = np.zeros((num_mels, bins), dtype=np.float64)
    for m in range(num_mels):
        left, centre, right = edges[m], edges[m + 1], edges[m + 2]
        norm = 2.0 / (right - left)
        for k, f in enumerate(freqs):
            if left < f < right:
                basis[m, k] = norm * min((f - left) / (centre - left), (right - f) / (right - centre))
    return basis.astype(np.float32)


def _get_handle(device_index):
    with _lock:
        h = _handle.get(device_index)
        if h is None:             # one handle per device (a handle belongs to the device of its first call)
            h = _handle[device_index] = _lib.Handle(emb=256, heads=4)
        return h


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    if center or win_size != n_fft:
        raise NotImplementedError("mel_spectrogram: only the reference's configuration (center=False, win_size == n_fft)")
    if not (torch.is_tensor(y) and y.is_cuda):
        raise RuntimeError("mel_spectrogram runs on the MI355X: pass a CUDA tensor [Batch, Samples] (no CPU fallback)")
    if y.dim() == 1:
        y = y.unsqueeze(0)
    y = y.contiguous().float()
    key = (sampling_rate, n_fft, num_mels, float(fmin), None if fmax is None else float(fmax), y.device.index)
    with _lock:
        basis = _basis_cache.get(key)
    if basis is None:
        basis = torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)).to(y.device)
        with _lock:
            _basis_cache[key] = basis
    hnd = _get_handle(y.device.index)
    batch, samples = y.shape
    frames = hnd.mel_frames(samples, n_fft, hop_size)
    nbytes = hnd.mel_workspace_bytes(batch, samples, n_fft, hop_size, num_mels)
    if frames <= 0 or nbytes == 0:
        raise RuntimeError(f"mel_spectrogram: unsupported geometry (samples={samples}, n_fft={n_fft}, hop={hop_size}, mels={num_mels})")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=y.device)
    out = torch.empty(batch, num_mels, frames, dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        hnd.mel_spectrogram(torch.cuda.current_stream(y.device).cuda_stream, y, n_fft, hop_size, num_mels, basis, out, ws)
    return out
