"""CPU restatement of the reference's wav -> log-mel front-end.  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing else); the product path is the HIP front-end behind ge2e_mel_spectrogram.

Follows reference meldataset.py:73-96 (`mel_spectrogram`, called from Inference.py:71-81 and
Pattern_Generator.py:96-106):  reflect-pad (n_fft - hop)/2 samples on both sides, torch.stft(center=False) with a
periodic Hann window of win_size, magnitude sqrt(re^2 + im^2 + 1e-9), librosa mel filterbank (librosa.filters.mel
defaults: Slaney scale, Slaney area normalisation, fmax = sr/2 when None), log(clamp(x, 1e-5)).

Pinning: the STFT half is checked against torch.stft -- the very call the reference makes (tests/test_mel_frontend.py).
The filterbank is the published librosa/Slaney construction restated from its documentation; librosa is not installed
here and the reference ships no mel fixtures, so **the filterbank values are parity-unpinned** (only their defining
properties are tested).
"""
import numpy as np


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """[n_mels, n_fft // 2 + 1] float32: triangular filters on the Slaney mel scale, each scaled by 2 / bandwidth."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def hann_periodic(n):
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)     # torch.hann_window default


def stft_magnitude(y, n_fft, hop, win):
    """y: [B, L] float -> [B, n_fft // 2 + 1, frames] float32 (meldataset.py:84-91)."""
    assert win == n_fft, "the reference's configuration (Frame_Length == N_FFT); torch.stft would centre a shorter window"
    y = np.asarray(y, dtype=np.float32)
    pad = (n_fft - hop) // 2
    yp = np.pad(y, ((0, 0), (pad, pad)), mode="reflect")
    frames = (yp.shape[1] - n_fft) // hop + 1
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    seg = yp[:, idx].astype(np.float64) * hann_periodic(win).astype(np.float64)[None, None, :]
    spec = np.fft.rfft(seg, n=n_fft, axis=-1)
    mag = np.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9)
    return np.transpose(mag, (0, 2, 1)).astype(np.float32)


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin=0.0, fmax=None):
    """[B, L] -> log-mel [B, num_mels, frames] float32 (meldataset.py:73-96)."""
    mag = stft_magnitude(y, n_fft, hop_size, win_size).astype(np.float64)
    mel = np.einsum("mk,bkf->bmf", mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax).astype(np.float64), mag)
    return np.log(np.clip(mel, 1e-5, None)).astype(np.float32)
