"""the wav -> log-mel front-end (reference meldataset.py:73-96).

CPU: the numpy oracle's STFT half against torch.stft -- the call the reference itself makes -- and the defining
properties of the Slaney filterbank (librosa is not installed and the reference ships no mel fixtures: the
filterbank VALUES are parity-unpinned, see oracle/mel_oracle.py).  GPU: the HIP front-end against the oracle, and a
wav file through Inferencer."""
import os

import numpy as np
import pytest
import torch

from oracle import mel_oracle as M
from conftest import rel_l2

SOUND = dict(n_fft=1024, hop=256, win=1024, mels=80, sr=22050)


def _audio(batch, frames, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(frames * SOUND["hop"]) / SOUND["sr"]
    y = np.stack([0.4 * np.sin(2 * np.pi * (110.0 * (b + 1)) * t) + 0.2 * np.sin(2 * np.pi * 1800.0 * t + b)
                  + 0.05 * rng.standard_normal(t.shape) for b in range(batch)])
    return (y / np.abs(y).max() * 0.95).astype(np.float32)


def test_oracle_stft_matches_torch_stft():
    y = _audio(2, 40)
    mag = M.stft_magnitude(y, SOUND["n_fft"], SOUND["hop"], SOUND["win"])
    yt = torch.nn.functional.pad(torch.from_numpy(y).unsqueeze(1), (384, 384), mode="reflect").squeeze(1)   # meldataset.py:84-85
    spec = torch.stft(yt, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    ref = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9).numpy()                                         # meldataset.py:91
    assert mag.shape == ref.shape == (2, 513, 40)
    assert np.abs(mag - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_slaney_filterbank_properties_and_product_table():
    from speaker_embedding_torch_amd.meldataset import slaney_mel_basis
    for sr, n_fft, mels, fmin, fmax in ((22050, 1024, 80, 0.0, None), (16000, 512, 40, 50.0, 7000.0)):
        b = M.mel_basis(sr, n_fft, mels, fmin, fmax)
        assert b.shape == (mels, n_fft // 2 + 1) and (b >= 0).all()
        freqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
        peaks = freqs[b.argmax(1)]
        assert (np.diff(peaks) > 0).all()                                   # centres rise monotonically
        # Slaney area normalisation: every triangle integrates to ~1 over Hz (exactly for a continuous triangle)
        area = b.sum(1) * (freqs[1] - freqs[0])
        assert np.abs(area[5:] - 1.0).max() < 0.12
        # below 1 kHz the scale is linear: equal spacing of the centres
        low = peaks[peaks < 900.0]
        assert len(low) > 3 and np.abs(np.diff(low) - np.diff(low).mean()).max() <= (freqs[1] - freqs[0]) + 1e-6
        assert np.abs(slaney_mel_basis(sr, n_fft, mels, fmin, fmax) - b).max() < 1e-7      # the product's own table


@pytest.mark.gpu
@pytest.mark.parametrize("batch,frames", [(3, 37), (1, 4), (5, 160)])
def test_hip_mel_spectrogram_matches_oracle(batch, frames):
    from speaker_embedding_torch_amd.meldataset import mel_spectrogram
    y = _audio(batch, frames, seed=frames)
    ref = M.mel_spectrogram(y, SOUND["n_fft"], SOUND["mels"], SOUND["sr"], SOUND["hop"], SOUND["win"], 0, None)
    got = mel_spectrogram(torch.from_numpy(y).cuda(), SOUND["n_fft"], SOUND["mels"], SOUND["sr"], SOUND["hop"], SOUND["win"],
                          0, None, center=False)
    assert got.shape == (batch, 80, frames) and got.dtype == torch.float32
    g = got.cpu().numpy()
    assert np.isfinite(g).all()
    assert np.abs(g - ref).max() < 2e-3 and rel_l2(g, ref) < 1e-4           # log domain, fp32 DFT of 1024 points


@pytest.mark.gpu
def test_mel_frontend_rejects_what_the_reference_never_uses():
    from speaker_embedding_torch_amd.meldataset import mel_spectrogram
    y = torch.zeros(1, 4096).cuda()
    with pytest.raises(NotImplementedError):
        mel_spectrogram(y, 1024, 80, 22050, 256, 800, 0, None)
    with pytest.raises(RuntimeError):
        mel_spectrogram(y.cpu(), 1024, 80, 22050, 256, 1024, 0, None)
    with pytest.raises(RuntimeError):
        mel_spectrogram(torch.zeros(1, 100).cuda(), 1024, 80, 22050, 256, 1024, 0, None)       # shorter than the reflect pad


@pytest.mark.gpu
def test_inferencer_takes_wav_files(tmp_path):
    """wav -> (HIP front-end) -> multi-slice d-vectors == the same utterance fed as the oracle's mel pattern."""
    from scipy.io.wavfile import write
    from speaker_embedding_torch_amd.Inference import Inferencer
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hp_path = os.path.join(repo, "speaker_embedding_torch_amd", "Hyper_Parameters.yaml")
    inf = Inferencer(hp_path, None, batch_size=4, precision="fp32")
    paths, npys = [], []
    for i in range(3):
        y = _audio(1, 260 + 11 * i, seed=10 + i)[0]
        pcm = np.round(y * 32767.0).astype(np.int16)
        p = str(tmp_path / f"u{i}.wav")
        write(p, 22050, pcm)
        paths.append(p)
        a = pcm.astype(np.float32) / 32768.0
        a = a / np.abs(a).max() * 0.95
        a = a[:len(a) - len(a) % 256]
        mel = M.mel_spectrogram(a[None], 1024, 80, 22050, 256, 1024, 0, None)[0][:, :len(a) // 256]
        q = str(tmp_path / f"u{i}.npy")
        np.save(q, mel)
        npys.append(q)
    np.random.seed(4); e_wav, l1 = inf.Inference(paths, list("abc"))
    np.random.seed(4); e_npy, l2 = inf.Inference(npys, list("abc"))
    assert l1 == l2 == list("abc") and e_wav.shape == (3, 256)
    assert (e_wav - e_npy).abs().max().item() < 2e-4
    write(str(tmp_path / "other_rate.wav"), 16000, np.zeros(16000, dtype=np.int16))
    with pytest.raises(NotImplementedError):
        inf.Inference([str(tmp_path / "other_rate.wav")], ["x"])
