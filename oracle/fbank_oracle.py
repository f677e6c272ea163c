"""CPU restatement of the Kaldi-compatible log-mel filterbank the reference calls.

TEST INFRASTRUCTURE ONLY (same rule as oracle/wkv6_oracle.c).

*** parity unpinned ***  (cross-checked, not pinned: tests/test_oracle_goldens.py compares this restatement with the
independent Kaldi-compatible extractor shipped in `transformers`, max |delta log-mel| 1.6e-4.)  The arithmetic lives in a third-party dependency that is absent from /root/reference and
from this image: torchaudio.compliance.kaldi.fbank (requirements.txt:17 `torchaudio>=2.2.2`; the README installs
PyTorch 2.5.1 / torchaudio 2.5.1, README_RevPaper_Choose3.md:41).  The reference's call sites are
wenet/dataset/processor.py:363-369, wenet/bin/encoder-rtf.py:575-583, wenet/bin/recognize_wav2.py:510-518:
    kaldi.fbank(waveform * 2**15 | int16-valued float, num_mel_bins=80, frame_length=25, frame_shift=10,
                dither=0.0 (eval) | 1.0 (train), energy_floor=0.0, sample_frequency=16000)
with every other argument at its default.  The reference holds no fixture for it, so this file restates the
published algorithm (Kaldi's compute-fbank-feats as torchaudio implements it) with those defaults:
  snip_edges framing (m = 1 + (S - 400) // 160), optional dither, per-frame DC removal, pre-emphasis 0.97 with a
  replicated first sample, povey window hann(400, periodic=False) ** 0.85, zero-pad to 512, |rfft|^2,
  80 triangular mel filters between 20 Hz and Nyquist on the 1127 ln(1 + f/700) scale (slopes in mel space),
  log(max(energy, float32 eps)).
All arithmetic in float32 torch CPU ops, in the order torchaudio performs it.
"""
import math

import torch

EPS = torch.tensor(torch.finfo(torch.float32).eps)


def povey_window(n: int = 400) -> torch.Tensor:
    return torch.hann_window(n, periodic=False, dtype=torch.float32).pow(0.85)


def mel_banks(num_bins: int = 80, padded: int = 512, sample_freq: float = 16000.0, low_freq: float = 20.0,
              high_freq: float = 0.0) -> torch.Tensor:
    """(num_bins, padded // 2 + 1) float32; the last column (Nyquist bin) is the zero column torchaudio pads on."""
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left = mel_low + b * delta
    center = mel_low + (b + 1.0) * delta
    right = mel_low + (b + 2.0) * delta
    melf = 1127.0 * (1.0 + fft_bin_width * torch.arange(num_fft_bins, dtype=torch.float32) / 700.0).log()
    melf = melf.unsqueeze(0)
    up = (melf - left) / (center - left)
    down = (right - melf) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return torch.nn.functional.pad(bins, (0, 1), mode="constant", value=0)


def fbank(waveform: torch.Tensor, num_mel_bins: int = 80, frame_length: float = 25.0, frame_shift: float = 10.0,
          dither: float = 0.0, energy_floor: float = 0.0, sample_frequency: float = 16000.0,
          noise: torch.Tensor = None) -> torch.Tensor:
    """waveform (1, S) float32 in int16 range -> (m, num_mel_bins) float32.  `noise` (m, window) replaces the
    randn draw of the dither step so that a test can feed the same noise to both implementations."""
    assert waveform.dim() == 2 and energy_floor == 0.0
    wave = waveform[0].to(torch.float32)
    shift = int(sample_frequency * frame_shift * 0.001)
    size = int(sample_frequency * frame_length * 0.001)
    padded = 1 << (size - 1).bit_length()
    S = wave.numel()
    if S < size:
        return torch.empty((0, num_mel_bins))
    m = 1 + (S - size) // shift
    frames = wave.as_strided((m, size), (shift, 1)).clone()
    if dither != 0.0:
        frames = frames + (noise if noise is not None else torch.randn(frames.shape)) * dither
    frames = frames - frames.mean(dim=1, keepdim=True)
    prev = torch.nn.functional.pad(frames.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)
    frames = frames - 0.97 * prev[:, :-1]
    frames = frames * povey_window(size).unsqueeze(0)
    frames = torch.nn.functional.pad(frames, (0, padded - size), mode="constant", value=0)
    spec = torch.fft.rfft(frames).abs().pow(2.0)
    mel = torch.mm(spec, mel_banks(num_mel_bins, padded, sample_frequency).T)
    return torch.max(mel, EPS).log()
