"""80-bin Kaldi log-mel fbank on the MI355X, behind the call the reference makes.

Reference call sites: `compute_fbank(sample, num_mel_bins, frame_length, frame_shift, dither)`
wenet/dataset/processor.py:343-371 and `compute_feats` wenet/bin/encoder-rtf.py:558-585, both of which call
torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=..., frame_length=25, frame_shift=10, dither=...,
energy_floor=0.0, sample_frequency=16000).  `fbank()` below takes the same arguments with the same meaning and
returns the same (frames, num_mel_bins) float32 tensor -- on the device the waveform lives on.

The constant tables (povey window, DFT matrix, mel filters) are built once per device in float64 / float32 on
the host exactly as torchaudio builds them and handed to the kernel; the library itself keeps no state."""
import math
from typing import Dict, Optional, Tuple

import torch

from .. import _lib
from .._lib import c_int, c_void_p

_tables: Dict[Tuple[str, int], dict] = {}


def _bind():
    L = _lib.lib()
    if getattr(L, "_pafc_fbank_bound", False):
        return L
    from ctypes import c_float, c_long
    P, I = c_void_p, c_int
    _lib._sig(L.pafc_fbank_num_frames, c_long, c_long)
    _lib._sig(L.pafc_fbank_tables_cols, I)
    _lib._sig(L.pafc_fbank_f32, I, P, c_long, P, P, P, P, P, I, P, c_float, c_float, P, P)
    L._pafc_fbank_bound = True
    return L


def mel_banks(num_bins: int, padded: int = 512, sample_freq: float = 16000.0, low_freq: float = 20.0) -> torch.Tensor:
    """(num_bins, padded/2 + 1) triangular filters in mel space, float32 arithmetic as torchaudio's get_mel_banks
    (+ its zero column for the Nyquist bin)."""
    nfft_bins = padded // 2
    high_freq = 0.5 * sample_freq
    width = sample_freq / padded
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(low_freq), mel(high_freq)
    delta = (hi - lo) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    melf = (1127.0 * (1.0 + width * torch.arange(nfft_bins, dtype=torch.float32) / 700.0).log()).unsqueeze(0)
    up = (melf - left) / (center - left)
    down = (right - melf) / (right - center)
    w = torch.clamp_min(torch.min(up, down), 0.0)
    return torch.nn.functional.pad(w, (0, 1))


def _get_tables(device: torch.device, num_mel_bins: int) -> dict:
    key = (str(device), num_mel_bins)
    t = _tables.get(key)
    if t is not None:
        return t
    cols = _bind().pafc_fbank_tables_cols()
    n = torch.arange(400, dtype=torch.float64).unsqueeze(1)
    k = torch.arange(257, dtype=torch.float64).unsqueeze(0)
    ang = 2.0 * math.pi * n * k / 512.0
    dft = torch.zeros(400, cols, dtype=torch.float64)
    dft[:, 0:514:2] = torch.cos(ang)
    dft[:, 1:514:2] = -torch.sin(ang)
    melw = mel_banks(num_mel_bins)
    nz = melw > 0
    idx = torch.arange(257).unsqueeze(0).expand_as(melw)
    lo = torch.where(nz, idx, torch.full_like(idx, 257)).min(dim=1).values
    hi = torch.where(nz, idx + 1, torch.zeros_like(idx)).max(dim=1).values
    t = dict(window=torch.hann_window(400, periodic=False, dtype=torch.float32).pow(0.85).to(device),
             dft=dft.to(torch.float32).to(device).contiguous(), melw=melw.to(device).contiguous(),
             lo=lo.to(torch.int32).to(device), hi=hi.to(torch.int32).to(device))
    _tables[key] = t
    return t


def fbank(waveform: torch.Tensor, num_mel_bins: int = 23, frame_length: float = 25.0, frame_shift: float = 10.0,
          dither: float = 0.0, energy_floor: float = 0.0, sample_frequency: float = 16000.0,
          noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """waveform: (channels, S) float tensor in int16 range ON THE GPU (channel 0 is used, as torchaudio's default
    channel=-1 -> first channel); returns (frames, num_mel_bins) float32.  dither != 0 draws standard-normal noise
    per frame sample on the device (or uses `noise` (frames, 400) if given)."""
    if (frame_length, frame_shift, sample_frequency, energy_floor) != (25.0, 10.0, 16000.0, 0.0):
        raise _lib.PafcError("fbank kernel is built for the reference's configuration: 25 ms / 10 ms frames at "
                             "16 kHz, energy_floor 0 (processor.py:363-369)")
    if waveform.dim() != 2:
        raise _lib.PafcError("waveform must be (channels, samples)")
    _lib.require_gpu(waveform)
    L = _bind()
    wave = waveform[0].to(torch.float32).contiguous()
    S = wave.numel()
    m = L.pafc_fbank_num_frames(S)
    out = torch.empty((m, num_mel_bins), dtype=torch.float32, device=wave.device)
    if m == 0:
        return out
    t = _get_tables(wave.device, num_mel_bins)
    if dither != 0.0 and noise is None:
        noise = torch.randn((m, 400), dtype=torch.float32, device=wave.device)
    if noise is not None:
        _lib.require_gpu(noise)
        if noise.shape != (m, 400) or noise.dtype != torch.float32:
            raise _lib.PafcError("noise must be float32 (frames, 400)")
    rc = L.pafc_fbank_f32(_lib.ptr(wave), S, _lib.ptr(t["window"]), _lib.ptr(t["dft"]), _lib.ptr(t["melw"]),
                          _lib.ptr(t["lo"]), _lib.ptr(t["hi"]), num_mel_bins,
                          _lib.ptr(noise if dither != 0.0 else None), float(dither), 0.97, _lib.ptr(out),
                          _lib.stream_of(wave))
    _lib.check(rc, "pafc_fbank_f32")
    return out


def compute_fbank(sample: dict, num_mel_bins: int = 23, frame_length: int = 25, frame_shift: int = 10,
                  dither: float = 0.0) -> dict:
    """wenet/dataset/processor.py:343-371: {key, wav (float in [-1, 1)), sample_rate} -> adds 'feat'."""
    assert "sample_rate" in sample and "wav" in sample and "key" in sample
    waveform = sample["wav"] * (1 << 15)
    sample["feat"] = fbank(waveform, num_mel_bins=num_mel_bins, frame_length=float(frame_length),
                           frame_shift=float(frame_shift), dither=dither, energy_floor=0.0,
                           sample_frequency=float(sample["sample_rate"]))
    return sample
