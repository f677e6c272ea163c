"""HIP fbank vs the CPU restatement of Kaldi's fbank (oracle/fbank_oracle.py; parity unpinned: torchaudio is not
in the image -- if it is present on the box, it is used as a second, third-party cross-check)."""
import pytest
import torch

from oracle import fbank_oracle as FO
from tests import synth

pytestmark = pytest.mark.gpu


def _wave(S, seed):
    t = torch.arange(S, dtype=torch.float32)
    env = 0.2 + 0.8 * (0.5 + 0.5 * torch.sin(2 * 3.14159265 * t / 9000.0))
    x = synth.randn((S,), seed) * 2500.0 * env + 600.0 * torch.sin(2 * 3.14159265 * 440.0 * t / 16000.0) + 37.0
    return x.round().clamp(-32768, 32767).unsqueeze(0)


@pytest.mark.parametrize("S", [400, 559, 560, 16000, 16000 * 7 + 123])
def test_fbank_matches_oracle(hip, S):
    from paper_accurate_fast_cheap_amd.dataset.fbank import fbank
    w = _wave(S, 5)
    ref = FO.fbank(w, num_mel_bins=80)
    got = fbank(w.cuda(), num_mel_bins=80, frame_length=25.0, frame_shift=10.0, dither=0.0, energy_floor=0.0,
                sample_frequency=16000.0).cpu()
    assert got.shape == ref.shape == (1 + (S - 400) // 160, 80)
    # log-mel values are ~10..25; direct fp32 DFT vs fp32 FFT differ by ~1e-6 relative on the power spectrum
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-3)
    assert float((got - ref).abs().mean()) < 1e-4


def test_fbank_dither_and_short_and_quiet(hip):
    from paper_accurate_fast_cheap_amd.dataset.fbank import fbank
    w = _wave(8000, 6)
    m = 1 + (8000 - 400) // 160
    noise = synth.randn((m, 400), 7)
    ref = FO.fbank(w, num_mel_bins=80, dither=1.0, noise=noise)
    got = fbank(w.cuda(), num_mel_bins=80, dither=1.0, noise=noise.cuda()).cpu()
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-3)
    assert fbank(_wave(399, 1).cuda(), num_mel_bins=80).shape == (0, 80)
    z = fbank(torch.zeros(1, 4000).cuda(), num_mel_bins=80).cpu()     # digital silence -> log(eps) floor
    torch.testing.assert_close(z, torch.full_like(z, float(torch.log(FO.EPS))), rtol=0, atol=1e-5)
    got23 = fbank(w.cuda(), num_mel_bins=23).cpu()                     # the function's own default bin count
    torch.testing.assert_close(got23, FO.fbank(w, num_mel_bins=23), rtol=0, atol=2e-3)


def test_fbank_vs_torchaudio_if_present(hip):
    ta = pytest.importorskip("torchaudio")
    from paper_accurate_fast_cheap_amd.dataset.fbank import fbank
    w = _wave(48000, 8)
    ref = ta.compliance.kaldi.fbank(w, num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0, energy_floor=0.0,
                                    sample_frequency=16000)
    torch.testing.assert_close(fbank(w.cuda(), num_mel_bins=80).cpu(), ref, rtol=0, atol=2e-3)
    torch.testing.assert_close(FO.fbank(w, num_mel_bins=80), ref, rtol=0, atol=1e-4)


def test_fbank_30_minutes_shape_and_linearity(hip):
    """Full BASELINE size (28.8 M samples): frame count, and a size-independent property: scaling the waveform
    by g adds 2 ln g to every log-mel value (all stages before the log are linear; power spectrum is quadratic)."""
    from paper_accurate_fast_cheap_amd.dataset.fbank import fbank
    S = 1800 * 16000
    g = torch.Generator(device="cuda").manual_seed(1)
    w = (torch.randn(1, S, device="cuda", generator=g) * 3000).round()
    a = fbank(w, num_mel_bins=80)
    b = fbank(w * 4.0, num_mel_bins=80)
    assert a.shape == (179998, 80)
    torch.testing.assert_close(b - a, torch.full_like(a, 2 * 1.3862943611198906), rtol=0, atol=2e-4)
