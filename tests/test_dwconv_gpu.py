"""Channels-last depthwise conv kernel vs torch.nn.functional.conv1d on CPU (fp32 reference of the same op)."""
import pytest
import torch
import torch.nn.functional as F

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,K", [(1, 1, 128, 31), (2, 50, 128, 31), (3, 97, 512, 31), (2, 33, 256, 15), (1, 40, 128, 7)])
@pytest.mark.parametrize("glu", [False, True])
def test_dwconv_matches_conv1d(hip, dtype, B, T, C, K, glu):
    from paper_accurate_fast_cheap_amd.hip_ops import depthwise_conv1d_cl
    x = synth.randn((B, T, 2 * C if glu else C), 1).to(dtype)
    w = synth.randn((C, 1, K), 2, 0.2).to(dtype)
    b = synth.randn((C,), 3, 0.1).to(dtype)
    lens = torch.tensor([T, max(1, T // 2), max(1, T - 3)][:B], dtype=torch.int32)
    xin = F.glu(x, dim=-1) if glu else x
    keep = (torch.arange(T)[None, :] < lens[:, None]).unsqueeze(-1)
    xin = xin.masked_fill(~keep, 0.0)
    ref = F.conv1d(xin.float().transpose(1, 2), w.float(), b.float(), padding=(K - 1) // 2, groups=C).transpose(1, 2)
    got = depthwise_conv1d_cl(x.cuda(), w.cuda(), b.cuda(), (K - 1) // 2, T, glu=glu, lens=lens.cuda()).cpu()
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(got.float(), ref, **tol)


def test_dwconv_causal_form(hip):
    from paper_accurate_fast_cheap_amd.hip_ops import depthwise_conv1d_cl
    B, T, C, K = 2, 41, 128, 15
    x = synth.randn((B, T + K - 1, C), 5)
    w = synth.randn((C, 1, K), 6, 0.2)
    ref = F.conv1d(x.transpose(1, 2), w, None, padding=0, groups=C).transpose(1, 2)
    got = depthwise_conv1d_cl(x.cuda(), w.cuda(), None, 0, T).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)
