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


@pytest.mark.parametrize("B,T,K,causal", [(1, 1, 31, False), (2, 50, 31, False), (3, 97, 15, False), (2, 77, 15, True), (1, 1000, 31, False)])
@pytest.mark.parametrize("mean_shift", [0.0, 40.0])
def test_dwconv_with_layernorm_silu_epilogue(hip, B, T, K, causal, mean_shift):
    """The conv module's `activation(norm(depthwise_conv(x)))` (convolution.py:131-138) in one kernel against (i) the two-kernel
    path it replaces (same roundings; the row sums are added in another order: at most one bf16 step apart) and (ii) the fp32
    module chain on the bf16-rounded intermediates.  Ragged last tile (T % 16), masked lengths, the causal form, rows with a
    mean far from zero (two-pass variance), non-trivial gamma / beta."""
    from paper_accurate_fast_cheap_amd.hip_ops import add_layernorm, depthwise_conv1d_cl, depthwise_conv1d_cl_ln_silu
    C, bf = 512, torch.bfloat16
    Tin, left = (T + K - 1, 0) if causal else (T, (K - 1) // 2)
    x = synth.randn((B, Tin, C), 11).to(bf)
    w = synth.randn((C, 1, K), 12, 0.2).to(bf)
    b = (synth.randn((C,), 13, 0.1) + mean_shift).to(bf)
    gamma, beta = (1 + synth.randn((C,), 14, 0.3)).to(bf), synth.randn((C,), 15, 0.2).to(bf)
    lens = None if causal else torch.tensor([Tin, max(1, Tin // 2), max(1, Tin - 3)][:B], dtype=torch.int32).cuda()
    g = [t.cuda() for t in (x, w, b, gamma, beta)]
    got = depthwise_conv1d_cl_ln_silu(g[0], g[1], g[2], left, T, g[3], g[4], 1e-5, lens=lens)
    dw = depthwise_conv1d_cl(g[0], g[1], g[2], left, T, lens=lens)
    two = add_layernorm(dw, None, 1.0, g[3], g[4], silu=True, eps=1e-5)[1]
    assert got.shape == two.shape and torch.isfinite(got).all()
    d = (got.float() - two.float()).abs()
    assert float(d.max()) <= 2 ** -7 * float(two.float().abs().max()) + 1e-3, float(d.max())
    assert float((got != two).float().mean()) < 0.02          # all but a few ties of the reduction order are bit-identical
    ln = F.layer_norm(dw.float(), (C,), gamma.float().cuda(), beta.float().cuda(), 1e-5).to(bf).float()
    ref = (ln * torch.sigmoid(ln)).to(bf)
    torch.testing.assert_close(got.float(), ref.float(), rtol=2 ** -6, atol=2e-2)


def test_dwconv_causal_form(hip):
    from paper_accurate_fast_cheap_amd.hip_ops import depthwise_conv1d_cl
    B, T, C, K = 2, 41, 128, 15
    x = synth.randn((B, T + K - 1, C), 5)
    w = synth.randn((C, 1, K), 6, 0.2)
    ref = F.conv1d(x.transpose(1, 2), w, None, padding=0, groups=C).transpose(1, 2)
    got = depthwise_conv1d_cl(x.cuda(), w.cuda(), None, 0, T).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,K,causal", [(1, 1, 128, 31, False), (2, 50, 128, 31, False), (3, 300, 512, 31, False),
                                            (2, 133, 256, 15, True), (2, 40, 128, 7, False), (32, 129, 128, 15, False)])
def test_dwconv_gradients_match_conv1d_autograd(hip, dtype, B, T, C, K, causal):
    """The training-side op: dx / dw / dbias against float64 autograd through F.conv1d on the same (rounded) operands."""
    from paper_accurate_fast_cheap_amd.hip_ops import depthwise_conv1d_cl_autograd
    T_in = T + K - 1 if causal else T
    x = synth.randn((B, T_in, C), 1).to(dtype)
    w = synth.randn((C, 1, K), 2, 0.2)                 # fp32 master parameters, cast inside (as autocast would)
    b = synth.randn((C,), 3, 0.1)
    gy = synth.randn((B, T, C), 4).to(dtype)
    xr = x.double().requires_grad_()
    wr = w.to(dtype).double().requires_grad_()
    br = b.to(dtype).double().requires_grad_()
    ref = F.conv1d(xr.transpose(1, 2), wr, br, padding=0 if causal else (K - 1) // 2, groups=C).transpose(1, 2)
    ref.backward(gy.double())
    xg = x.cuda().requires_grad_()
    wg = w.cuda().requires_grad_()
    bg = b.cuda().requires_grad_()
    got = depthwise_conv1d_cl_autograd(xg, wg, bg, 0 if causal else (K - 1) // 2, T)
    got.backward(gy.cuda())
    assert wg.grad.dtype == torch.float32 and wg.grad.shape == w.shape and xg.grad.dtype == dtype
    lo = dtype == torch.bfloat16
    torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), rtol=2 ** -7 if lo else 1e-4, atol=2e-2 if lo else 1e-5)
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, rtol=2 ** -7 if lo else 1e-4, atol=2e-2 if lo else 1e-5)
    # the sums over B*T are accumulated in fp32 from exact products; bf16 only rounds the result once
    scale = float(wr.grad.abs().max())
    assert float((wg.grad.cpu().double() - wr.grad).abs().max()) <= (1e-2 if lo else 1e-5) * scale
    assert float((bg.grad.cpu().double() - br.grad).abs().max()) <= (1e-2 if lo else 1e-5) * float(br.grad.abs().max())


def test_dwconv_gradients_without_bias_and_frozen_weight(hip):
    from paper_accurate_fast_cheap_amd.hip_ops import depthwise_conv1d_cl_autograd
    x = synth.randn((2, 37, 128), 1).cuda().requires_grad_()
    w = synth.randn((128, 1, 15), 2, 0.2).cuda()                      # frozen (the FT-LFXL recipe trains the slot only)
    y = depthwise_conv1d_cl_autograd(x, w, None, 7, 37)
    y.sum().backward()
    ref = F.conv_transpose1d(torch.ones(2, 128, 37), w.cpu(), padding=7, groups=128).transpose(1, 2)
    torch.testing.assert_close(x.grad.cpu(), ref, rtol=1e-4, atol=1e-5)


def test_conv_module_training_path_uses_the_kernels(hip):
    """ConvolutionModule under autograd == the library path it replaced (nn.Conv1d between two transposes)."""
    from paper_accurate_fast_cheap_amd.transformer.convolution import ConvolutionModule
    torch.manual_seed(0)
    m = ConvolutionModule(128, 15, torch.nn.SiLU(), "layer_norm", causal=False, bias=True).cuda()
    x = synth.randn((3, 61, 128), 7).cuda().requires_grad_()
    mask = (torch.arange(61)[None, :] < torch.tensor([61, 40, 13])[:, None]).unsqueeze(1).cuda()
    y, _ = m(x, mask)
    y.square().sum().backward()
    got = {n: p.grad.clone() for n, p in m.named_parameters()}
    gx = x.grad.clone()
    m.zero_grad(); x.grad = None
    xm = x.masked_fill(~mask.transpose(1, 2), 0.0)
    h = F.glu(F.linear(xm, m.pointwise_conv1.weight.squeeze(-1), m.pointwise_conv1.bias), dim=-1)
    # the depthwise convolution spelled out with element-wise ops (unfold * taps, summed): the library's grouped
    # convolution would JIT-compile a kernel here, which takes seconds to minutes on a fresh box
    K = m.kernel_size
    hp = F.pad(h, (0, 0, (K - 1) // 2, (K - 1) // 2)).unfold(1, K, 1)                  # (B, T, C, K)
    h = (hp * m.depthwise_conv.weight.squeeze(1)).sum(-1) + m.depthwise_conv.bias
    h = F.linear(m.activation(m.norm(h)), m.pointwise_conv2.weight.squeeze(-1), m.pointwise_conv2.bias)
    h = h.masked_fill(~mask.transpose(1, 2), 0.0)
    torch.testing.assert_close(y, h, rtol=1e-4, atol=1e-5)
    h.square().sum().backward()
    torch.testing.assert_close(gx, x.grad, rtol=1e-3, atol=1e-4)
    for n, p in m.named_parameters():
        torch.testing.assert_close(got[n], p.grad, rtol=1e-3, atol=1e-3 * float(p.grad.abs().max()))
