"""Fused inference executor (transformer/fused.py) and its glue kernels vs the op-by-op module path and fp32
PyTorch references of the same ops.  GPU only."""
import contextlib

import pytest
import torch
import torch.nn.functional as F

from tests import synth
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,out_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                             (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("C", [128, 512])
def test_add_layernorm_matches_torch(hip, dtype, out_dtype, C):
    from paper_accurate_fast_cheap_amd.hip_ops import add_layernorm
    B, T = 3, 37
    x = synth.randn((B, T, C), 1).to(dtype)
    y = synth.randn((B, T, C), 2).to(dtype)
    g1, b1, g2, b2 = (synth.randn((C,), s, 0.3).to(dtype) + (1 if s % 2 else 0) for s in (3, 4, 5, 6))
    lens = torch.tensor([37, 20, 1], dtype=torch.int32)
    keep = (torch.arange(T)[None, :] < lens[:, None]).unsqueeze(-1)
    # op-by-op reference in the same dtype (what PyTorch would materialise), on CPU
    xn = x + 0.5 * y.masked_fill(~keep, 0.0)
    o1 = F.silu(F.layer_norm(xn, (C,), g1, b1, 1e-5).to(out_dtype)).masked_fill(~keep, 0.0)
    o2 = F.layer_norm(o1.to(dtype), (C,), g2, b2, 1e-5).to(out_dtype)
    gx, go1, go2 = add_layernorm(x.cuda(), y.cuda(), 0.5, g1.cuda(), b1.cuda(), out_dtype=out_dtype, silu=True,
                                 zero_rows=True, lens=lens.cuda(), T=T, mask_y=True, gamma2=g2.cuda(), beta2=b2.cuda())
    tol = dict(rtol=1e-4, atol=1e-5) if out_dtype == torch.float32 else dict(rtol=2 ** -7, atol=2 ** -7)
    torch.testing.assert_close(gx.cpu().float(), xn.float(), rtol=1e-6 if dtype == torch.float32 else 2 ** -8, atol=1e-6)
    torch.testing.assert_close(go1.cpu().float(), o1.float(), **tol)
    torch.testing.assert_close(go2.cpu().float(), o2.float(), rtol=tol["rtol"] * 4, atol=tol["atol"] * 4)


def test_add_layernorm_side_by_side_outputs(hip):
    from paper_accurate_fast_cheap_amd.hip_ops import add_layernorm
    M, C = 50, 128
    ya, yb = synth.randn((M, C), 1), synth.randn((M, C), 2)
    g, b = synth.randn((C,), 3) + 1, synth.randn((C,), 4)
    cat = torch.zeros(M, 2 * C, device="cuda")
    add_layernorm(ya.cuda(), None, 1.0, g.cuda(), b.cuda(), out1=cat[:, :C])
    add_layernorm(yb.cuda(), None, 1.0, b.cuda(), g.cuda(), out1=cat[:, C:])
    ref = torch.cat([F.layer_norm(ya, (C,), g, b), F.layer_norm(yb, (C,), b, g)], dim=1)
    torch.testing.assert_close(cat.cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tmix_glue_matches_module_chain(hip, dtype):
    """pafc_tmix_shift_mix / pafc_tmix_mix4 vs the reference's op chain (src/model.py:274-284) in the same dtype."""
    from paper_accurate_fast_cheap_amd.hip_ops import tmix_mix4, tmix_shift_mix
    B, T, C = 2, 19, 128
    x = synth.randn((B, T, C), 1).to(dtype)
    maa = [synth.randn((C,), 10 + i, 0.5).to(dtype) for i in range(2)]
    maa4 = synth.randn((2, 4, C), 20, 0.5).to(dtype)
    m = synth.randn((2, 4, B * T, C), 30, 0.3).to(dtype)
    prev = [F.pad(x, (0, 0, 1, -1)), F.pad(x, (0, 0, -1, 1))]
    got = tmix_shift_mix(x.cuda(), maa[0].cuda(), maa[1].cuda()).cpu()
    z = tmix_mix4(x.cuda(), m.cuda(), maa4.cuda()).cpu()
    # bf16: every intermediate is rounded where PyTorch rounds it -> bit-identical.  fp32: the compiler may
    # contract x + xx * m into one fma (one rounding fewer than the op chain) -> 1 ulp
    tol = dict(rtol=0, atol=0) if dtype == torch.bfloat16 else dict(rtol=1e-6, atol=1e-6)
    for d in range(2):
        xx = prev[d] - x
        torch.testing.assert_close(got[d], x + xx * maa[d], **tol)
        for q in range(4):
            ref = x + xx * (maa4[d, q] + m[d, q].view(B, T, C))
            torch.testing.assert_close(z[q, d].view(B, T, C), ref, **tol)
    got1 = tmix_shift_mix(x.cuda(), maa[0].cuda(), None, reverse0=True).cpu()
    torch.testing.assert_close(got1[0], x + (prev[1] - x) * maa[0], **tol)


@pytest.mark.parametrize("variant", ["bf16slot", "f32", "uni_bf16slot", "uni_bf16model"])
def test_fused_executor_equals_module_path(hip, variant):
    """Same weights, same inputs: the fused executor against the op-by-op module path on the GPU (ragged batch)."""
    from paper_accurate_fast_cheap_amd.transformer.cmvn import GlobalCMVN
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    g = load_golden("encoder_reduced_" + variant)
    enc = ConformerEncoder(80, global_cmvn=GlobalCMVN(torch.zeros(80), torch.ones(80)), **g["conf"])
    enc.load_state_dict(synth.synth_state_dict(g["spec"], g["seed"]))
    if variant == "uni_bf16model":
        enc = enc.to(torch.bfloat16)
    enc = enc.cuda().eval()
    xs, lens = g["xs"].cuda(), g["lens"].cuda()
    with torch.no_grad():
        assert enc._fused(xs) is not None, "fused executor not eligible"
        out_f, masks_f, layers_f = enc.forward_return_layers(xs, lens, want_layers=True)
        yc_f, _, _ = enc.forward_chunk(g["chunk_x"].cuda(), 0, -1)
        enc.fused_inference = False
        out_p, masks_p, layers_p = enc.forward_return_layers(xs, lens, want_layers=True)
        yc_p, _, _ = enc.forward_chunk(g["chunk_x"].cuda(), 0, -1)
    assert torch.equal(masks_f, masks_p)
    if variant == "f32":
        tol = dict(rtol=1e-3, atol=1e-4)
    elif variant == "uni_bf16model":
        tol = dict(rtol=0.0, atol=0.4)
    else:
        tol = dict(rtol=0.0, atol=0.1)
    for a, b in ((layers_f[0], layers_p[0]), (out_f, out_p), (yc_f, yc_p)):
        torch.testing.assert_close(a.float(), b.float(), **tol)
        assert float((a.float() - b.float()).abs().mean()) <= (1e-5 if variant == "f32" else 2e-2 if variant == "uni_bf16model" else 6e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_linear_bias_silu_epilogue(hip, dtype):
    from paper_accurate_fast_cheap_amd.hip_ops import linear_bias_act
    x = synth.randn((3, 41, 128), 1).to(dtype)
    w = synth.randn((256, 128), 2, 0.1).to(dtype)
    b = synth.randn((256,), 3, 0.2).to(dtype)
    ref = F.silu(F.linear(x.float(), w.float(), b.float()))
    got = linear_bias_act(x.cuda(), w.cuda(), b.cuda(), "silu").cpu()
    got_id = linear_bias_act(x.cuda(), w.cuda(), b.cuda(), "none").cpu()
    tol = dict(rtol=2 ** -7, atol=1e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got.float(), ref, **tol)
    torch.testing.assert_close(got_id.float(), F.linear(x.float(), w.float(), b.float()), **tol)
    # residual-fused form: res + alpha * x W^T + bias, out of place and over the residual
    res = synth.randn((3, 41, 256), 4).to(dtype)
    want = res.float() + 0.5 * F.linear(x.float(), w.float()) + b.float()
    got_r = linear_bias_act(x.cuda(), w.cuda(), b.cuda(), "none", alpha=0.5, residual=res.cuda())
    torch.testing.assert_close(got_r.cpu().float(), want, **tol)
    buf = res.cuda().clone()
    same = linear_bias_act(x.cuda(), w.cuda(), None, "none", residual=buf, inplace=True)
    assert same.data_ptr() == buf.data_ptr()
    torch.testing.assert_close(buf.cpu().float(), res.float() + F.linear(x.float(), w.float()), **tol)


@pytest.mark.parametrize("M,N,K,act,res,bias", [
    (1, 1, 4, "none", False, False), (37, 5, 12, "tanh", False, True), (499, 512, 512, "silu", False, True),
    (4990, 512, 2048, "none", True, True), (700, 5000, 512, "none", False, True), (130, 130, 36, "relu", True, False),
    (31936, 2048, 512, "silu", False, True), (44998, 512, 9728, "none", False, True)])
def test_gemm_f32_own_kernel(hip, M, N, K, act, res, bias):
    """pafc_gemm_f32 (csrc/gemm_f32.hip: exact fp32 products on the fp32 matrix cores, fused epilogue) against a float64 product:
    ragged M / N / K tails (rows clamped, quads beyond K zero), both tile sizes, every activation, residual in place with
    alpha, the shapes of a pure-fp32 model (FFN, Linear(9728, 512), CTC head) -- to fp32 round-off of a K-term sum."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_f32
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g) * 0.3 if bias else None
    r = torch.randn(M, N, device="cuda", generator=g) if res else None
    alpha = 0.5 if res else 1.0
    want = alpha * (a.double() @ w.double().t())
    if b is not None:
        want = want + b.double()
    if r is not None:
        want = want + r.double()
    want = {"none": lambda t: t, "silu": F.silu, "tanh": torch.tanh, "relu": F.relu}[act](want)
    if res:
        buf = torch.full((M + 8, N), 7.0, device="cuda")
        buf[:M] = r
        got = gemm_f32(a, w, b, act, alpha=alpha, residual=buf[:M], out=buf[:M])
        assert got.data_ptr() == buf.data_ptr() and bool((buf[M:] == 7.0).all())
    else:
        got = gemm_f32(a, w, b, act, alpha=alpha)
    torch.testing.assert_close(got.double(), want, rtol=2e-5, atol=2e-5 * max(1.0, K ** 0.5 / 16))


def test_gemm_f32_batched_strided_views(hip):
    """The batched forms the fp32 time-mix uses (fused.slot_forward without the bf16 slot): stacked projections (Z, M, K) x
    (Z, N, K), and the LoRA up-projection whose A operand is a strided view -- four (M, 32) slices of one (M, 128) tensor,
    batch stride 32, row stride 128 -- each against torch.bmm in float64."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_f32
    g = torch.Generator(device="cuda").manual_seed(3)
    z = torch.randn(6, 1203, 128, device="cuda", generator=g)
    wz = torch.randn(6, 128, 128, device="cuda", generator=g) / 11
    torch.testing.assert_close(gemm_f32(z, wz).double(), torch.bmm(z.double(), wz.double().transpose(1, 2)), rtol=2e-5, atol=2e-5)
    t = torch.tanh(torch.randn(1203, 128, device="cuda", generator=g))
    w2t = torch.randn(4, 512, 32, device="cuda", generator=g) / 6          # Linear layout (N, K) per slice
    m = torch.empty(4, 1203, 512, device="cuda")
    gemm_f32(t.view(1203, 4, 32).transpose(0, 1), w2t, out=m)
    want = torch.bmm(t.double().view(1203, 4, 32).transpose(0, 1), w2t.double().transpose(1, 2))
    torch.testing.assert_close(m.double(), want, rtol=2e-5, atol=2e-5)
    bias = torch.randn(6, 128, device="cuda", generator=g)
    got = gemm_f32(z, wz, bias, "tanh")
    torch.testing.assert_close(got.double(), torch.tanh(torch.bmm(z.double(), wz.double().transpose(1, 2)) + bias.double()[:, None]),
                               rtol=2e-5, atol=2e-5)


@pytest.mark.timeout(180)
def test_fp32_products_with_two_streams_in_flight(hip):
    """The round-5 c2 stall, as a regression test.  Its cause (tools/micro/two_stream_linear.py, profiles/r06_c2_stall_*): the
    framework's fp32 GEMM -- what the fp32 projections of a ragged decode batch fell back to -- never finishes when two HIP
    streams issue it concurrently; with one stream the same pass takes a second.  The fp32 products of the inference paths now run
    on pafc_gemm_f32 (no workspace, no plan objects, no library call): the library-bound part of a c2 pass (24 decode batches
    x 12 layers x 4 fp32 projections + the CTC head, two batches in flight, the host running ahead of the device) finishes, twice,
    with the results of the same pass on ONE stream, bit for bit."""
    from paper_accurate_fast_cheap_amd import hip_ops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(1)
    rows = [64 * t for t in range(499, 20, -20)]                    # 24 batches, 31 936 ... 2 496 rows
    x512 = torch.randn(rows[0], 512, device=dev, generator=g)
    w1, b1 = torch.randn(2048, 512, device=dev, generator=g) / 23, torch.randn(2048, device=dev, generator=g) * 0.1
    w2, b2 = torch.randn(512, 2048, device=dev, generator=g) / 45, torch.randn(512, device=dev, generator=g) * 0.1
    wp, bp = torch.randn(512, 512, device=dev, generator=g) / 23, torch.randn(512, device=dev, generator=g) * 0.1
    wc, bc = torch.randn(5000, 512, device=dev, generator=g) / 23, torch.randn(5000, device=dev, generator=g) * 0.1
    L = hip_ops.linear_bias_act

    def one_pass(streams):
        main = torch.cuda.current_stream()
        for s_ in streams:
            s_.wait_stream(main)
        outs = []
        for i, m in enumerate(rows):
            with (torch.cuda.stream(streams[i % len(streams)]) if streams else contextlib.nullcontext()):
                x = x512[:m]
                for _ in range(12):
                    h = L(x, w1, b1, "silu")
                    x = L(h, w2, b2, "none", alpha=0.5, residual=x)
                    c = L(x, wp, bp, "none")
                    x = L(c, wp, bp, "none", residual=x)
                    x = x * 0.25                                   # keeps twelve layers of random weights in range
                outs.append(L(x, wc, bc, "none").abs().mean())
        for s_ in streams:
            main.wait_stream(s_)
        return torch.stack(outs).cpu()                             # the pass's only host wait

    want = one_pass([])
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(2):
        got = one_pass(side)
        assert torch.equal(got, want) and bool(torch.isfinite(got).all())


@pytest.mark.parametrize("B,T,C,nd", [(2, 37, 128, 2), (2, 2101, 128, 2), (1, 4099, 64, 1)])
def test_lora_mix4_fused_equals_two_step(hip, B, T, C, nd):
    """pafc_tmix_lora_mix4_bf16 (LoRA up-projection on MFMA inside the lerp pass) vs bmm + pafc_tmix_mix4; from 256 row tiles
    on, the weight-stationary kernel (a block keeps its W2 slice in registers and walks over row tiles)."""
    from paper_accurate_fast_cheap_amd.hip_ops import tmix_lora_mix4, tmix_mix4
    x = synth.randn((B, T, C), 1).bfloat16().cuda()
    t = torch.tanh(synth.randn((nd, B * T, 128), 2)).bfloat16().cuda()
    w2 = (synth.randn((nd, 4, 32, C), 3) * 0.2).bfloat16().cuda()
    maa = synth.randn((nd, 4, C), 4, 0.5).bfloat16().cuda()
    m = torch.stack([torch.bmm(t[d].view(B * T, 4, 32).transpose(0, 1), w2[d]) for d in range(nd)])   # (nd,4,M,C)
    ref = tmix_mix4(x, m.contiguous(), maa)
    got = tmix_lora_mix4(x, t, w2.transpose(2, 3).contiguous(), maa)
    # identical op chain; the only freedom is the K = 32 summation order inside the MFMA vs the library GEMM
    torch.testing.assert_close(got.float(), ref.float(), rtol=2 ** -7, atol=2 ** -7)
    assert float((got.float() - ref.float()).abs().mean()) < 1e-3


@pytest.mark.parametrize("B,T,C,nd,rev0", [(2, 37, 512, 2, False), (1, 4099, 512, 2, False), (3, 65, 512, 1, True),
                                           (1, 1, 512, 1, False), (2, 19, 128, 2, False)])
def test_tmix_lora_down_equals_shift_mix_plus_gemm(hip, monkeypatch, B, T, C, nd, rev0):
    """pafc_tmix_lora_down_bf16 (token shift + first lerp + LoRA down-projection + tanh, W1 resident in LDS) vs
    pafc_tmix_shift_mix followed by the tanh-epilogue GEMM: xxx is formed with the same roundings, so t differs only by the
    fp32 summation order of the K = C product.  C = 128 takes the two-step fallback of the wrapper."""
    from paper_accurate_fast_cheap_amd import hip_ops
    from paper_accurate_fast_cheap_amd.hip_ops import tmix_lora_down, tmix_shift_mix, gemm_bf16
    monkeypatch.setattr(hip_ops, "_LDS_RESIDENT_MIN_ROWS", 1)       # the one-pass kernel at every size (product: from 8192 rows)
    x = synth.randn((B, T, C), 1).bfloat16().cuda()
    maa = synth.randn((nd, C), 2, 0.5).bfloat16().cuda()
    w1n = (synth.randn((nd, 128, C), 3) * (1.5 / C ** 0.5)).bfloat16().cuda()
    xxx = tmix_shift_mix(x, maa[0], maa[1] if nd == 2 else None, reverse0=rev0)
    ref = gemm_bf16(xxx.view(nd, B * T, C), w1n, act="tanh")
    got = tmix_lora_down(x, maa, w1n, reverse0=rev0)
    assert got.shape == (nd, B * T, 128) and torch.isfinite(got.float()).all()
    torch.testing.assert_close(got.float(), ref.float(), rtol=2 ** -7, atol=2 ** -8)
    assert float((got.float() - ref.float()).abs().mean()) < 2e-4
    ref32 = torch.tanh(xxx.view(nd, B * T, C).float() @ w1n.float().transpose(1, 2))
    torch.testing.assert_close(got.float(), ref32, rtol=2 ** -7, atol=2 ** -8)


@pytest.mark.parametrize("rows,C,nd,with_bias", [(37, 512, 2, False), (4099, 512, 2, False), (300, 512, 1, True),
                                                 (1, 512, 1, False), (70, 128, 2, True)])
def test_decay_lora_equals_two_gemms(hip, monkeypatch, rows, C, nd, with_bias):
    """pafc_decay_lora_bf16 (both LoRA matrices resident in LDS, the 64-wide hidden tensor never in memory) vs the two GEMMs it
    replaces: the hidden values are rounded to bf16 at the same place, so w differs only by the fp32 summation order of the two
    products (and, rarely, a hidden value that rounds the other way).  C = 128 takes the two-GEMM fallback of the wrapper."""
    from paper_accurate_fast_cheap_amd import hip_ops
    from paper_accurate_fast_cheap_amd.hip_ops import decay_lora, gemm_bf16
    monkeypatch.setattr(hip_ops, "_LDS_RESIDENT_MIN_ROWS", 1)       # the one-pass kernel at every size (product: from 8192 rows)
    zw = synth.randn((nd, rows, C), 1).bfloat16().cuda()
    d1n = (synth.randn((nd, 64, C), 5) * (2.0 / C ** 0.5)).bfloat16().cuda()
    d2n = (synth.randn((nd, C, 64), 6) * 0.3).bfloat16().cuda()
    bias = (synth.randn((nd, C), 7) - 3).bfloat16().cuda() if with_bias else None
    w_ref = gemm_bf16(gemm_bf16(zw, d1n, act="tanh"), d2n)
    td32 = torch.tanh(zw.float() @ d1n.float().transpose(1, 2)).bfloat16().float()
    w32 = td32 @ d2n.float().transpose(1, 2)
    if with_bias:
        w_ref = w_ref + bias.view(nd, 1, C)
        w32 = (w32.bfloat16() + bias.view(nd, 1, C)).float()
    w = decay_lora(zw, d1n, d2n, bias)
    assert w.shape == (nd, rows, C) and torch.isfinite(w.float()).all()
    # a hidden value that rounds the other way moves w by at most |D2| * ulp(td); everything else is summation order
    torch.testing.assert_close(w.float(), w_ref.float(), rtol=2 ** -6, atol=3e-2)
    assert float((w.float() - w_ref.float()).abs().mean()) < 2e-3
    torch.testing.assert_close(w.float(), w32, rtol=2 ** -6, atol=3e-2)     # the same chain in fp32 on the bf16 inputs


@pytest.mark.parametrize("B,T1,F1,C", [(1, 9, 39, 128), (2, 37, 39, 128), (1, 201, 39, 512), (3, 5, 7, 256)])
def test_conv3x3s2_implicit_gemm(hip, B, T1, F1, C):
    """pafc_conv3x3s2_nhwc_bf16 vs torch conv2d (fp32 reference of the same op on bf16 inputs)."""
    from paper_accurate_fast_cheap_amd.hip_ops import conv3x3s2_nhwc
    x = synth.randn((B, T1, F1, C), 1).bfloat16()
    w = (synth.randn((C, C, 3, 3), 2) / (3 * C ** 0.5)).bfloat16()
    b = synth.randn((C,), 3, 0.1).bfloat16()
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), stride=2)).permute(0, 2, 3, 1)
    got = conv3x3s2_nhwc(x.cuda(), w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous().cuda(), b.cuda(), relu=True).cpu()
    assert got.shape == ref.shape
    torch.testing.assert_close(got.float(), ref, rtol=2 ** -7, atol=2e-2)
    assert float((got.float() - ref).abs().mean()) < 3e-3


@pytest.mark.parametrize("tile_m", [256, 192])
@pytest.mark.parametrize("B,T1,F1,C", [(2, 37, 19, 256), (3, 11, 9, 128), (1, 203, 39, 512), (1, 8001, 39, 512)])
def test_conv3x3s2_phase_pipelined(hip, tile_m, B, T1, F1, C):
    """The second subsampling convolution as an implicit GEMM on the phase-pipelined kernel (tap x 64-channel K-steps,
    tiles that run across batch entries, 2 / 4 / 8 K-steps per tap; the last case has more tiles than CUs: several tiles
    per block) vs torch conv2d in fp32, with and without ReLU."""
    from paper_accurate_fast_cheap_amd import _lib
    from paper_accurate_fast_cheap_amd.hip_ops import conv3x3s2_nhwc_ph
    x = synth.randn((B, T1, F1, C), 1).bfloat16()
    w = (synth.randn((C, C, 3, 3), 2) / (3 * C ** 0.5)).bfloat16()
    b = synth.randn((C,), 3, 0.1).bfloat16()
    dev = "cuda" if T1 > 1000 else "cpu"           # the long case: the fp32 reference on the GPU too
    lin = F.conv2d(x.to(dev).float().permute(0, 3, 1, 2), w.to(dev).float(), b.to(dev).float(), stride=2).permute(0, 2, 3, 1).cpu()
    taps = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous().cuda()
    with pytest.raises(_lib.PafcError):            # 64 input channels = 9 K-steps, an odd number: the other kernel's problem
        conv3x3s2_nhwc_ph(x[..., :64].contiguous().cuda(), taps[:, :64, :64].contiguous(), None, relu=True, tile_m=tile_m)
    got = conv3x3s2_nhwc_ph(x.cuda(), taps, b.cuda(), relu=True, tile_m=tile_m).cpu()
    assert got.shape == lin.shape
    torch.testing.assert_close(got.float(), F.relu(lin), rtol=2 ** -7, atol=2e-2)
    got = conv3x3s2_nhwc_ph(x.cuda(), taps, None, relu=False, tile_m=tile_m).cpu()
    torch.testing.assert_close(got.float(), lin - b.float(), rtol=2 ** -7, atol=2e-2)


@pytest.mark.parametrize("M,N,K,Z", [(300, 128, 64, 1), (129, 256, 512, 1), (1000, 64, 512, 2), (257, 2048, 512, 1),
                                     (128, 512, 2048, 1), (77, 512, 512, 6), (5, 8, 64, 1)])
@pytest.mark.parametrize("act", ["none", "silu", "tanh", "relu"])
def test_gemm_bf16_hand_written(hip, M, N, K, Z, act):
    """pafc_gemm_bf16 (hand-written MFMA GEMM) vs fp32 torch: tails in M and N, batching, every epilogue."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16
    bf = torch.bfloat16
    shp = (lambda *s: (Z,) + s) if Z > 1 else (lambda *s: s)
    a = synth.randn(shp(M, K), 1).to(bf)
    w = synth.randn(shp(N, K), 2, 0.05).to(bf)
    b = synth.randn(shp(N), 3, 0.2).to(bf)
    r = synth.randn(shp(M, N), 4).to(bf)
    f = {"none": lambda t: t, "silu": F.silu, "tanh": torch.tanh, "relu": F.relu}[act]
    lin = torch.matmul(a.float(), w.float().transpose(-1, -2))
    bb = b.float().unsqueeze(-2) if Z > 1 else b.float()
    tol = dict(rtol=2 ** -7, atol=2e-2)
    got = gemm_bf16(a.cuda(), w.cuda(), b.cuda(), act)
    torch.testing.assert_close(got.cpu().float(), f(lin + bb), **tol)
    got = gemm_bf16(a.cuda(), w.cuda(), None, act, alpha=0.5, residual=r.cuda())
    torch.testing.assert_close(got.cpu().float(), f(0.5 * lin) + r.float(), **tol)
    buf = r.cuda().clone()
    assert gemm_bf16(a.cuda(), w.cuda(), b.cuda(), "none", residual=buf, out=buf).data_ptr() == buf.data_ptr()
    torch.testing.assert_close(buf.cpu().float(), lin + bb + r.float(), **tol)


@pytest.mark.parametrize("tile", ["128x128", "128x64", "64x64", "auto"])
@pytest.mark.parametrize("M,N,K,Z,act", [(3992, 512, 2048, 1, "none"), (3992, 2048, 512, 1, "silu"), (2056, 512, 512, 6, "none"),
                                         (1235, 512, 1024, 1, "tanh"), (4160, 64, 512, 2, "none"), (999, 520, 192, 1, "relu")])
def test_gemm_bf16_tile_variants_mid_rows(hip, monkeypatch, tile, M, N, K, Z, act):
    """The 128 x 128 kernel's tile variants (128 x 64, 64 x 64: a few thousand rows -- a batch of 2 000-frame windows, 16-64
    streams -- would leave CUs idle on 128 x 128 tiles), each forced through PAFC_GEMM_TILE and as the dispatcher picks them,
    plain and residual epilogues, against the fp32 product of the same bf16 operands."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16
    if tile != "auto":
        monkeypatch.setenv("PAFC_GEMM_TILE", tile)
    bf = torch.bfloat16
    shp = (lambda *s: (Z,) + s) if Z > 1 else (lambda *s: s)
    a = synth.randn(shp(M, K), 1).to(bf).cuda()
    w = (synth.randn(shp(N, K), 2) / K ** 0.5).to(bf).cuda()
    b = (synth.randn(shp(N), 3) * 0.2).to(bf).cuda()
    r = synth.randn(shp(M, N), 4).to(bf).cuda()
    f = {"none": lambda t: t, "silu": F.silu, "tanh": torch.tanh, "relu": F.relu}[act]
    lin = torch.matmul(a.float(), w.float().transpose(-1, -2))
    bb = b.float().unsqueeze(-2) if Z > 1 else b.float()
    tol = dict(rtol=2 ** -7, atol=1e-2)
    torch.testing.assert_close(gemm_bf16(a, w, b, act).float(), f(lin + bb), **tol)
    torch.testing.assert_close(gemm_bf16(a, w, None, act, alpha=0.5, residual=r).float(), f(0.5 * lin) + r.float(), **tol)
    # nothing is written past the matrix: rows beyond M of a larger buffer keep their sentinel
    buf = torch.full(shp(M + 70, N), 7.0, dtype=bf, device="cuda")
    view = buf[..., :M, :]
    if Z == 1:
        gemm_bf16(a, w, b, act, out=view)
        assert bool((buf[M:] == 7.0).all())


def test_gemm_bf16_strided_rows_and_errors(hip):
    from paper_accurate_fast_cheap_amd._lib import PafcError
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16
    bf = torch.bfloat16
    wide = synth.randn((200, 256), 5).to(bf).cuda()
    a = wide[:, 64:192]                                   # row stride 256, offset 128 B: still 16-byte aligned rows
    w = synth.randn((128, 128), 6, 0.05).to(bf).cuda()
    outw = torch.zeros(200, 384, dtype=bf, device="cuda")
    gemm_bf16(a, w, out=outw[:, 128:256])
    torch.testing.assert_close(outw[:, 128:256].cpu().float(), a.cpu().float() @ w.cpu().float().t(), rtol=2 ** -7, atol=2e-2)
    assert float(outw[:, :128].abs().max()) == 0 and float(outw[:, 256:].abs().max()) == 0
    with pytest.raises(PafcError):
        gemm_bf16(a, synth.randn((128, 100), 7).to(bf).cuda())          # K mismatch
    with pytest.raises(PafcError):
        gemm_bf16(wide[:, :96], synth.randn((128, 96), 7).to(bf).cuda())  # K % 64
    with pytest.raises(PafcError):
        gemm_bf16(a.float(), w.float())


@pytest.mark.parametrize("M", [1, 16, 64, 78, 130, 640])
def test_decay_lora_skinny_equals_its_two_launches(hip, M):
    """The chunk step's decay LoRA in one launch (hidden tile in LDS) against the two few-rows GEMMs it replaces: bit-identical
    (same K split, same order of the partial sums, same roundings), with and without the time_decay bias; and against the fp32
    op chain with the reference's roundings (src/model.py:286-289)."""
    from paper_accurate_fast_cheap_amd import hip_ops
    C, H, bf = 512, 64, torch.bfloat16
    x = synth.randn((M, C), 40 + M, 1.0).to(bf).cuda()
    d1n = (synth.randn((H, C), 41, 0.05)).to(bf).cuda()
    d2n = (synth.randn((C, H), 42, 0.3)).to(bf).cuda()
    bias = synth.randn((C,), 43, 1.0).to(bf).cuda()
    for b in (bias, None):
        one = hip_ops.decay_lora_skinny(x, d1n, d2n, b)
        td = hip_ops.gemm_skinny(x, d1n, None, "tanh")
        two = hip_ops.gemm_skinny(td, d2n, b, round_first=b is not None)
        assert torch.equal(one, two)
    t_ref = torch.tanh(x.float() @ d1n.float().t()).to(bf)
    ref = ((t_ref.float() @ d2n.float().t()).to(bf).float() + bias.float()).to(bf)
    torch.testing.assert_close(hip_ops.decay_lora_skinny(x, d1n, d2n, bias).float(), ref.float(), rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("M,N,K,Z", [(64, 2048, 512, 1), (65, 512, 2048, 1), (78, 512, 512, 3), (1, 512, 512, 1), (17, 64, 1024, 2),
                                     (130, 512, 512, 1), (160, 2048, 512, 1)])
@pytest.mark.parametrize("act", ["none", "silu", "tanh", "relu"])
def test_gemm_skinny_streaming_chunk_shapes(hip, M, N, K, Z, act):
    """pafc_gemm_skinny_bf16 (the few-rows GEMM of the streaming chunk step) vs fp32 torch: row tails, more rows than one
    row group, batching, every epilogue, in-place residual."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_skinny
    bf = torch.bfloat16
    shp = (lambda *s: (Z,) + s) if Z > 1 else (lambda *s: s)
    a = synth.randn(shp(M, K), 1).to(bf)
    w = synth.randn(shp(N, K), 2, 0.05).to(bf)
    b = synth.randn(shp(N), 3, 0.2).to(bf)
    r = synth.randn(shp(M, N), 4).to(bf)
    f = {"none": lambda t: t, "silu": F.silu, "tanh": torch.tanh, "relu": F.relu}[act]
    lin = torch.matmul(a.float(), w.float().transpose(-1, -2))
    bb = b.float().unsqueeze(-2) if Z > 1 else b.float()
    tol = dict(rtol=2 ** -7, atol=2e-2)
    got = gemm_skinny(a.cuda(), w.cuda(), b.cuda(), act)
    torch.testing.assert_close(got.cpu().float(), f(lin + bb), **tol)
    got = gemm_skinny(a.cuda(), w.cuda(), None, act, alpha=0.5, residual=r.cuda())
    torch.testing.assert_close(got.cpu().float(), f(0.5 * lin + r.float()), **tol)
    buf = r.cuda().clone()
    assert gemm_skinny(a.cuda(), w.cuda(), b.cuda(), "none", residual=buf, out=buf).data_ptr() == buf.data_ptr()
    torch.testing.assert_close(buf.cpu().float(), lin + bb + r.float(), **tol)


@pytest.mark.parametrize("M", [64, 78, 16, 1])
def test_gemm_skinny_embedding_linear_k9728_vs_fp32(hip, M):
    """The chunk step's Linear(9728, 512) behind the subsampling (K split over eight waves, 304 K-steps) on the few-rows
    kernel, element-wise against the fp32 product of the same bf16 operands -- the largest K the kernel serves."""
    from paper_accurate_fast_cheap_amd.hip_ops import chunk_step, gemm_skinny, skinny_ok
    from tests import parity_log
    bf, K, N = torch.bfloat16, 9728, 512
    with chunk_step():
        assert skinny_ok(M, N, K)                  # the chunk step's dispatch sends this shape to the few-rows kernel
    a = F.relu(synth.randn((M, K), 11)).to(bf).cuda()                  # conv2's ReLU output: half zeros, non-negative
    w = (synth.randn((N, K), 12) / K ** 0.5).to(bf).cuda()
    b = (synth.randn((N,), 13) * 0.2).to(bf).cuda()
    got = gemm_skinny(a, w, b, "none")
    want = F.linear(a.float(), w.float(), b.float())
    d = (got.float() - want).abs()
    tol = 2 ** -7 * want.abs() + 1e-2
    parity_log.record(f"gemm_skinny K=9728 M={M}", max_abs_err=float(d.max()), mean_abs_err=float(d.mean()),
                      worst_err_over_tol=float((d / tol).max()), want_abs_max=float(want.abs().max()))
    assert got.shape == (M, N) and bool((d <= tol).all()), float(d.max())
    assert float(d.mean()) <= 2 ** -9 * float(want.abs().mean()) + 1e-4     # rounding of the output only: no K-split loss


@pytest.mark.parametrize("B,T", [(1, 64), (8, 64), (2, 37)])
def test_gemm_skinny_mix_and_norm_producers_at_c512_vs_fp32(hip, B, T):
    """The two operand PRODUCERS of the chunk step at the model's width (C = 512) against fp32 torch, not against other
    kernels: MIX -- token shift (carried frame in front) + first lerp + tanh(. W1), src/model.py:274-277 -- and NRM --
    silu(LayerNorm(.)) as pointwise_conv2's operand, convolution.py:136-141.  Reference rounding points (every op rounds to
    bf16) restated here in torch."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_skinny
    from tests import parity_log
    bf, C = torch.bfloat16, 512
    r16 = lambda t: t.to(bf).float()
    x = synth.randn((B, T, C), 21, 1.3).to(bf)
    prev = synth.randn((B, 1, C), 22, 1.3).to(bf)
    maa = torch.rand(C, generator=torch.Generator().manual_seed(23)).to(bf)
    w1 = (synth.randn((128, C), 24) * 0.08).to(bf)                      # time_maa_rkvw_w1^T
    xc = torch.cat([prev, x], 1).float()
    xx = r16(xc[:, :-1] - xc[:, 1:])                                     # model.py:274  xx = shift(x) - x
    xxx = r16(x.float() + r16(xx * maa.float()))                         # model.py:276
    want = torch.tanh(F.linear(xxx, w1.float())).view(B * T, 128)        # model.py:277 (fp32 product of the bf16 operand)
    got = gemm_skinny(x.view(B * T, C).cuda(), w1.cuda(), None, "tanh", mix_maa=maa.cuda(), mix_prev=prev.cuda(), mix_T=T)
    d = (got.float().cpu() - want).abs()
    parity_log.record(f"gemm_skinny MIX producer C=512 B={B} T={T}", max_abs_err=float(d.max()), mean_abs_err=float(d.mean()))
    assert float(d.max()) <= 2 ** -7 + 4e-3 and float(d.mean()) <= 1.5e-3, (float(d.max()), float(d.mean()))
    # NRM
    M = B * T
    a = (synth.randn((M, C), 25, 1.7) + 0.4).to(bf)
    gamma = (1 + 0.2 * synth.randn((C,), 26)).to(bf)
    beta = (0.1 * synth.randn((C,), 27)).to(bf)
    w = (synth.randn((C, C), 28) / C ** 0.5).to(bf)
    b = (synth.randn((C,), 29) * 0.2).to(bf)
    res = synth.randn((M, C), 30).to(bf)
    exact = F.linear(F.silu(F.layer_norm(a.float(), (C,), gamma.float(), beta.float(), 1e-5)), w.float(), b.float()) + res.float()
    got = gemm_skinny(a.cuda(), w.cuda(), b.cuda(), residual=res.cuda(), norm_silu=(gamma.cuda(), beta.cuda(), 1e-5))
    d = (got.float().cpu() - exact).abs()
    parity_log.record(f"gemm_skinny NRM producer C=512 M={M}", max_abs_err=float(d.max()), mean_abs_err=float(d.mean()))
    torch.testing.assert_close(got.float().cpu(), exact, rtol=2 ** -6, atol=4e-2)
    assert float(d.mean()) <= 6e-3


@pytest.mark.parametrize("M", [64, 78, 3, 150])
def test_gemm_skinny_glu_strided_rows_and_errors(hip, M):
    """act "glu" on the module's own (2C, K) weight (value rows then gate rows); strided operands; refused shapes."""
    from paper_accurate_fast_cheap_amd._lib import PafcError
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_skinny
    bf = torch.bfloat16
    K, N = 512, 1024
    wide = synth.randn((M, K + 128), 1).to(bf).cuda()
    a = wide[:, 64:64 + K]                                          # row stride K + 128, 16-byte aligned rows
    w = synth.randn((N, K), 2, 0.08).to(bf).cuda()
    b = synth.randn((N,), 3, 0.3).to(bf).cuda()
    want = F.glu(F.linear(a.cpu().float(), w.cpu().float(), b.cpu().float()), dim=-1)
    outw = torch.zeros(M, N, dtype=bf, device="cuda")
    got = gemm_skinny(a, w, b, act="glu", out=outw[:, N // 4:N // 4 + N // 2])
    assert got.shape == (M, N // 2)
    torch.testing.assert_close(got.cpu().float(), want, rtol=2 ** -7, atol=1e-2)
    assert float(outw[:, :N // 4].abs().max()) == 0 and float(outw[:, N // 4 + N // 2:].abs().max()) == 0
    with pytest.raises(PafcError):
        gemm_skinny(a[:, :48], w[:, :48].contiguous())             # K % 32
    with pytest.raises(PafcError):
        gemm_skinny(a, w[:1000].contiguous(), act="glu")           # GLU: N % 32
    with pytest.raises(PafcError):
        gemm_skinny(a, w, act="glu", residual=outw[:, :N // 2])    # GLU takes no residual
    with pytest.raises(PafcError):
        gemm_skinny(a.float(), w.float())


@pytest.mark.parametrize("M,N,act", [(64, 2048, "silu"), (78, 1024, "glu"), (130, 512, "none")])
def test_gemm_skinny_folded_layernorm_and_row_statistics(hip, M, N, act):
    """The LayerNorm in front of the projection folded into the launch (statistics from the partials a producer left), and
    the producer side: a residual GEMM that also writes the partial statistics of the rows it stores."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_skinny
    bf = torch.bfloat16
    C = 512
    x0 = synth.randn((M, C), 1, 1.5).to(bf).cuda()
    # producer: x = x0 + g W2^T, with statistics
    g_ = synth.randn((M, 512), 5).to(bf).cuda()
    w2 = synth.randn((C, 512), 6, 0.05).to(bf).cuda()
    st = torch.empty(M, C // 16, 2, dtype=torch.float32, device="cuda")
    x = gemm_skinny(g_, w2, None, residual=x0, stats_out=st)
    xf = x.float()
    torch.testing.assert_close(st[:, :, 0].sum(1), xf.sum(1), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(st[:, :, 1].sum(1), (xf * xf).sum(1), rtol=1e-4, atol=1e-2)
    # consumer: act(LN(x) W^T + b) with the LayerNorm folded in
    gamma = (1 + 0.2 * synth.randn((C,), 7)).float().cuda()
    beta = (0.1 * synth.randn((C,), 8)).float().cuda()
    w = synth.randn((N, C), 2, 0.05).to(bf).cuda()
    b = synth.randn((N,), 3, 0.2).to(bf).cuda()
    wp = (w.float() * gamma).to(bf)
    bp = (b.float() + w.float() @ beta).to(bf)
    cs = wp.float().sum(-1).contiguous()
    got = gemm_skinny(x, wp, bp, act, ln_stats=st, ln_csum=cs, ln_eps=1e-5)
    ln = F.layer_norm(xf, (C,), gamma, beta, 1e-5)
    lin = F.linear(ln, w.float(), b.float())
    want = {"silu": F.silu, "none": lambda t: t, "glu": lambda t: F.glu(t, dim=-1)}[act](lin)
    torch.testing.assert_close(got.float(), want, rtol=2 ** -6, atol=4e-2)


@pytest.mark.parametrize("B,T", [(1, 64), (2, 37), (3, 5)])
@pytest.mark.parametrize("with_prev", [False, True])
def test_gemm_skinny_token_shift_operand_equals_shift_mix_plus_gemm(hip, B, T, with_prev):
    """mix mode: the LoRA down-projection of a chunk in one launch = tmix_shift_mix (with the carried frame) + tanh GEMM."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_skinny, tmix_shift_mix
    bf = torch.bfloat16
    C, N = 512, 128
    x = synth.randn((B, T, C), 1).to(bf).cuda()
    maa = synth.randn((C,), 2, 0.5).to(bf).cuda()
    w1n = synth.randn((N, C), 3, 0.05).to(bf).cuda()
    prev = synth.randn((B, 1, C), 4).to(bf).cuda() if with_prev else None
    xxx = tmix_shift_mix(x, maa, None, prev=prev)[0]
    if with_prev:      # the definition: the carried frame stands in front of each sequence
        xc = torch.cat([prev, x], 1).float()
        xx = (xc[:, :-1] - xc[:, 1:]).to(bf).float()
        want_xxx = (x.float() + (xx * maa.float()).to(bf).float()).to(bf)
        assert torch.equal(xxx, want_xxx)
    want = gemm_bf16(xxx.view(B * T, C), w1n, None, "tanh")
    got = gemm_skinny(x.view(B * T, C), w1n, None, "tanh", mix_maa=maa, mix_prev=prev, mix_T=T)
    torch.testing.assert_close(got.float(), want.float(), rtol=2 ** -7, atol=1e-2)


@pytest.mark.parametrize("M", [64, 37, 200])
def test_gemm_skinny_layernorm_silu_operand_equals_the_two_passes(hip, M):
    """norm_silu: silu(LayerNorm(a)) formed in registers as the operand == add_layernorm(silu=True) followed by the GEMM."""
    from paper_accurate_fast_cheap_amd.hip_ops import add_layernorm, gemm_skinny
    bf = torch.bfloat16
    C = 512
    a = synth.randn((M, C), 1, 1.7).to(bf).cuda()
    gamma = (1 + 0.2 * synth.randn((C,), 7)).to(bf).cuda()
    beta = (0.1 * synth.randn((C,), 8)).to(bf).cuda()
    w = synth.randn((C, C), 2, 0.05).to(bf).cuda()
    b = synth.randn((C,), 3, 0.2).to(bf).cuda()
    r = synth.randn((M, C), 9).to(bf).cuda()
    _, g, _ = add_layernorm(a, None, 1.0, gamma, beta, silu=True, eps=1e-5)
    want = gemm_skinny(g, w, b, residual=r)
    got = gemm_skinny(a, w, b, residual=r, norm_silu=(gamma, beta, 1e-5))
    torch.testing.assert_close(got.float(), want.float(), rtol=2 ** -7, atol=2e-2)
    exact = F.linear(F.silu(F.layer_norm(a.float(), (C,), gamma.float(), beta.float(), 1e-5)), w.float(), b.float()) + r.float()
    torch.testing.assert_close(got.float(), exact, rtol=2 ** -6, atol=4e-2)


@pytest.mark.parametrize("M,N,K", [(64, 512, 512), (70, 2048, 512), (33, 512, 2048)])
def test_gemm_skinny_layernorm_from_its_own_operand_and_short_k(hip, M, N, K):
    """ln_self: the folded LayerNorm's statistics come from the operand fragments of the launch itself; round_first and a
    K of two steps (the decay LoRA's second product)."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_skinny
    bf = torch.bfloat16
    x = synth.randn((M, K), 1, 1.5).to(bf).cuda()
    r = synth.randn((M, N), 9).to(bf).cuda()
    gamma = (1 + 0.2 * synth.randn((K,), 7)).float().cuda()
    beta = (0.1 * synth.randn((K,), 8)).float().cuda()
    w = synth.randn((N, K), 2, 0.05).to(bf).cuda()
    wp = (w.float() * gamma).to(bf)
    bp = (w.float() @ beta).to(bf)
    got = gemm_skinny(x, wp, bp, residual=r, ln_self=True, ln_csum=wp.float().sum(-1).contiguous(), ln_eps=1e-5)
    want = F.linear(F.layer_norm(x.float(), (K,), gamma, beta, 1e-5), w.float()) + r.float()
    torch.testing.assert_close(got.float(), want, rtol=2 ** -6, atol=4e-2)
    # round_first with K = 64
    t = synth.randn((M, 64), 3).to(bf).cuda()
    d2 = synth.randn((N, 64), 4, 0.1).to(bf).cuda()
    b = synth.randn((N,), 5).to(bf).cuda()
    got = gemm_skinny(t, d2, b, round_first=True)
    want = (b.float() + (t.float() @ d2.float().t()).to(bf).float()).to(bf)
    diff = (got.float() - want.float()).abs()
    assert float(diff.max()) <= 2 ** -6 * float(want.float().abs().max())       # one ulp where the fp32 sums differ in the last bit
    assert float((diff > 0).float().mean()) < 0.02


@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (129, 1024, 512), (64, 256, 128)])
def test_gemm_bf16_glu_epilogue(hip, M, N, K):
    """act "glu": Linear -> F.glu as one GEMM, with the caller interleaving value / gate rows (glu_interleave)."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, glu_interleave
    bf = torch.bfloat16
    a = synth.randn((M, K), 1).to(bf)
    w = synth.randn((N, K), 2, 0.08).to(bf)
    b = synth.randn((N,), 3, 0.3).to(bf)
    want = F.glu(F.linear(a.float(), w.float(), b.float()), dim=-1)
    got = gemm_bf16(a.cuda(), glu_interleave(w.cuda()), glu_interleave(b.cuda()), act="glu")
    assert got.shape == (M, N // 2)
    torch.testing.assert_close(got.cpu().float(), want, rtol=2 ** -7, atol=1e-2)
    got = gemm_bf16(a.cuda(), glu_interleave(w.cuda()), None, act="glu")
    torch.testing.assert_close(got.cpu().float(), F.glu(F.linear(a.float(), w.float()), dim=-1), rtol=2 ** -7, atol=1e-2)


@pytest.mark.parametrize("B,T,Fd,C", [(1, 7, 80, 512), (3, 64, 80, 256), (2, 33, 23, 128), (1, 1003, 80, 512)])
def test_conv3x3s2_c1_nhwc(hip, B, T, Fd, C):
    """First subsampling convolution (1 channel in) as the direct NHWC kernel vs F.conv2d in fp32."""
    from paper_accurate_fast_cheap_amd.hip_ops import conv3x3s2_c1_nhwc
    bf = torch.bfloat16
    x = synth.randn((B, T, Fd), 1).to(bf)
    w = synth.randn((C, 1, 3, 3), 2, 0.3).to(bf)
    b = synth.randn((C,), 3, 0.2).to(bf)
    want = F.relu(F.conv2d(x.float().unsqueeze(1), w.float(), b.float(), stride=2)).permute(0, 2, 3, 1)
    got = conv3x3s2_c1_nhwc(x.cuda(), w.cuda(), b.cuda(), relu=True)
    assert got.shape == want.shape
    torch.testing.assert_close(got.cpu().float(), want, rtol=2 ** -7, atol=1e-2)
    got = conv3x3s2_c1_nhwc(x.cuda(), w.cuda(), None, relu=False)
    want = F.conv2d(x.float().unsqueeze(1), w.float(), None, stride=2).permute(0, 2, 3, 1)
    torch.testing.assert_close(got.cpu().float(), want, rtol=2 ** -7, atol=1e-2)


@pytest.mark.parametrize("B,T,Fd,C", [(1, 35, 80, 128), (2, 131, 80, 256), (1, 403, 40, 512)])
def test_conv_sub_f32split_matches_fp32_convolutions(hip, B, T, Fd, C):
    """fp32 activations through the bf16 matrix cores with hi + lo split operands: both subsampling convolutions vs
    F.conv2d in fp32 (float64 reference), error ~1e-5 relative -- two orders inside the 1e-3 parity bar."""
    from paper_accurate_fast_cheap_amd.hip_ops import conv_sub_f32split, split_bf16
    x = synth.randn((B, T, Fd), 1, 2.0)
    w1 = synth.randn((C, 1, 3, 3), 2, 0.3)
    b1 = synth.randn((C,), 3, 0.2)
    w2 = synth.randn((C, C, 3, 3), 4, 0.02)
    b2 = synth.randn((C,), 5, 0.2)
    y1 = F.relu(F.conv2d(x.double().unsqueeze(1), w1.double(), b1.double(), stride=2))
    want = F.relu(F.conv2d(y1, w2.double(), b2.double(), stride=2)).permute(0, 2, 3, 1).float()
    taps = w2.permute(2, 3, 0, 1).reshape(9, C, C).contiguous().cuda()
    hi, lo = split_bf16(taps)
    got = conv_sub_f32split(x.cuda(), w1.cuda(), b1.cuda(), hi, lo, b2.cuda()).cpu()
    assert got.shape == want.shape and got.dtype == torch.float32
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-4 * scale, float((got - want).abs().max()) / scale
    # and the fp32 framework convolution itself is not closer to float64 by more than an order of magnitude
    fw = F.relu(F.conv2d(F.relu(F.conv2d(x.cuda().unsqueeze(1), w1.cuda(), b1.cuda(), stride=2)), w2.cuda(), b2.cuda(),
                         stride=2)).permute(0, 2, 3, 1).cpu()
    assert float((got - want).abs().max()) <= 30 * max(float((fw - want).abs().max()), 1e-7 * scale)


@pytest.mark.parametrize("R,M,N", [(64, 128, 128), (1, 8, 8), (63, 136, 72), (1000, 512, 512), (4097, 2048, 512),
                                   (16000, 512, 2048), (300, 5000, 512),
                                   # round 6 (ring of four stages, two K groups): every ring fill 1 .. 5 K-steps with a ragged
                                   # last one, and the LoRA shapes of the training step
                                   (129, 2048, 2048), (200, 2048, 2048), (257, 2048, 2048), (15392, 512, 128),
                                   (15392, 64, 512), (15392, 512, 512)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_gemm_tn_weight_gradient(hip, R, M, N, out_dtype):
    """dw = dy^T x (nn.Linear's weight gradient): every tile / split / ragged-R / ragged-column path against a float64
    product of the same bf16 operands; the fp32 result carries no bf16 rounding."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_tn
    dy = synth.randn((R, M), 11, 1.0).to(torch.bfloat16)
    x = synth.randn((R, N), 12, 1.0).to(torch.bfloat16)
    ref = dy.double().t() @ x.double()
    got = gemm_tn(dy.cuda(), x.cuda(), out_dtype).cpu()
    assert got.dtype == out_dtype and got.shape == (M, N)
    scale = float(ref.abs().max())
    tol = 2e-5 * scale * max(1.0, (R / 1000) ** 0.5) if out_dtype == torch.float32 else 2 ** -7 * scale
    assert float((got.double() - ref).abs().max()) <= tol
    got2, db = gemm_tn(dy.cuda(), x.cuda(), out_dtype, want_bias=True)     # + the bias gradient from the same pass
    assert torch.equal(got2.cpu(), got) and db.shape == (M,) and db.dtype == out_dtype
    rb = dy.double().sum(0)
    btol = (2e-5 if out_dtype == torch.float32 else 2 ** -7) * max(1.0, float(rb.abs().max()))
    assert float((db.cpu().double() - rb).abs().max()) <= btol


@pytest.mark.parametrize("R,M,N,Z", [(300, 128, 128, 3), (1924, 512, 512, 3), (777, 136, 72, 2), (15392, 512, 512, 3)])
def test_gemm_tn_batched(hip, R, M, N, Z):
    """pafc_gemm_tn_bf16_batched (round 6): Z products of one shape in one launch pair -- strided batch entries (slices of wider
    buffers), the bias gradients of every entry from the same pass, bit-identical to Z single calls (same splits per entry only when
    the plan agrees, so: against float64, and repeatable)."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_tn
    dyw = synth.randn((Z + 1, R, M + 8), 21, 1.0).to(torch.bfloat16).cuda()
    xw = synth.randn((Z, R, N), 22, 1.0).to(torch.bfloat16).cuda()
    dy, x = dyw[:Z, :, :M], xw
    dw, db = gemm_tn(dy, x, torch.float32, want_bias=True)
    assert dw.shape == (Z, M, N) and db.shape == (Z, M)
    for z in range(Z):
        ref = dy[z].double().t() @ x[z].double()
        assert float((dw[z].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) * max(1.0, (R / 1000) ** 0.5)
        rb = dy[z].double().sum(0)
        assert float((db[z].double() - rb).abs().max()) <= 2e-5 * max(1.0, float(rb.abs().max()))
    again = gemm_tn(dy, x, torch.float32)
    assert torch.equal(again, dw)
    lo = gemm_tn(dy, x, torch.bfloat16)
    assert lo.dtype == torch.bfloat16 and float((lo.double() - dw.double()).abs().max()) <= 2 ** -7 * float(dw.abs().max())


def test_gemm_tn_strided_operands_and_determinism(hip):
    """Column slices of wider activations (row stride > width) and bit-identical repeats (fixed summation order)."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_tn
    wide_a = synth.randn((777, 1024), 13, 1.0).to(torch.bfloat16).cuda()
    wide_b = synth.randn((777, 640), 14, 1.0).to(torch.bfloat16).cuda()
    dy, x = wide_a[:, 256:768], wide_b[:, 64:576]
    a = gemm_tn(dy, x)
    b = gemm_tn(dy, x)
    assert torch.equal(a, b)
    ref = dy.double().t() @ x.double()
    assert float((a.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_linear_train_matches_autograd(hip):
    """The training Linear (library forward / dgrad, hand-written wgrad): fp32 master weights under bf16 autocast and a
    bf16 module, against float64 autograd on the same rounded operands."""
    from paper_accurate_fast_cheap_amd.hip_ops import linear_train, linear_train_eligible
    x = synth.randn((4, 300, 512), 15, 1.0).cuda().requires_grad_()
    w = synth.randn((2048, 512), 16, 0.05).cuda().requires_grad_()
    b = synth.randn((2048,), 17, 0.1).cuda().requires_grad_()
    gy = synth.randn((4, 300, 2048), 18, 1.0).cuda()
    assert not linear_train_eligible(x, w)                      # fp32 outside autocast: not this path
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert linear_train_eligible(x, w)
        y = linear_train(x, w, b)
    y.backward(gy.to(y.dtype))
    assert y.dtype == torch.bfloat16 and w.grad.dtype == torch.float32 and x.grad.dtype == torch.float32
    xr = x.detach().to(torch.bfloat16).double().cpu().requires_grad_()
    wr = w.detach().to(torch.bfloat16).double().cpu().requires_grad_()
    br = b.detach().to(torch.bfloat16).double().cpu().requires_grad_()
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(gy.to(torch.bfloat16).double().cpu())
    assert float((y.double().cpu() - yr).abs().max()) <= 2 ** -7 * float(yr.abs().max())
    assert float((w.grad.double().cpu() - wr.grad).abs().max()) <= 1e-4 * float(wr.grad.abs().max())
    assert float((b.grad.double().cpu() - br.grad).abs().max()) <= 1e-4 * float(br.grad.abs().max())
    assert float((x.grad.double().cpu() - xr.grad).abs().max()) <= 2 ** -6 * float(xr.grad.abs().max())
    xb = x.detach().to(torch.bfloat16).requires_grad_()
    wb = w.detach().to(torch.bfloat16).requires_grad_()
    assert linear_train_eligible(xb, wb)
    linear_train(xb, wb, None).backward(gy.to(torch.bfloat16))
    assert wb.grad.dtype == torch.bfloat16
    assert float((wb.grad.double().cpu() - wr.grad).abs().max()) <= 2 ** -7 * float(wr.grad.abs().max())


@pytest.mark.parametrize("xd,yd", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                   (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("rows,C", [(1, 8), (5, 512), (64, 512), (1000, 512), (257, 1024), (130, 136)])
def test_layernorm_backward_kernel(hip, xd, yd, rows, C):
    """dx, dgamma, dbeta of LayerNorm against float64 autograd on the same operands (every dtype pairing the training
    step produces: fp32 norm with fp32 or bf16 output, bf16 norm)."""
    from paper_accurate_fast_cheap_amd.hip_ops import layernorm_bwd
    x = synth.randn((rows, C), 31, 1.5).to(xd)
    g = (1 + 0.3 * synth.randn((C,), 32, 1.0)).to(xd)
    dy = synth.randn((rows, C), 33, 1.0).to(yd)
    xr = x.double().requires_grad_()
    gr = g.double().requires_grad_()
    br = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-5).backward(dy.double())
    dx, dg, db = layernorm_bwd(x.cuda(), dy.cuda(), g.cuda(), 1e-5)
    assert dx.dtype == xd and dg.dtype == db.dtype == torch.float32
    lo = xd == torch.bfloat16
    assert float((dx.cpu().double() - xr.grad).abs().max()) <= (2 ** -7 if lo else 2e-5) * max(1.0, float(xr.grad.abs().max()))
    assert float((dg.cpu().double() - gr.grad).abs().max()) <= 2e-5 * max(1.0, float(gr.grad.abs().max())) * max(1.0, rows ** 0.5 / 8)
    assert float((db.cpu().double() - br.grad).abs().max()) <= 2e-5 * max(1.0, float(br.grad.abs().max())) * max(1.0, rows ** 0.5 / 8)


def test_layernorm_module_training_path(hip):
    """The LayerNorm module under autograd on the GPU: kernels' result == nn.LayerNorm's, with and without bf16
    autocast; under autocast a norm whose consumer casts writes bf16 and still hands fp32 gradients to fp32 leaves."""
    from paper_accurate_fast_cheap_amd.transformer.layer_norm import LayerNorm
    torch.manual_seed(0)
    m = LayerNorm(512, eps=1e-5).cuda()
    with torch.no_grad():
        m.weight.normal_(1.0, 0.2); m.bias.normal_(0.0, 0.2)
    ref = torch.nn.LayerNorm(512, eps=1e-5).cuda()
    ref.load_state_dict(m.state_dict())
    x = synth.randn((3, 50, 512), 41, 1.5).cuda().requires_grad_()
    xr = x.detach().clone().requires_grad_()
    gy = synth.randn((3, 50, 512), 42, 1.0).cuda()
    m(x).backward(gy)
    ref(xr).backward(gy)
    torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(m.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(m.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-4)
    m.zero_grad(); x.grad = None
    m.consumer_casts = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
        yr = ref(xr)
    assert y.dtype == torch.bfloat16 and yr.dtype == torch.float32
    # the same values up to the last bf16 bit (the two kernels sum a row in different orders)
    torch.testing.assert_close(y.float(), yr.to(torch.bfloat16).float(), rtol=2 ** -7, atol=2 ** -8)
    y.backward(gy.to(torch.bfloat16))
    assert x.grad.dtype == torch.float32 and m.weight.grad.dtype == torch.float32
    with torch.no_grad():
        assert m(x).dtype == torch.float32        # no autograd: nn.LayerNorm itself


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("B,T,C", [(1, 1, 64), (2, 37, 128), (3, 50, 512), (1, 21, 1024)])
def test_tmix_elementwise_backward_kernels(hip, dtype, reverse, B, T, C):
    """shift_mix_train / mix4_train (forward kernels + pafc_tmix_*_bwd) against float64 autograd through the op-by-op
    chain of src/model.py:274-284 on the same operands."""
    import torch.nn.functional as F
    from paper_accurate_fast_cheap_amd.hip_ops import mix4_train, shift_mix_train
    x = synth.randn((B, T, C), 51, 1.0).to(dtype)
    maa_x = torch.rand(1, 1, C).to(dtype)
    maa4 = torch.rand(4, C).to(dtype)
    m = (0.3 * synth.randn((4, B, T, C), 52, 1.0)).to(dtype)
    g1 = synth.randn((B, T, C), 53, 1.0).to(dtype)
    g4 = [synth.randn((B, T, C), 54 + q, 1.0).to(dtype) for q in range(4)]

    def shift(t):
        return F.pad(t, (0, 0, -1, 1)) if reverse else F.pad(t, (0, 0, 1, -1))
    xr, ar, a4r, mr = (t.double().requires_grad_() for t in (x, maa_x, maa4, m))
    xx = shift(xr) - xr
    xxx_ref = xr + xx * ar
    z_ref = [xr + xx * (a4r[q].view(1, 1, C) + mr[q]) for q in range(4)]
    (xxx_ref * g1.double()).sum().backward(retain_graph=True)
    gx1, ga1 = xr.grad.clone(), ar.grad.clone()
    xr.grad = None
    sum((z_ref[q] * g4[q].double()).sum() for q in range(4)).backward()

    xg, ag, a4g, mg = (t.cuda().requires_grad_() for t in (x, maa_x, maa4, m))
    xxx = shift_mix_train(xg, ag, reverse)
    lo = dtype == torch.bfloat16
    # bf16: the forward kernels round where the op-by-op chain rounds (three times); values reach |x| + |xx| ~ 8
    tol = dict(rtol=2 ** -6, atol=6e-2) if lo else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xxx.detach().cpu().double(), xxx_ref.detach(), **tol)
    xxx.backward(g1.cuda())
    torch.testing.assert_close(xg.grad.cpu().double(), gx1, **tol)
    assert ag.grad.shape == maa_x.shape
    assert float((ag.grad.cpu().double() - ga1).abs().max()) <= (2 ** -6 if lo else 1e-4) * max(1.0, float(ga1.abs().max()))
    xg.grad = None
    z = mix4_train(xg, mg, a4g, reverse)
    for q in range(4):
        torch.testing.assert_close(z[q].detach().cpu().double(), z_ref[q].detach(), **tol)
    sum((z[q] * g4[q].cuda()).sum() for q in range(4)).backward()
    tol4 = dict(rtol=2 ** -5, atol=1e-1) if lo else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, **tol4)
    torch.testing.assert_close(mg.grad.cpu().double(), mr.grad, **tol)
    assert float((a4g.grad.cpu().double() - a4r.grad).abs().max()) <= (2 ** -6 if lo else 1e-4) * max(1.0, float(a4r.grad.abs().max()))


@pytest.mark.parametrize("bf16_out", [False, True])
def test_layernorm_with_skip_adds_the_residual_gradient_in_its_backward(hip, bf16_out):
    """layer_norm_with_skip (round 6): a pre-norm residual branch x + f(norm(x)) takes its residual input from the norm's second
    output, and the gradient of that path is added inside the norm's backward kernel (pafc_layernorm_bwd_add) -- the same
    gradients as norm + autograd's accumulation, fp32 stream, fp32 or bf16 norm output."""
    import torch.nn.functional as F
    from paper_accurate_fast_cheap_amd.hip_ops import layer_norm_with_skip
    C = 512
    x = synth.randn((3, 77, C), 95, 1.0)
    g = 1.0 + 0.2 * synth.randn((C,), 96, 1.0)
    b = 0.1 * synth.randn((C,), 97, 1.0)
    w = synth.randn((C, C), 98, 1.0) / C ** 0.5
    dy = synth.randn((3, 77, C), 99, 1.0)
    xr, gr, br = (t.double().requires_grad_() for t in (x, g, b))
    out_ref = xr + 0.5 * torch.tanh(F.layer_norm(xr, (C,), gr, br, 1e-5) @ w.double())
    out_ref.backward(dy.double())
    xg, gg, bg = (t.cuda().requires_grad_() for t in (x, g, b))
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16_out):
        got = layer_norm_with_skip(xg, gg, bg, 1e-5, bf16_out=bf16_out)
        assert got is not None
        h, xs = got
        assert h.dtype == (torch.bfloat16 if bf16_out else torch.float32) and xs.dtype == torch.float32 and xs.data_ptr() == xg.data_ptr()
    out = xs + 0.5 * torch.tanh(h.float() @ w.cuda())
    out.backward(dy.cuda())
    tol = dict(rtol=2e-2, atol=3e-2) if bf16_out else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.detach().cpu().double(), out_ref.detach(), **tol)
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, **tol)
    for got_g, want in ((gg.grad, gr.grad), (bg.grad, br.grad)):
        assert float((got_g.cpu().double() - want).abs().max()) <= (3e-2 if bf16_out else 1e-4) * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("xdtype,gdtype", [(torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)])
@pytest.mark.parametrize("B,T,C", [(1, 1, 64), (3, 50, 512), (2, 481, 512), (1, 37, 1024)])
def test_layernorm_silu_training_kernels(hip, xdtype, gdtype, B, T, C):
    """ln_silu_train (round 6: the conv module's `activation(norm(x))`, convolution.py:136-138, as one kernel each way) against
    float64 autograd through F.layer_norm + F.silu on the same (rounded) operands: bf16 x with the norm's fp32 parameters (bf16
    autocast over an fp32 model: the chain it replaces is cast, LayerNorm, SiLU, cast), bf16 with bf16 parameters, fp32."""
    import torch.nn.functional as F
    from paper_accurate_fast_cheap_amd.hip_ops import ln_silu_train, ln_silu_train_eligible
    x = synth.randn((B, T, C), 91, 1.5).to(xdtype)
    g = (1.0 + 0.3 * synth.randn((C,), 92, 1.0)).to(gdtype)
    b = (0.2 * synth.randn((C,), 93, 1.0)).to(gdtype)
    dy = synth.randn((B, T, C), 94, 1.0).to(xdtype)
    xr, gr, br = (t.double().requires_grad_() for t in (x, g, b))
    ref = F.silu(F.layer_norm(xr, (C,), gr, br, 1e-5))
    ref.backward(dy.double())
    xg, gg, bg = (t.cuda().requires_grad_() for t in (x, g, b))
    assert ln_silu_train_eligible(xg, gg, bg)
    y = ln_silu_train(xg, gg, bg, 1e-5)
    assert y.dtype == xdtype and y.shape == x.shape
    lo = xdtype == torch.bfloat16
    torch.testing.assert_close(y.detach().cpu().double(), ref.detach(), **(dict(rtol=2 ** -7, atol=2e-2) if lo else dict(rtol=1e-5, atol=1e-5)))
    y.backward(dy.cuda())
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, **(dict(rtol=2 ** -6, atol=3e-2) if lo else dict(rtol=1e-4, atol=1e-5)))
    for got, want in ((gg.grad, gr.grad), (bg.grad, br.grad)):
        assert got.dtype == gdtype and got.shape == (C,)
        tol = 2 ** -7 if gdtype == torch.bfloat16 else 1e-4
        assert float((got.cpu().double() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("wdtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,C", [(2, 150, 128), (4, 481, 512)])
def test_grouped_linear_training_function(hip, wdtype, B, T, C):
    """linear_group_train (round 6: the r / k / v projections of a time-mix block as ONE batched launch each way -- forward
    against a grouped bf16 copy of the three weights, input gradients against grouped transposed copies, the three weight
    gradients from one batched gemm_tn pair) against float64 autograd through three F.linear on the same bf16-rounded operands;
    the inputs are slices of one tensor as the lerp kernel leaves them, the incoming gradients slices of one tensor as the WKV
    backward leaves them (no stacking copy), and once more as separate tensors (stacked).  Stale copies: a second step after an
    in-place weight update must see the new weights."""
    import torch.nn.functional as F
    from paper_accurate_fast_cheap_amd import hip_ops
    z = synth.randn((4, B, T, C), 81, 1.0).to(torch.bfloat16).cuda()
    ws = [torch.nn.Parameter((synth.randn((C, C), 82 + i, 1.0) / C ** 0.5).to(wdtype).cuda()) for i in range(3)]
    g4 = synth.randn((4, B, T, C), 86, 1.0).to(torch.bfloat16).cuda()

    def reference(ws_now):
        zr = z[:3].double().detach().requires_grad_()
        wr = [w.detach().to(torch.bfloat16).double().requires_grad_() for w in ws_now]
        ys = [F.linear(zr[i], wr[i]) for i in range(3)]
        sum((ys[i] * g4[i].double()).sum() for i in range(3)).backward()
        return ys, zr.grad, [w.grad for w in wr]

    def check(sep_grads):
        for w in ws:
            w.grad = None
        zz = z.clone().requires_grad_()
        xs = tuple(zz[i] for i in range(3))
        with hip_ops.train_shadows():
            assert hip_ops.linear_group_train_eligible(xs, ws)
            ys = hip_ops.linear_group_train(xs, ws)
        gs = [g4[i].clone() for i in range(3)] if sep_grads else [g4[i] for i in range(3)]
        torch.autograd.backward(ys, gs)
        ry, rdz, rdw = reference(ws)
        for i in range(3):
            assert ys[i].shape == (B, T, C) and ys[i].dtype == torch.bfloat16
            torch.testing.assert_close(ys[i].detach().double(), ry[i].detach(), rtol=2 ** -7, atol=2e-2)
            torch.testing.assert_close(zz.grad[i].double(), rdz[i], rtol=2 ** -7, atol=2e-2)
            assert ws[i].grad.dtype == wdtype and ws[i].grad.shape == (C, C)
            tol = 2 ** -7 if wdtype == torch.bfloat16 else 2e-5 * max(1.0, (B * T / 1000) ** 0.5)
            assert float((ws[i].grad.double() - rdw[i]).abs().max()) <= tol * float(rdw[i].abs().max())
        assert float(zz.grad[3].abs().max()) == 0.0

    check(False)
    check(True)
    with torch.no_grad():
        for w in ws:
            w.mul_(-0.5)                  # an optimizer step: the kept copies are stale until train_shadows() is entered again
    hip_ops.bump_param_epoch()
    check(False)


@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("B,T,C", [(2, 150, 128), (3, 333, 512)])
def test_lora_up_and_lerps_on_own_kernels_each_way(hip, xdtype, reverse, B, T, C):
    """lora_mix4_train (round 6: the LoRA-up products of src/model.py:277-278 as one block-diagonal batched GEMM, their input
    gradient as one batched GEMM over dm's column blocks, their weight gradient as the diagonal blocks of one dy^T x product)
    against float64 autograd through torch.bmm + the four lerps on the same bf16-rounded operands; x fp32 (autocast over an fp32
    residual stream) and bf16 (the slot)."""
    import torch.nn.functional as F
    from paper_accurate_fast_cheap_amd.hip_ops import lora_mix4_train, lora_mix4_train_eligible
    R = 32
    x = synth.randn((B, T, C), 71, 1.0).to(xdtype)
    t = torch.tanh(synth.randn((B, T, 4 * R), 72, 1.0)).to(torch.bfloat16)
    w2 = (0.05 * synth.randn((4, R, C), 73, 1.0)).to(torch.bfloat16)
    maa4 = torch.rand(4, C).to(xdtype)
    g4 = [synth.randn((B, T, C), 74 + q, 1.0).to(xdtype) for q in range(4)]

    def shift(v):
        return F.pad(v, (0, 0, -1, 1)) if reverse else F.pad(v, (0, 0, 1, -1))
    xr, tr, wr, ar = (v.double().requires_grad_() for v in (x, t, w2, maa4))
    m_ref = torch.bmm(tr.view(B * T, 4, R).transpose(0, 1), wr).view(4, B, T, C)
    xx = shift(xr) - xr
    z_ref = [xr + xx * (ar[q].view(1, 1, C) + m_ref[q]) for q in range(4)]
    sum((z_ref[q] * g4[q].double()).sum() for q in range(4)).backward()

    xg, tg, wg = (v.cuda().requires_grad_() for v in (x, t, w2))
    # the four lerp coefficients as the module holds them: separate (1, 1, C) parameters (moved into one buffer on first use)
    maas = [torch.nn.Parameter(maa4[q].view(1, 1, C).clone().cuda()) for q in range(4)]
    assert lora_mix4_train_eligible(xg, tg, wg)
    z = lora_mix4_train(xg, tg, wg, maas, reverse)
    lo = xdtype == torch.bfloat16
    # m is rounded to bf16 between the product and the lerp (as torch.bmm's bf16 output is); |xx| reaches ~8
    tol = dict(rtol=2 ** -6, atol=8e-2) if lo else dict(rtol=2 ** -7, atol=4e-2)
    for q in range(4):
        torch.testing.assert_close(z[q].detach().cpu().double(), z_ref[q].detach(), **tol)
    sum((z[q] * g4[q].cuda()).sum() for q in range(4)).backward()
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, **(dict(rtol=2 ** -5, atol=1e-1) if lo else dict(rtol=2 ** -7, atol=4e-2)))
    # dt = dm W2^T: sums of C products of bf16-rounded dm; dW2 = t^T dm: sums of B T products, fp32 accumulation
    for got, ref in ((tg.grad, tr.grad), (wg.grad, wr.grad)):
        assert got.dtype == torch.bfloat16 and got.shape == ref.shape
        assert float((got.cpu().double() - ref).abs().max()) <= 2 ** -6 * float(ref.abs().max()) + 1e-3
    ag = torch.stack([m_.grad.reshape(C) for m_ in maas])
    assert all(m_.grad.shape == (1, 1, C) and m_.grad.dtype == xdtype for m_ in maas)
    assert float((ag.cpu().double() - ar.grad).abs().max()) <= (2 ** -6 if lo else 1e-4) * max(1.0, float(ar.grad.abs().max()))
    assert maas[1].data_ptr() - maas[0].data_ptr() == C * maas[0].element_size()          # one buffer now
    z2 = lora_mix4_train(xg, tg, wg, maas, reverse)                                        # ... found again without a copy
    torch.testing.assert_close(z2[0].detach(), z[0].detach(), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reverse", [False, True])
def test_tmix_block_training_path_equals_framework_autograd(hip, dtype, reverse, monkeypatch):
    """RWKV_Tmix_x060c.mix_project under autograd: the kernel path (shift_mix / mix4 / linear / matmul functions) gives
    the outputs and parameter gradients of the op-by-op framework path (PAFC_TRAIN_KERNELS=0)."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.tmix import RWKV_Tmix_x060c
    torch.manual_seed(4)
    blk = RWKV_Tmix_x060c(64, 12, 128, 128, 3)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if n.endswith("_w1") or n.endswith("_w2"):
                p.normal_(0, 0.1)
    blk = blk.to(dtype).cuda()
    x = synth.randn((2, 300, 128), 61, 1.0).to(dtype).cuda()
    gs = [synth.randn((2, 300, 128), 62 + i, 1.0).to(dtype).cuda() for i in range(4)]

    def run():
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_()
        outs = blk.mix_project(xi, reverse)
        sum((o * g).sum() for o, g in zip(outs, gs)).backward()
        return [o.detach().float() for o in outs], xi.grad.float(), {n: p.grad.float().clone() for n, p in blk.named_parameters()
                                                                      if p.grad is not None}
    o_k, gx_k, gp_k = run()
    monkeypatch.setenv("PAFC_TRAIN_KERNELS", "0")
    o_f, gx_f, gp_f = run()
    lo = dtype == torch.bfloat16
    for a, b in zip(o_k, o_f):
        torch.testing.assert_close(a, b, rtol=2 ** -6 if lo else 1e-4, atol=5e-2 if lo else 1e-4)
    scale = float(gx_f.abs().max())
    assert float((gx_k - gx_f).abs().max()) <= (0.05 if lo else 1e-3) * scale
    assert set(gp_k) == set(gp_f)
    for n in gp_f:
        s = max(float(gp_f[n].abs().max()), 1e-6)
        assert float((gp_k[n] - gp_f[n]).abs().max()) <= (0.08 if lo else 2e-3) * s, n


@pytest.mark.parametrize("tile_n,tile_m", [(256, 256), (256, 192), (256, 128), (256, 64)])
@pytest.mark.parametrize("M,N,K,Z,act", [(256, 256, 128, 1, "none"), (1000, 512, 512, 1, "silu"), (513, 264, 384, 1, "tanh"),
                                         (300, 1024, 256, 2, "relu"), (2049, 512, 1024, 1, "none"), (777, 2048, 512, 1, "silu"),
                                         (260, 512, 2048, 3, "none")])
def test_gemm_phase_pipelined(hip, tile_n, tile_m, M, N, K, Z, act):
    """csrc/gemm_ph.hip (tile_m x 256 tiles, 8 waves, counted LDS-DMA waits) vs fp32 torch: tails in M and N, K-step counts
    from the minimum of two up, several tiles per block (the cross-tile prefetch), batching, every epilogue including the
    in-place residual, row counts per tile from the full 256 down to one MFMA tile."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16_ph
    bf = torch.bfloat16
    shp = (lambda *s: (Z, *s)) if Z > 1 else (lambda *s: s)
    a = synth.randn(shp(M, K), 11).to(bf)
    w = synth.randn(shp(N, K), 12, 1.0 / K ** 0.5).to(bf)
    b = synth.randn(shp(N), 13, 0.3).to(bf)
    r = synth.randn(shp(M, N), 14).to(bf)
    lin = torch.matmul(a.float(), w.float().transpose(-1, -2))
    f = {"none": lambda t: t, "silu": F.silu, "tanh": torch.tanh, "relu": F.relu}[act]
    want = f(lin + (b.float().unsqueeze(-2) if Z > 1 else b.float()))
    got = gemm_bf16_ph(a.cuda(), w.cuda(), b.cuda(), act, tile_n=tile_n, tile_m=tile_m)
    torch.testing.assert_close(got.cpu().float(), want, rtol=2 ** -7, atol=2e-2)
    got = gemm_bf16_ph(a.cuda(), w.cuda(), None, "none", alpha=0.5, residual=r.cuda(), tile_n=tile_n, tile_m=tile_m)
    torch.testing.assert_close(got.cpu().float(), r.float() + 0.5 * lin, rtol=2 ** -7, atol=2e-2)
    buf = r.cuda().clone()
    same = gemm_bf16_ph(a.cuda(), w.cuda(), b.cuda(), "none", residual=buf, out=buf, tile_n=tile_n, tile_m=tile_m)
    assert same.data_ptr() == buf.data_ptr()
    torch.testing.assert_close(buf.cpu().float(), r.float() + lin + (b.float().unsqueeze(-2) if Z > 1 else b.float()),
                               rtol=2 ** -7, atol=2e-2)


@pytest.mark.parametrize("tile_n", [256])
@pytest.mark.parametrize("M,N,K", [(700, 1024, 512), (256, 256, 128), (1025, 512, 384)])
def test_gemm_phase_pipelined_glu(hip, tile_n, M, N, K):
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16_ph, glu_interleave
    bf = torch.bfloat16
    a = synth.randn((M, K), 21).to(bf)
    w = synth.randn((N, K), 22, 1.0 / K ** 0.5).to(bf)
    b = synth.randn((N,), 23, 0.3).to(bf)
    want = F.glu(F.linear(a.float(), w.float(), b.float()), dim=-1)
    half = tile_n // 8
    got = gemm_bf16_ph(a.cuda(), glu_interleave(w.cuda(), half), glu_interleave(b.cuda(), half), act="glu", tile_n=tile_n)
    assert got.shape == (M, N // 2)
    torch.testing.assert_close(got.cpu().float(), want, rtol=2 ** -7, atol=2e-2)




@pytest.mark.parametrize("K,N,Z,act,res,name", [
    (512, 2048, 1, "silu", False, "ffn w_1 + SiLU"), (2048, 512, 1, "none", True, "ffn w_2 + residual"),
    (512, 1024, 1, "glu", False, "pointwise_conv1 + GLU"), (512, 512, 1, "none", True, "pointwise_conv2 + residual"),
    (1024, 512, 1, "none", True, "slot output + residual"), (512, 512, 6, "none", False, "r,k,v stack"),
    (512, 5000, 1, "none", False, "CTC head")])
def test_gemm_at_the_production_shape(hip, K, N, Z, act, res, name):
    """The dispatching entry point (pafc_gemm_bf16 -> the phase-pipelined kernel) at the bench's row count, M = 44 998
    (176 row tiles, tile_m balancing, the descriptor extents), element-wise against an fp32 product of the same bf16
    operands computed on the GPU: every layer shape of the 30-minute pass and the CTC head."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16, gemm_glu_half, glu_interleave
    from tests import parity_log
    M, bf, dev = 44998, torch.bfloat16, "cuda"
    g = torch.Generator(device=dev).manual_seed(1234 + K + N)
    shp = (lambda *s: (Z, *s)) if Z > 1 else (lambda *s: s)
    a = torch.randn(shp(M, K), device=dev, generator=g).to(bf)
    w = (torch.randn(shp(N, K), device=dev, generator=g) / K ** 0.5).to(bf)
    b = None if Z > 1 else (torch.randn(N, device=dev, generator=g) * 0.3).to(bf)
    r = torch.randn(shp(M, N), device=dev, generator=g).to(bf) if res else None
    lin = torch.matmul(a.float(), w.float().transpose(-1, -2))
    if b is not None:
        lin = lin + b.float()
    alpha = 0.5 if res else 1.0
    if act == "glu":
        half = gemm_glu_half(M, N, K)
        got = gemm_bf16(a, glu_interleave(w, half), glu_interleave(b, half), act="glu")
        want = F.glu(lin, dim=-1)
    elif res:
        wb = alpha * (lin - (b.float() if b is not None else 0)) + (b.float() if b is not None else 0) + r.float()
        got = gemm_bf16(a, w, b, "none", alpha=alpha, residual=r)
        want = wb
    else:
        got = gemm_bf16(a, w, b, act)
        want = {"none": lambda t: t, "silu": F.silu}[act](lin)
    assert got.shape == want.shape and got.dtype == bf
    d = (got.float() - want).abs()
    tol = 2 ** -7 * want.abs() + 2e-2
    parity_log.record(f"gemm M=44998/{name}", max_abs_err=float(d.max()), mean_abs_err=float(d.mean()),
                      worst_err_over_tol=float((d / tol).max()), want_abs_max=float(want.abs().max()))
    assert bool((d <= tol).all()), (name, float(d.max()))
    # the last row tile is ragged (44 998 = 175 * 256 + 198): nothing may be written past the matrix
    if res:
        buf = torch.cat([r.reshape(-1, N), torch.full((64, N), 7.0, device=dev, dtype=bf)]) if Z == 1 else None
        if buf is not None:
            view = buf[:M]
            gemm_bf16(a, w, b, "none", alpha=alpha, residual=view, out=view)
            assert bool((buf[M:] == 7.0).all())
            torch.testing.assert_close(view, got)


def _planes_value(p):
    """(…, 2K) bf16 planes [hi | lo] -> the fp32 value hi + lo."""
    K = p.shape[-1] // 2
    return p[..., :K].float() + p[..., K:].float()


def test_layernorm_and_split_plane_outputs(hip):
    """PAFC_SPLIT_BF16 outputs: split_planes and the LayerNorm pass write an fp32 result as bf16 planes hi | lo whose sum is
    the fp32 value to 2^-16 relative (the A operand of the split-operand GEMM); the triple form is the weight's."""
    from paper_accurate_fast_cheap_amd.hip_ops import add_layernorm, split_planes
    x = (synth.randn((3, 37, 512), 5) * 3).cuda()
    p = split_planes(x)
    assert p.shape == (3, 37, 1024) and p.dtype == torch.bfloat16
    assert torch.equal(p[..., :512], x.bfloat16())
    torch.testing.assert_close(_planes_value(p), x, rtol=2 ** -15, atol=1e-30)
    w3 = split_planes(x[0], triple=True)
    assert w3.shape == (37, 1536) and torch.equal(w3[:, :512], w3[:, 512:1024])
    assert torch.equal(w3[:, :512], p[0][:, :512]) and torch.equal(w3[:, 1024:], p[0][:, 512:])
    g, b = synth.randn((512,), 6).cuda() + 1, synth.randn((512,), 7).cuda()
    g2, b2 = synth.randn((512,), 8).cuda() + 1, synth.randn((512,), 9).cuda()
    _, o1, o2 = add_layernorm(x, None, 1.0, g, b, gamma2=g2, beta2=b2)
    _, s1, s2 = add_layernorm(x, None, 1.0, g, b, gamma2=g2, beta2=b2, split1=True)
    _, f1, t2 = add_layernorm(x, None, 1.0, g, b, gamma2=g2, beta2=b2, split2=True)
    assert s1.shape == (3, 37, 1024) and s2.shape == (3, 37, 1024) and f1.dtype == torch.float32
    torch.testing.assert_close(_planes_value(s1), o1, rtol=2 ** -15, atol=1e-7)
    torch.testing.assert_close(_planes_value(s2), o2, rtol=2 ** -15, atol=1e-7)
    assert torch.equal(f1, o1) and torch.equal(t2, s2)
    _, sg, _ = add_layernorm(x, None, 1.0, g, b, silu=True, split1=True)
    _, og, _ = add_layernorm(x, None, 1.0, g, b, silu=True)
    torch.testing.assert_close(_planes_value(sg), og, rtol=2 ** -15, atol=1e-7)


@pytest.mark.parametrize("M,N,K", [(1000, 512, 256), (2500, 2048, 512), (301, 1024, 128), (499, 512, 2048), (999, 512, 2048),
                                   # round 6, late: long-K products on 128 x 128 tiles with 4 / 4 / 3 / 2 K shares per tile
                                   (1996, 512, 2048), (3992, 512, 2048), (5000, 512, 2048), (7984, 512, 2048)])
def test_gemm_split_operand_forms(hip, M, N, K):
    """pafc_gemm_ph_ex: fp32 activations and weights as bf16 planes (three bf16 products per fp32 product), fp32 / plane
    outputs, fp32 bias and residual, GLU -- against fp32 torch: the error is that of 16-bit significands, not of bf16.  Up to
    8 192 rows (and 2^22 outputs) the non-GLU forms run on the small tiles of csrc/gemm_bf16.hip (pafc_gemm_bf16_f32out); the
    K = 2048 shapes also take its K split over blocks with the reducing second launch (64 x 64 tiles at few rows, 128 x 128 tiles
    with 2-4 K shares from ~1 000 rows on)."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_ph_ex, glu_interleave, split_planes
    a = synth.randn((M, K), 31).cuda()
    w = (synth.randn((N, K), 32) / K ** 0.5).cuda()
    b = (synth.randn((N,), 33) * 0.3).cuda()
    r = synth.randn((M, N), 34).cuda()
    ap, w3 = split_planes(a), split_planes(w, triple=True)
    lin = a.double() @ w.double().t()
    tol = dict(rtol=1e-4, atol=1e-4)
    got = gemm_ph_ex(ap, w3, b, a_split=True, out_kind="f32")
    torch.testing.assert_close(got.double(), lin + b.double(), **tol)
    buf = r.clone()
    same = gemm_ph_ex(ap, w3, b, alpha=0.5, residual=buf, out=buf, a_split=True, out_kind="f32")
    assert same.data_ptr() == buf.data_ptr()
    torch.testing.assert_close(buf.double(), 0.5 * lin + b.double() + r.double(), **tol)
    got = gemm_ph_ex(ap, w3, b, "silu", a_split=True, out_kind="planes")
    assert got.shape == (M, 2 * N) and got.dtype == torch.bfloat16
    torch.testing.assert_close(_planes_value(got).double(), F.silu(lin + b.double()), **tol)
    got = gemm_ph_ex(ap, split_planes(glu_interleave(w, 32), triple=True), glu_interleave(b, 32), "glu", a_split=True, out_kind="f32")
    torch.testing.assert_close(got.double(), F.glu(lin + b.double(), dim=-1), **tol)
    # bf16 operands into an fp32 stream (the slot's output projection): exact products, fp32 residual and output
    ab, wb = a.bfloat16(), w.bfloat16()
    got = gemm_ph_ex(ab, wb, None, residual=r, out_kind="f32")
    torch.testing.assert_close(got.double(), ab.double() @ wb.double().t() + r.double(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M", [61, 499, 3992, 4500])
def test_gemm_split_plane_blocks_at_few_rows(hip, M):
    """Linear(F' C, odim) behind the subsampling convolutions at window row counts: A = conv2's plane output, [hi C | lo C] per
    frequency bin (a_plane_block = C), K = 19 x 512 = 9 728.  Round 6: up to 4 096 rows this runs on the small tiles
    (pafc_gemm_bf16_f32out_pb; with the K split over blocks at 61 and 499 rows) instead of 1-16 tiles of the 256-wide kernel;
    4 500 rows still take that one.  Against a float64 product of the same fp32 operands."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_ph_ex, split_planes
    Fp, C = 19, 512
    a = (synth.randn((M, Fp * C), 41) * 0.5).cuda()
    w = (synth.randn((512, Fp * C), 42) / (Fp * C) ** 0.5).cuda()
    b = (synth.randn((512,), 43) * 0.3).cuda()
    ap = split_planes(a.view(M * Fp, C)).view(M, Fp * 2 * C)
    got = gemm_ph_ex(ap, split_planes(w, triple=True), b, a_split=True, out_kind="f32", a_plane_block=C)
    torch.testing.assert_close(got.double(), a.double() @ w.double().t() + b.double(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("form", ["w_1 planes+SiLU", "w_2 residual", "w_2 residual in place", "pointwise_conv1 GLU",
                                  "pointwise_conv2 residual in place", "slot output bf16 operands", "CTC head",
                                  "Linear(9728,512) plane blocks"])
def test_gemm_split_forms_at_the_production_shape(hip, form):
    """Every split-operand form the headline (fp32 model + bf16 slot, fused.layer_forward_split / the subsampling Linear / the
    CTC head) launches, at ITS row count M = 44 998 (175 full 256-row tiles + a ragged one of 198 rows) -- through the same
    entry point with the same arguments -- against a float64 product of the same fp32 operands on the GPU.  Error of 16-bit
    significands (three bf16 products per fp32 product), not of bf16; 64 guard rows behind every output stay untouched."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_ph_ex, glu_interleave, split_planes
    from tests import parity_log
    M, dev = 44998, "cuda"
    g = torch.Generator(device=dev).manual_seed(4321)
    rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev, generator=g) * scale
    tol = dict(rtol=1e-4, atol=1e-4)

    def guarded(cols, dtype, fill=None):
        buf = torch.full((M + 64, cols), 7.0, device=dev, dtype=dtype)
        if fill is not None:
            buf[:M] = fill
        return buf

    def check(got, want, buf, name=form, **kw):
        d = (got.double() - want).abs()
        parity_log.record(f"split gemm M=44998/{name}", max_abs_err=float(d.max()), mean_abs_err=float(d.mean()),
                          want_abs_max=float(want.abs().max()), last_tile_max_abs_err=float(d[175 * 256:].max()))
        torch.testing.assert_close(got.double(), want, **(kw or tol))
        assert bool((buf[M:] == 7.0).all()), "rows behind the matrix were written"

    if form == "w_1 planes+SiLU":
        a, w, b = rnd(M, 512), rnd(2048, 512, scale=512 ** -0.5), rnd(2048, scale=0.3)
        buf = guarded(4096, torch.bfloat16)
        got = gemm_ph_ex(split_planes(a), split_planes(w, triple=True), b, "silu", out=buf[:M], a_split=True, out_kind="planes")
        check(_planes_value(got), F.silu(a.double() @ w.double().t() + b.double()), buf)
    elif form.startswith("w_2 residual"):
        a, w, b, r = rnd(M, 2048), rnd(512, 2048, scale=2048 ** -0.5), rnd(512, scale=0.3), rnd(M, 512)
        want = 0.5 * (a.double() @ w.double().t()) + b.double() + r.double()
        if form.endswith("in place"):
            buf = guarded(512, torch.float32, r)
            got = gemm_ph_ex(split_planes(a), split_planes(w, triple=True), b, alpha=0.5, residual=buf[:M], out=buf[:M], a_split=True, out_kind="f32")
        else:
            buf = guarded(512, torch.float32)
            got = gemm_ph_ex(split_planes(a), split_planes(w, triple=True), b, alpha=0.5, residual=r, out=buf[:M], a_split=True, out_kind="f32")
        check(got, want, buf)
    elif form == "pointwise_conv1 GLU":
        a, w, b = rnd(M, 512), rnd(1024, 512, scale=512 ** -0.5), rnd(1024, scale=0.3)
        buf = guarded(512, torch.float32)
        got = gemm_ph_ex(split_planes(a), split_planes(glu_interleave(w, 32), triple=True), glu_interleave(b, 32), "glu", out=buf[:M],
                         a_split=True, out_kind="f32")
        check(got, F.glu(a.double() @ w.double().t() + b.double(), dim=-1), buf)
    elif form == "pointwise_conv2 residual in place":
        a, w, b, r = rnd(M, 512), rnd(512, 512, scale=512 ** -0.5), rnd(512, scale=0.3), rnd(M, 512)
        buf = guarded(512, torch.float32, r)
        got = gemm_ph_ex(split_planes(a), split_planes(w, triple=True), b, residual=buf[:M], out=buf[:M], a_split=True, out_kind="f32")
        check(got, a.double() @ w.double().t() + b.double() + r.double(), buf)
    elif form == "slot output bf16 operands":
        a, w, r = rnd(M, 1024).bfloat16(), rnd(512, 1024, scale=1024 ** -0.5).bfloat16(), rnd(M, 512)
        buf = guarded(512, torch.float32, r)
        got = gemm_ph_ex(a, w, None, residual=buf[:M], out=buf[:M], out_kind="f32")
        check(got, a.double() @ w.double().t() + r.double(), buf, rtol=1e-5, atol=1e-5)
    elif form == "CTC head":
        a, w, b = rnd(M, 512), rnd(5000, 512, scale=512 ** -0.5), rnd(5000, scale=0.3)
        buf = guarded(5000, torch.float32)
        got = gemm_ph_ex(split_planes(a), split_planes(w, triple=True), b, out=buf[:M], a_split=True, out_kind="f32")
        check(got, a.double() @ w.double().t() + b.double(), buf)
    else:
        Fp, C = 19, 512       # conv2's output pixels as [hi C | lo C] blocks: the A operand of the subsampling Linear (K = 9 728)
        a, w, b = rnd(M, Fp * C, scale=0.5), rnd(512, Fp * C, scale=(Fp * C) ** -0.5), rnd(512, scale=0.3)
        ap = split_planes(a.view(M * Fp, C)).view(M, Fp * 2 * C)
        buf = guarded(512, torch.float32)
        got = gemm_ph_ex(ap, split_planes(w, triple=True), b, out=buf[:M], a_split=True, out_kind="f32", a_plane_block=C)
        check(got, a.double() @ w.double().t() + b.double(), buf, rtol=2e-4, atol=2e-4)


def test_conv_sub_split_at_the_thirty_minute_image(hip):
    """The subsampling convolutions of the headline at the 30-minute file's size: x (1, 179 998, 80) -> conv1 planes (89 998 x 39
    pixels x 1 024 bf16 = the 7.2 GB image) -> conv2 as the split-operand implicit GEMM over 44 998 x 19 = 854 962 output rows
    (3 339 full 256-row tiles + a ragged one; 64-bit per-tile rebased descriptors) -- against float64 torch convolutions of the
    same fp32 operands on WINDOWS of output frames: the first ones, the ones on either side of 2^31 and 2^32 bytes into the image,
    random ones, and the last 300 frames (the ragged tile and the frames before it)."""
    from paper_accurate_fast_cheap_amd.hip_ops import conv_sub_f32split_planes, split_planes
    from tests import parity_log
    C, T, dev = 512, 179998, "cuda"
    g = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(1, T, 80, device=dev, generator=g) * 2.0
    w1 = torch.randn(C, 1, 3, 3, device=dev, generator=g) / 3
    b1 = torch.randn(C, device=dev, generator=g) * 0.1
    w2 = torch.randn(C, C, 3, 3, device=dev, generator=g) / (3 * C ** 0.5)
    b2 = torch.randn(C, device=dev, generator=g) * 0.1
    taps3 = split_planes(w2.permute(2, 3, 0, 1).reshape(9, C, C).contiguous(), triple=True)
    y = conv_sub_f32split_planes(x, w1, b1, taps3, b2)
    T2 = ((T - 3) // 2 + 1 - 3) // 2 + 1
    assert T2 == 44998 and y.shape == (1, T2, 19, 2 * C) and y.dtype == torch.bfloat16
    row_bytes1 = 39 * 2 * C * 2                                     # one conv1 frame of the plane image
    starts = {0, T2 - 300}
    for edge in (2 ** 31, 2 ** 32, 3 * 2 ** 31):                    # output frames whose inputs straddle these image offsets
        starts.add(max(0, edge // row_bytes1 // 2 - 20))
    gen = torch.Generator().manual_seed(5)
    starts |= {int(s) for s in torch.randint(0, T2 - 300, (5,), generator=gen)}
    worst = 0.0
    w1d, b1d, w2d, b2d = w1.double().cpu(), b1.double().cpu(), w2.double().cpu(), b2.double().cpu()   # float64 on the host
    for a in sorted(starts):
        n = 300 if a == T2 - 300 else 64
        xin = x[:, 4 * a:4 * (a + n - 1) + 7].double().unsqueeze(1).cpu()            # frames the window's outputs read
        ref = F.relu(F.conv2d(F.relu(F.conv2d(xin, w1d, b1d, stride=2)), w2d, b2d, stride=2)).permute(0, 2, 3, 1)
        assert ref.shape == (1, n, 19, C)
        got = (y[:, a:a + n, :, :C].double() + y[:, a:a + n, :, C:].double()).cpu()
        d = float((got - ref).abs().max())
        worst = max(worst, d)
        torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4, msg=lambda m, a=a: f"output frames from {a}: {m}")
    parity_log.record("conv2 split at the 30-minute image", windows=len(starts), max_abs_err=worst, output_rows=T2 * 19)
    assert bool(torch.isfinite(y.float()).all())


@pytest.mark.parametrize("B,T,C", [(2, 203, 256), (1, 1203, 512)])
def test_conv_sub_split_operand_planes(hip, B, T, C):
    """The subsampling front end of an fp32 model at long-form sizes: conv1 (fp32 arithmetic) writing planes, conv2 as the
    split-operand implicit GEMM (planes in, planes out), Linear(F' C, odim) reading those planes in blocks of C columns --
    against fp32 torch on the same operands (16-bit significands: ~1e-4, not bf16's 1e-2)."""
    from paper_accurate_fast_cheap_amd.hip_ops import conv_sub_f32split_planes, gemm_ph_ex, split_planes
    x = synth.randn((B, T, 80), 41, 2.0).cuda()
    w1 = (synth.randn((C, 1, 3, 3), 42) / 3).cuda()
    b1 = (synth.randn((C,), 43) * 0.1).cuda()
    w2 = (synth.randn((C, C, 3, 3), 44) / (3 * C ** 0.5)).cuda()
    b2 = (synth.randn((C,), 45) * 0.1).cuda()
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2)), w2, b2, stride=2))      # (B, C, T', F')
    ref = ref.permute(0, 2, 3, 1).contiguous()                                                          # (B, T', F', C)
    taps3 = split_planes(w2.permute(2, 3, 0, 1).reshape(9, C, C).contiguous(), triple=True)
    y = conv_sub_f32split_planes(x, w1, b1, taps3, b2)
    assert y.shape == ref.shape[:-1] + (2 * C,) and y.dtype == torch.bfloat16
    got = y[..., :C].float() + y[..., C:].float()
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)
    Bt, Tp, Fp = ref.shape[0], ref.shape[1], ref.shape[2]
    wl = (synth.randn((256, Fp * C), 46) / (Fp * C) ** 0.5).cuda()
    bl = (synth.randn((256,), 47) * 0.1).cuda()
    out = gemm_ph_ex(y.view(Bt * Tp, Fp * 2 * C), split_planes(wl, triple=True), bl, a_split=True, out_kind="f32", a_plane_block=C)
    want = F.linear(ref.reshape(Bt * Tp, Fp * C).double(), wl.double(), bl.double())
    torch.testing.assert_close(out.double(), want, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("offset", [0.0, 20.0, 100.0])
def test_folded_layernorm_rows_with_large_mean(hip, offset):
    """Every folded LayerNorm forms var = E[x^2] - mean^2 from fp32 partial sums and rstd (acc - mean csum): rows with
    |mean| >> std are where that cancels.  Rows of std ~1 shifted by `offset` (mean / std up to 100, far beyond what a residual
    stream shows) with non-trivial gamma / beta: the folded forms -- gemm_ph LNF consumer after an LNF producer, and
    gemm_skinny ln_self / statistics-in -- against the two-pass LayerNorm + Linear of the SAME stored rows in fp32.  The bf16
    rows themselves carry 2^-9 |offset| of rounding, which the reference's LayerNorm sees identically; what is bounded here
    is the fold's own loss on top of it."""
    from paper_accurate_fast_cheap_amd.hip_ops import gemm_bf16_ln, gemm_skinny
    bf, C, M = torch.bfloat16, 512, 900
    a = synth.randn((M, 1024), 61).to(bf).cuda()
    wo = (synth.randn((C, 1024), 62) / 32).to(bf).cuda()
    r = (synth.randn((M, C), 63) + offset).to(bf).cuda()
    st = torch.empty((M, 8, 2), device="cuda")
    x = gemm_bf16_ln(a, wo, None, st, residual=r)                      # stored rows: mean ~ offset, std ~ 1.4
    xf = x.float()
    g, be = (synth.randn((C,), 64) * 0.3 + 1).to(bf).cuda(), (synth.randn((C,), 65) * 0.3).to(bf).cuda()
    w = (synth.randn((2048, C), 66) / C ** 0.5).to(bf).cuda()
    b = (synth.randn((2048,), 67) * 0.3).to(bf).cuda()
    wp = (w.float() * g.float()).to(bf).contiguous()
    bp = (b.float() + w.float() @ be.float()).to(bf).contiguous()
    cs = wp.float().sum(-1).contiguous()
    ln = F.layer_norm(xf, (C,), None, None, 1e-5)                       # two-pass, fp32, on the stored rows
    want = F.silu(ln @ wp.float().t() + bp.float())                     # the fold's own identity with exact statistics
    got = gemm_bf16_ln(x, wp, bp, st, act="silu", csum=cs, eps=1e-5)
    d = (got.float() - want).abs()
    # cancellation scale: fp32 sums of squares of magnitude C offset^2 lose ~2^-24 C offset^2 of the variance (std^2 ~ 2)
    tol_abs = 3e-2 + 4e-3 * (offset / 20.0) ** 2
    assert float(d.max()) <= 2 ** -6 * float(want.abs().max()) + tol_abs, (offset, float(d.max()))
    assert float(d.mean()) <= 4e-3 + 1e-3 * (offset / 20.0) ** 2, (offset, float(d.mean()))
    # few-rows kernel, statistics from its own operand fragments (ln_self) and from a producer's partials (ln_stats)
    st16 = torch.empty(130, C // 16, 2, dtype=torch.float32, device="cuda")
    xs = gemm_skinny(a[:130].contiguous(), wo, None, residual=r[:130].contiguous(), stats_out=st16)
    want = F.silu(F.layer_norm(xs.float(), (C,), None, None, 1e-5) @ wp.float().t() + bp.float())
    for kw in (dict(ln_self=True), dict(ln_stats=st16)):
        got = gemm_skinny(xs, wp, bp, "silu", ln_csum=cs, ln_eps=1e-5, **kw)
        d = (got.float() - want).abs()
        assert float(d.max()) <= 2 ** -6 * float(want.abs().max()) + tol_abs, (offset, list(kw), float(d.max()))
        assert float(d.mean()) <= 4e-3 + 1e-3 * (offset / 20.0) ** 2, (offset, list(kw), float(d.mean()))


def test_subsampling_f32_long_form_follows_weight_updates(hip, monkeypatch):
    """The fp32 long-form front end keeps derived copies of its weights (NHWC taps, permuted Linear weight and their split
    planes).  Updating the module's weights -- through load_state_dict / copy_ (visible to Tensor._version) AND behind the
    version counter's back the way a fused optimizer does, followed by eval() -- must be followed by the next call; twice in
    a row, so that a freed derived tensor's id / storage coming back cannot resurrect an old entry."""
    from paper_accurate_fast_cheap_amd import hip_ops
    from paper_accurate_fast_cheap_amd.transformer.embedding import RelPositionalEncoding
    from paper_accurate_fast_cheap_amd.transformer.subsampling import Conv2dSubsampling4
    monkeypatch.setattr(hip_ops, "_SPLIT_GEMM_MIN_ROWS", 256)
    C = 256
    torch.manual_seed(3)
    m = Conv2dSubsampling4(80, C, 0.0, RelPositionalEncoding(C, 0.0)).cuda().eval()
    x = synth.randn((1, 4 * 300 + 7, 80), 48, 2.0).cuda()
    mask = torch.ones(1, 1, x.size(1), dtype=torch.bool, device="cuda")
    seen = []
    real = hip_ops.gemm_ph_ex
    monkeypatch.setattr(hip_ops, "gemm_ph_ex", lambda *a, **k: (seen.append(1), real(*a, **k))[1])

    def ref():
        y = F.relu(F.conv2d(F.relu(F.conv2d(x.unsqueeze(1), m.conv[0].weight, m.conv[0].bias, stride=2)), m.conv[2].weight,
                            m.conv[2].bias, stride=2))
        b, c, t, f = y.shape
        return F.linear(y.transpose(1, 2).reshape(b, t, c * f), m.out[0].weight, m.out[0].bias) * C ** 0.5
    with torch.no_grad():
        for update in ("none", "copy_", "data", "copy_", "data"):
            if update == "copy_":
                m.out[0].weight.copy_(torch.randn_like(m.out[0].weight) / 70)
                m.conv[2].weight.copy_(torch.randn_like(m.conv[2].weight) / 48)
            elif update == "data":       # what a fused optimizer does: no version bump ...
                v = m.out[0].weight._version
                m.out[0].weight.data.mul_(-0.5)
                m.conv[2].weight.data.mul_(1.5)
                assert m.out[0].weight._version == v
                m.train()
                m.eval()
                hip_ops.bump_param_epoch()   # ... the encoder's train() / eval() or train_step does this
            got, _, _ = m(x, mask)
            want = ref()
            torch.testing.assert_close(got, want, rtol=3e-4, atol=3e-4 * float(want.abs().max()))
    assert len(seen) == 5                 # the split-operand long-form path served every call


@pytest.mark.parametrize("M", [700, 5000, 44998])
def test_layernorm_folded_into_the_gemms_either_side(hip, M):
    """pafc_gemm_bf16_ph_ln: the residual GEMM writes each row's (sum, sum of squares) in eight 64-column slices; the SiLU / GLU
    projection that follows reads the UN-normalised rows and applies rstd (acc - mean csum) + b' in its epilogue -- against
    LayerNorm + Linear in fp32 torch on the same bf16 operands."""
    from paper_accurate_fast_cheap_amd.hip_ops import add_layernorm, gemm_bf16_ln, glu_interleave
    bf, C = torch.bfloat16, 512
    a = synth.randn((M, 1024), 51).to(bf).cuda()
    wo = (synth.randn((C, 1024), 52) / 32).to(bf).cuda()
    r = (synth.randn((M, C), 53) * 2 + 0.7).to(bf).cuda()
    st = torch.full((M, 8, 2), float("nan"), device="cuda")
    x = gemm_bf16_ln(a, wo, None, st, residual=r)                      # producer: x = r + a wo^T, stats of x
    xf = r.float() + a.float() @ wo.float().t()
    torch.testing.assert_close(x.float(), xf, rtol=2 ** -7, atol=2e-2)
    # the statistics are those of the rows AS STORED (bf16-rounded): per 64-column slice, element for element
    xs_ = x.float().view(M, 8, 64)
    torch.testing.assert_close(st[:, :, 0], xs_.sum(-1), rtol=1e-5, atol=2e-4)
    torch.testing.assert_close(st[:, :, 1], (xs_ * xs_).sum(-1), rtol=1e-5, atol=2e-4)
    s = st.sum(1)                                                       # (M, 2): sum, sum of squares of the rows
    torch.testing.assert_close(s[:, 0], xf.sum(-1), rtol=1e-3, atol=0.5)      # (and close to the unrounded rows' too)
    torch.testing.assert_close(s[:, 1], (xf * xf).sum(-1), rtol=1e-2, atol=0.5)
    # statistics from the LayerNorm pass instead (of its input and of its output): same layout
    g0, b0 = (synth.randn((C,), 54) * 0.2 + 1).to(bf).cuda(), (synth.randn((C,), 55) * 0.2).to(bf).cuda()
    sx, so = torch.empty((M, 8, 2), device="cuda"), torch.empty((M, 8, 2), device="cuda")
    _, o1, _ = add_layernorm(x, None, 1.0, g0, b0, stats_x=sx, stats_out1=so)
    torch.testing.assert_close(sx.sum(1)[:, 0], x.float().sum(-1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(so.sum(1)[:, 1], (o1.float() ** 2).sum(-1), rtol=1e-5, atol=1e-3)
    assert float(sx[:, 1:].abs().max()) == 0.0
    # consumers: LN(x; g, be) w^T + b with SiLU, and with GLU
    g, be = (synth.randn((C,), 56) * 0.3 + 1).to(bf).cuda(), (synth.randn((C,), 57) * 0.3).to(bf).cuda()
    for N, act in ((2048, "silu"), (1024, "glu")):
        w = (synth.randn((N, C), 58) / C ** 0.5).to(bf).cuda()
        b = (synth.randn((N,), 59) * 0.3).to(bf).cuda()
        wp = (w.float() * g.float()).to(bf)
        bp = (b.float() + w.float() @ be.float()).to(bf)
        if act == "glu":
            wp, bp = glu_interleave(wp, 32), glu_interleave(bp, 32)
        got = gemm_bf16_ln(x, wp.contiguous(), bp.contiguous(), st, act=act, csum=wp.float().sum(-1).contiguous(), eps=1e-5)
        lin = F.linear(F.layer_norm(x.float(), (C,), g.float(), be.float(), 1e-5), w.float(), b.float())
        want = F.silu(lin) if act == "silu" else F.glu(lin, dim=-1)
        torch.testing.assert_close(got.float(), want, rtol=2 ** -6, atol=3e-2)
        assert float((got.float() - want).abs().mean()) < 4e-3


@pytest.mark.parametrize("xd,yd", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("scale,p", [(1.0, 0.0), (0.5, 0.0), (1.0, 0.1), (0.5, 0.3)])
def test_residual_dropout_training_kernel(hip, xd, yd, scale, p):
    """x + scale * dropout(y, p) (encoder_layer.py:205-255) as one kernel forward / one backward: p = 0 equals the operator chain
    (one rounding instead of several: within an ulp of the output dtype); p > 0: every element is either x (dropped) or
    x + scale / (1 - p') y (kept) with p' the realised 16-bit rate, the dropped share is p within sampling error, the backward
    pass regenerates the SAME mask, dx is the incoming gradient, and a step is reproducible under torch.manual_seed."""
    from paper_accurate_fast_cheap_amd import hip_ops
    n = (37, 123, 512)
    x = synth.randn(n, 1).to(xd).cuda().requires_grad_()
    y = synth.randn(n, 2).to(yd).cuda().requires_grad_()
    g = synth.randn(n, 3).to(xd).cuda()
    torch.manual_seed(11)
    hip_ops._dropout_calls = 0
    out = hip_ops.residual_dropout(x, y, scale, p, training=True)
    assert out.dtype == xd and out.grad_fn is not None and type(out.grad_fn).__name__.startswith("_ResidualDropout")
    out.backward(g)
    assert torch.equal(x.grad, g) and y.grad.dtype == yd
    thr = 0 if p == 0 else int(p * 65536 + 0.5)
    s = scale / (1 - thr / 65536)
    kept = x.detach().float() + s * y.detach().float()
    eps = 2 ** -7 if xd == torch.bfloat16 else 1e-6
    if p == 0:
        torch.testing.assert_close(out.float(), kept, rtol=eps, atol=eps)
        torch.testing.assert_close(y.grad.float(), s * g.float(), rtol=2 ** -7 if yd == torch.bfloat16 else 1e-6, atol=1e-6)
        # and the same numbers as the operator chain the module would run
        ref = x.detach() + (y.detach() if scale == 1.0 else scale * y.detach())
        torch.testing.assert_close(out.float(), ref.float(), rtol=2 * eps, atol=2 * eps)
        return
    # the mask, read off the backward pass (dy = 0 exactly where dropped; g has no zeros), must be the forward pass's
    gy = y.grad.float()
    drop_mask = (gy == 0) & (g.float() != 0)
    frac = float(drop_mask.float().mean())
    assert abs(frac - p) < 4 * (p * (1 - p) / x.numel()) ** 0.5 + 1e-3, frac
    assert bool((out.float()[drop_mask] == x.detach().float()[drop_mask]).all())
    torch.testing.assert_close(out.float()[~drop_mask], kept[~drop_mask], rtol=eps, atol=eps)
    torch.testing.assert_close(gy[~drop_mask], (s * g.float())[~drop_mask], rtol=2 ** -7 if yd == torch.bfloat16 else 1e-6, atol=1e-6)
    # no structure along rows or columns (a counter-based generator indexed by the element)
    assert abs(float(drop_mask.float().mean(dim=(0, 1)).std()) - (p * (1 - p) / (n[0] * n[1])) ** 0.5) < 0.01
    # reproducible: same seed + same call index -> same mask; another call index -> another mask
    torch.manual_seed(11)
    hip_ops._dropout_calls = 0
    again = hip_ops.residual_dropout(x, y, scale, p, training=True)
    assert torch.equal(again, out)
    other = hip_ops.residual_dropout(x, y, scale, p, training=True)
    assert not torch.equal(other, out)
    # eval: no dropout, framework operators
    ev = hip_ops.residual_dropout(x.detach(), y.detach(), scale, p, training=False)
    torch.testing.assert_close(ev.float(), (x.detach().float() + scale * y.detach().float()), rtol=2 * eps, atol=2 * eps)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_silu_dropout_training_kernel(hip, dt, p):
    """dropout(silu(h), p) of the feed-forward module (positionwise_feed_forward.py:47-55), one kernel each way: forward
    against F.silu (rounded to the activation dtype, then scaled), backward against the analytic derivative on the same mask."""
    from paper_accurate_fast_cheap_amd import hip_ops
    n = (61, 50, 2048)
    h = (synth.randn(n, 5) * 2).to(dt).cuda().requires_grad_()
    g = synth.randn(n, 6).to(dt).cuda()
    torch.manual_seed(12)
    hip_ops._dropout_calls = 0
    out = hip_ops.silu_dropout(h, p, training=True)
    assert out.dtype == dt and type(out.grad_fn).__name__.startswith("_SiluDropout")
    out.backward(g)
    thr = 0 if p == 0 else int(p * 65536 + 0.5)
    s = 1 / (1 - thr / 65536)
    hf = h.detach().float()
    act = F.silu(hf).to(dt).float()
    keep = (out.float() != 0) | (act == 0)
    eps = 2 ** -7 if dt == torch.bfloat16 else 2e-6
    torch.testing.assert_close(out.float()[keep], (s * act)[keep], rtol=eps, atol=1e-6)
    if p == 0:
        assert bool(keep.all())
    else:
        assert abs(float((~keep).float().mean()) - p) < 2e-3
    sg = torch.sigmoid(hf)
    want = s * g.float() * sg * (1 + hf * (1 - sg))
    assert bool((h.grad.float()[~keep] == 0).all())
    torch.testing.assert_close(h.grad.float()[keep], want[keep], rtol=2 * eps, atol=2e-3 if dt == torch.bfloat16 else 1e-5)
