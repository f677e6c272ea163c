"""HIP WKV-6 forward vs the CPU oracle, through the C ABI.  GPU only."""
import pytest
import torch

from oracle import wkv6_oracle as WO
from tests import synth

pytestmark = pytest.mark.gpu


def _inputs(B, T, C, H, seed, dtype, wshift=-3.0):
    r, k, v = (synth.randn((B, T, C), seed + i, 0.5) for i in range(3))
    w = synth.randn((B, T, C), seed + 3) + wshift
    u = synth.randn((H, C // H), seed + 4, 0.3)
    return [t.to(dtype).contiguous() for t in (r, k, v, w, u)]


def _tol(dtype):
    # f32: north_star's 1e-3 relative (we see ~1e-5: fast exp + a different but equivalent summation order);
    # bf16: same f32 arithmetic, one rounding at the store -> at most 1 bf16 ulp where the f32 values straddle
    return dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,H,chunk", [
    (1, 1, 64, 1, 0), (2, 7, 128, 2, 0), (2, 37, 128, 2, 8), (3, 100, 192, 3, 16), (1, 257, 512, 8, 64),
    (2, 64, 512, 8, 64), (2, 65, 512, 8, 64), (1, 499, 512, 8, 0), (1, 499, 512, 8, 10 ** 6),
])
@pytest.mark.parametrize("reverse", [False, True])
def test_forward_matches_oracle(hip, dtype, B, T, C, H, chunk, reverse):
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    a = _inputs(B, T, C, H, 1000 + T, dtype)
    ref = WO.forward(*a, reverse=reverse)
    got = wkv6_forward(*[t.cuda() for t in a], reverse=reverse, chunk_len=chunk).cpu()
    torch.testing.assert_close(got.float(), ref.float(), **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("chunk", [0, 16, 10 ** 6])
def test_state_carry(hip, dtype, chunk):
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    B, T, C, H = 2, 150, 128, 2
    a = _inputs(B, T, C, H, 77, dtype)
    g = [t.cuda() for t in a]
    y_ref, s_ref = WO.forward(*a, want_state=True)
    y, s = wkv6_forward(*g, want_state=True, chunk_len=chunk)
    torch.testing.assert_close(y.cpu().float(), y_ref.float(), **_tol(dtype))
    torch.testing.assert_close(s.cpu(), s_ref, rtol=1e-3, atol=1e-4)
    # chunked-with-carry == full sequence (BASELINE.md target c3)
    cut = lambda t, lo, hi: t[:, lo:hi].contiguous()
    y1, s1 = wkv6_forward(*(cut(t, 0, 61) for t in g[:4]), g[4], want_state=True, chunk_len=chunk)
    y2, s2 = wkv6_forward(*(cut(t, 61, T) for t in g[:4]), g[4], s_in=s1, want_state=True, chunk_len=chunk)
    torch.testing.assert_close(torch.cat([y1, y2], 1).float(), y.float(), **_tol(dtype))
    torch.testing.assert_close(s2, s, rtol=1e-3, atol=1e-4)   # state update runs on 16-bit split bf16 MFMA operands


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bidir_one_launch(hip, dtype):
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward_bidir
    B, T, C, H = 2, 203, 512, 8
    f = _inputs(B, T, C, H, 5, dtype)
    b = _inputs(B, T, C, H, 50, dtype)
    yf, yb = wkv6_forward_bidir([t.cuda() for t in f], [t.cuda() for t in b], chunk_len=32)
    torch.testing.assert_close(yf.cpu().float(), WO.forward(*f).float(), **_tol(dtype))
    torch.testing.assert_close(yb.cpu().float(), WO.forward(*b, reverse=True).float(), **_tol(dtype))


@pytest.mark.parametrize("B,T,chunk", [(2, 203, 32), (1, 1000, 0), (3, 77, 10 ** 6)])
def test_decay_bias_inside_the_scan_equals_bias_added_before_it(hip, B, T, chunk):
    """`time_decay + lora` (src/model.py:289, one bf16 rounding) either inside the scan (w_bias: the kernel variants that add
    and round per step, both passes) or beforehand (what the decay-LoRA kernel does when it carries the bias; the scan then runs
    the variants without a bias): the SAME arithmetic, so the outputs must be bit-identical; and both equal the oracle on the
    pre-added input.  T = 203 / 77 / 1000 end in ragged 16-step blocks (the liveness branch of pass C)."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward_bidir
    C, H = 512, 8
    f, b = _inputs(B, T, C, H, 900 + T, torch.bfloat16), _inputs(B, T, C, H, 950 + T, torch.bfloat16)
    wbf, wbb = (synth.randn((C,), 990 + i, 0.7).to(torch.bfloat16) for i in range(2))
    pre = lambda a, wb: a[:3] + [(a[3].float() + wb.float()).to(torch.bfloat16)] + a[4:]
    fg, bg = [t.cuda() for t in f], [t.cuda() for t in b]
    inside = wkv6_forward_bidir(fg, bg, chunk_len=chunk, w_bias=(wbf.cuda(), wbb.cuda()))
    before = wkv6_forward_bidir([t.cuda() for t in pre(f, wbf)], [t.cuda() for t in pre(b, wbb)], chunk_len=chunk)
    for y_in, y_pre in zip(inside, before):
        assert torch.equal(y_in, y_pre)
    torch.testing.assert_close(before[0].cpu().float(), WO.forward(*pre(f, wbf)).float(), **_tol(torch.bfloat16))
    torch.testing.assert_close(before[1].cpu().float(), WO.forward(*pre(b, wbb), reverse=True).float(), **_tol(torch.bfloat16))


def test_strong_decay_does_not_underflow_to_nan(hip):
    """w up to +3 -> d = exp(-20): chunk decay products reach 0 exactly; the scan only multiplies, so
    the result must stay finite and equal the serial answer ('Decay underflow')."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    a = _inputs(1, 300, 128, 2, 9, torch.float32, wshift=1.0)
    ref = WO.forward(*a)
    got = wkv6_forward(*[t.cuda() for t in a], chunk_len=16).cpu()
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4)


def test_long_sequence_property(hip):
    """T' = 44998 (30 min of audio): too long for the O(T) CPU oracle to be instant but fine once; plus the
    size-independent property chunked == serial on the GPU itself."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    a = _inputs(1, 44998, 512, 8, 31, torch.bfloat16)
    g = [t.cuda() for t in a]
    y_chunk = wkv6_forward(*g, chunk_len=0)
    y_serial = wkv6_forward(*g, chunk_len=10 ** 6)
    torch.testing.assert_close(y_chunk.float(), y_serial.float(), rtol=2 ** -7, atol=1e-3)
    ref = WO.forward(*a)
    torch.testing.assert_close(y_chunk.cpu().float(), ref.float(), rtol=2 ** -7, atol=1e-3)


def test_errors_are_reported_not_asserted(hip):
    from paper_accurate_fast_cheap_amd import _lib
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    a = [t.cuda() for t in _inputs(1, 8, 96, 3, 1, torch.float32)]  # N = 32
    with pytest.raises(_lib.PafcError, match="head size"):
        wkv6_forward(*a)
    with pytest.raises(_lib.PafcError, match="no CPU fallback"):
        wkv6_forward(*_inputs(1, 8, 64, 1, 1, torch.float32))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,H,chunk", [(1, 1, 64, 1, 0), (2, 2, 64, 1, 0), (2, 3, 128, 2, 0), (2, 37, 128, 2, 8),
                                            (3, 100, 192, 3, 16), (2, 257, 512, 8, 64), (1, 499, 512, 8, 0),
                                            (1, 499, 512, 8, 10 ** 6), (1, 1500, 128, 2, 0)])
def test_backward_matches_oracle(hip, dtype, B, T, C, H, chunk):
    """gr, gk, gv, gw, gu vs the C restatement of kernel_backward_101/102/103/201."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_backward
    a = _inputs(B, T, C, H, 2000 + T, dtype)
    gy = synth.randn((B, T, C), 2100 + T).to(dtype)
    ref = WO.backward(*a, gy)
    got = wkv6_backward(*[t.cuda() for t in a], gy.cuda(), chunk_len=chunk)
    # gradients are sums over up to T*N products of O(1) terms: compare against the gradient's own scale
    for name, g, rf in zip(("gr", "gk", "gv", "gw", "gu"), got, ref):
        scale = max(1.0, float(rf.float().abs().max()))
        tol = 2e-4 * scale if dtype == torch.float32 else 2 ** -6 * scale
        err = float((g.cpu().float() - rf.float()).abs().max())
        assert err <= tol, f"{name}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("T", [61, 1100])      # 1100: the epilogue's time chunks are longer than its minimum
def test_backward_reverse_direction_and_autograd(hip, T):
    """reverse=True gradients == gradients of the flipped problem; and autograd through the op equals the oracle."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6, wkv6_backward
    B, C, H = 2, 128, 2
    a = _inputs(B, T, C, H, 3000, torch.float32)
    gy = synth.randn((B, T, C), 3001)
    flip = lambda t: t.flip(1).contiguous()
    ref = WO.backward(*(flip(t) for t in a[:4]), a[4], flip(gy))
    got = wkv6_backward(*[t.cuda() for t in a], gy.cuda(), reverse=True, chunk_len=16)
    for name, g, rf in zip(("gr", "gk", "gv", "gw"), got[:4], ref[:4]):
        torch.testing.assert_close(g.cpu(), flip(rf), rtol=1e-3, atol=2e-4 * max(1.0, float(rf.abs().max())), msg=name)
    torch.testing.assert_close(got[4].cpu(), ref[4], rtol=1e-3, atol=2e-4 * float(ref[4].abs().max()))
    leaves = [t.cuda().requires_grad_() for t in a]
    wkv6(*leaves, False).backward(gy.cuda())
    ref_f = WO.backward(*a, gy)
    for leaf, rf in zip(leaves, ref_f):
        torch.testing.assert_close(leaf.grad.cpu(), rf, rtol=1e-3, atol=2e-4 * max(1.0, float(rf.abs().max())))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_matrix_core_and_valu_kernels_agree(hip, dtype, monkeypatch):
    """The two formulations of the chunk kernel (PAFC_WKV6_IMPL=valu / default matrix-core) on the same inputs,
    ragged T (tail block), strong decays included."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    a = _inputs(2, 333, 128, 2, 4242, dtype, wshift=-1.0)
    g = [t.cuda() for t in a]
    y_m, s_m = wkv6_forward(*g, want_state=True, chunk_len=48)
    monkeypatch.setenv("PAFC_WKV6_IMPL", "valu")
    y_v, s_v = wkv6_forward(*g, want_state=True, chunk_len=48)
    ref, s_ref = WO.forward(*a, want_state=True)
    torch.testing.assert_close(y_m.float(), y_v.float(), **_tol(dtype))
    torch.testing.assert_close(y_m.cpu().float(), ref.float(), **_tol(dtype))
    torch.testing.assert_close(s_m.cpu(), s_ref, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(s_v.cpu(), s_ref, rtol=1e-3, atol=1e-4)


def test_lane_ops_selftest(hip):
    """The in-row lane exchanges of the matrix-core kernel (DPP control codes) do what their names say."""
    import ctypes
    out = torch.zeros(2, 64, 4, device="cuda")
    hip.pafc_selftest_lane_ops.restype = ctypes.c_int
    rc = hip.pafc_selftest_lane_ops(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    o, o2 = out[0].cpu(), out[1].cpu()
    for lane in range(64):
        row, t = lane & ~15, lane & 15
        x = lambda l: float(l + 1)
        exp = [x(lane ^ 1), x(lane ^ 2), x(row + (t & 8) + 7 - (t & 7)), x(row + 15 - t)]
        assert o[lane].tolist() == exp, (lane, o[lane].tolist(), exp)
        # the exchanges between 16-lane rows (pass C's quad products): my position in the even / odd row of my row pair,
        # in the lower / upper 32 lanes
        exp2 = [x((lane & ~16)), x(lane | 16), x(lane & ~32), x(lane | 32)]
        assert o2[lane].tolist() == exp2, (lane, o2[lane].tolist(), exp2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C,H,chunk", [(2, 1, 64, 1, 0), (2, 2, 128, 2, 0), (2, 45, 128, 2, 16), (1, 150, 128, 2, 0),
                                            (1, 150, 128, 2, 10 ** 6), (2, 70, 64, 1, 32)])
@pytest.mark.parametrize("reverse", [False, True])
def test_backward_with_initial_state(hip, dtype, B, T, C, H, chunk, reverse):
    """WKV_6STATE (wkv6state_cuda.cu:66-296, built but never run by the reference): gradients of the recurrence started
    from a state, including the state's own, against float64 autograd through the recurrence (one chunk, several chunks,
    ragged chunks, both time directions)."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_backward, wkv6_forward
    a = _inputs(B, T, C, H, 5000 + T, dtype)
    s = synth.randn((B, H, 64, 64), 5100 + T, 0.5)
    gy = synth.randn((B, T, C), 5200 + T).to(dtype)
    flip = (lambda t: t.flip(1)) if reverse else (lambda t: t)
    leaves = [flip(t).double().requires_grad_() for t in a[:4]] + [a[4].double().requires_grad_(), s.double().requires_grad_()]
    y_ref, _ = WO.state_recurrence_f64(*leaves)
    y_ref.backward(flip(gy).double())
    y = wkv6_forward(*[t.cuda() for t in a], s_in=s.cuda(), reverse=reverse)
    torch.testing.assert_close(y.cpu().float(), flip(y_ref.detach()).float(), **_tol(dtype))
    got = wkv6_backward(*[t.cuda() for t in a], gy.cuda(), reverse=reverse, chunk_len=chunk, s_in=s.cuda(), want_gs=True)
    grad = lambda l: l.grad if l.grad is not None else torch.zeros_like(l)     # T == 1: nothing depends on w
    want = [flip(grad(l)) for l in leaves[:4]] + [grad(leaves[4]), grad(leaves[5])]
    for name, g, rf in zip(("gr", "gk", "gv", "gw", "gu", "gs"), got, want):
        scale = max(1.0, float(rf.abs().max()))
        tol = 3e-4 * scale if dtype == torch.float32 else 2 ** -6 * scale
        if T > 1 and name == "gw":           # the last step's gw is stored as zero, as the reference does (:294)
            last = 0 if reverse else T - 1
            assert float(g[:, last].abs().max()) == 0.0
            keep = [t for t in range(T) if t != last]
            g, rf = g[:, keep], rf[:, keep]
        err = float((g.cpu().double() - rf).abs().max())
        assert err <= tol, f"{name}: err {err:.3e} scale {scale:.3e}"


def test_state_autograd_function(hip):
    """wkv6_state under autograd: a state parameter broadcast over the batch receives the batch sum (model.py:100)."""
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_state
    B, T, C, H = 3, 33, 128, 2
    a = _inputs(B, T, C, H, 6000, torch.float32)
    s_param = synth.randn((H, 64, 64), 6001, 0.5).cuda().requires_grad_()
    leaves = [t.cuda().requires_grad_() for t in a]
    gy = synth.randn((B, T, C), 6002).cuda()
    wkv6_state(*leaves, s_param.unsqueeze(0).expand(B, -1, -1, -1), False).backward(gy)
    ref = [t.double().requires_grad_() for t in a] + [s_param.detach().cpu().double().requires_grad_()]
    y_ref, _ = WO.state_recurrence_f64(*ref[:5], ref[5].unsqueeze(0).expand(B, -1, -1, -1))
    y_ref.backward(gy.cpu().double())
    assert s_param.grad.shape == (H, 64, 64)
    torch.testing.assert_close(s_param.grad.cpu().double(), ref[5].grad, rtol=1e-3, atol=3e-4 * float(ref[5].grad.abs().max()))
    torch.testing.assert_close(leaves[3].grad[:, :-1].cpu().double(), ref[3].grad[:, :-1], rtol=1e-3,
                               atol=3e-4 * float(ref[3].grad.abs().max()))


@pytest.mark.parametrize("dtype,suffix", [(torch.bfloat16, ""), (torch.float32, "_fp32")])
def test_torch_ops_wkv6_as_the_reference_calls_them(hip, dtype, suffix):
    """INTEGRATION route C: `torch.ops.wkv6.forward(B, T, C, H, r, k, v, w, u, y)` and `.backward(..., gy, gr, gk, gv, gw,
    gu)` with caller-allocated outputs and gu summed over the batch by the caller -- the call pattern of the reference's
    WKV_6 / WKV_6_FP32 (wenet/rwkv_v6/src/model.py:108-152,161-214) -- against the C restatement of its CUDA kernels."""
    from paper_accurate_fast_cheap_amd.rwkv_v6 import torch_ops
    torch_ops.register()
    fwd_op, bwd_op = getattr(torch.ops.wkv6, "forward" + suffix), getattr(torch.ops.wkv6, "backward" + suffix)

    class WKV(torch.autograd.Function):
        @staticmethod
        def forward(ctx, r, k, v, w, u):
            B, T, C = r.size()
            H = C // 64
            ctx.dims = (B, T, C, H)
            ctx.save_for_backward(r, k, v, w, u)
            y = torch.empty((B, T, C), device=r.device, dtype=dtype, memory_format=torch.contiguous_format)
            fwd_op(B, T, C, H, r, k, v, w, u, y)
            return y

        @staticmethod
        def backward(ctx, gy):
            B, T, C, H = ctx.dims
            r, k, v, w, u = ctx.saved_tensors
            gr, gk, gv, gw = (torch.empty((B, T, C), device=gy.device, dtype=dtype) for _ in range(4))
            gu = torch.empty((B, C), device=gy.device, dtype=dtype)
            bwd_op(B, T, C, H, r, k, v, w, u, gy.contiguous(), gr, gk, gv, gw, gu)
            return gr, gk, gv, gw, torch.sum(gu, 0).view(H, C // H)

    B, T, C, H = 3, 211, 128, 2
    a = _inputs(B, T, C, H, 4242, dtype)
    gy = synth.randn((B, T, C), 4250, 0.5).to(dtype)
    leaves = [t.cuda().requires_grad_() for t in a]
    y = WKV.apply(*leaves)
    y.backward(gy.cuda())
    torch.testing.assert_close(y.detach().cpu().float(), WO.forward(*a).float(), **_tol(dtype))
    ref = WO.backward(*a, gy)
    for name, got, want in zip("gr gk gv gw gu".split(), [t.grad for t in leaves], ref):
        scale = max(float(want.float().abs().max()), 1e-3)
        bar = (2e-3 if dtype == torch.float32 else 3e-2) * scale
        assert float((got.cpu().float() - want.float()).abs().max()) <= bar, name
    with pytest.raises(Exception):      # checked, not asserted: a transposed (non-contiguous) operand is refused
        fwd_op(B, T, C, H, leaves[0].detach().transpose(0, 1).contiguous().transpose(0, 1), *[t.detach() for t in leaves[1:]],
               torch.empty_like(y))


@pytest.mark.parametrize("T", [17, 33, 48, 64])
@pytest.mark.parametrize("reverse", [False, True])
def test_few_blocks_kernel_matches_the_oracle(hip, monkeypatch, T, reverse):
    """wkv6_few_blocks_kernel (opt-in, PAFC_WKV6_FEW=1): the 2-4 blocks of a short bf16 sequence side by side in one launch,
    carried state in, new state out (also in place) -- against the C restatement of wkv6state_cuda.cu:6-65 and against the
    default serial walk of the same inputs."""
    from oracle import wkv6_oracle as WO
    from paper_accurate_fast_cheap_amd.rwkv_v6.wkv6_op import wkv6_forward
    B, C, H = 2, 128, 2
    bf = torch.bfloat16
    r, k, v = (synth.randn((B, T, C), 300 + i, 0.5).to(bf) for i in range(3))
    w = (synth.randn((B, T, C), 303) - 2.0).to(bf)
    u = synth.randn((H, 64), 304, 0.3).to(bf)
    s0 = synth.randn((B, H, 64, 64), 305, 0.5)
    y_ref, s_ref = WO.forward(r, k, v, w, u, s_in=s0, want_state=True, reverse=reverse)
    args = [t.cuda() for t in (r, k, v, w, u)]
    y_serial, s_serial = wkv6_forward(*args, s_in=s0.cuda(), want_state=True, reverse=reverse)
    monkeypatch.setenv("PAFC_WKV6_FEW", "1")
    y, s1 = wkv6_forward(*args, s_in=s0.cuda(), want_state=True, reverse=reverse)
    state = s0.cuda().clone()
    y2, s2 = wkv6_forward(*args, s_in=state, s_out=state, reverse=reverse)        # in place
    assert s2.data_ptr() == state.data_ptr() and torch.equal(y2, y) and torch.equal(s2, s1)
    torch.testing.assert_close(y.float().cpu(), y_ref.float(), rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(s1.cpu(), s_ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(y.float(), y_serial.float(), rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(s1, s_serial, rtol=1e-4, atol=1e-4)
