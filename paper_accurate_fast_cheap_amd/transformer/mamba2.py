"""Mamba-2 in the attention slot (registry key `mamba_att`).

Reference: wenet/transformer/mamba_att_wrapper.py:6-52 and wenet/transformer/mamba2_bidirectional.py:38-145, which
only ADAPT the third-party `mamba_ssm.modules.mamba2.Mamba2` (Rev fork, unpinned: requirements.txt:32, path.sh:10;
+ causal-conv1d >= 1.4.0): constructed as Mamba(dim_att, headdim=head_size) with every other argument at its default
(d_state 128, d_conv 4, expand 2, ngroups 1, rmsnorm, norm_before_gate False, conv_bias True, bias False, chunk 256).
The package is absent from the reference tree and from this image -> *** parity unpinned ***: what follows restates
the PUBLISHED Mamba-2 block (Dao & Gu 2024, "Transformers are SSMs", and mamba_ssm 2.x's module layout, whose
parameter names it keeps so that `self_attn.mamba[.mamba_forward|.mamba_backward].*` checkpoints have a home):

    z, xBC, dt = split(in_proj(u));  xBC = silu(causal_depthwise_conv1d_k4(xBC));  x, B, C = split(xBC)
    dt = softplus(dt + dt_bias);  a_t = exp(dt_t A),  A = -exp(A_log)                       (scalar per head)
    h_t = a_t h_{t-1} + dt_t B_t (x) x_t ;  y_t = C_t . h_t + D x_t                          (state 64 x 128 per head)
    out = out_proj( RMSNorm(y * silu(z)) * norm.weight )

The selective scan runs on the SAME chunked gfx950 scan kernel as RWKV-6: with S_t := a_t h_{t-1} the recurrence is
S_{t+1} = a_{t+1} S_t + (a_{t+1} B_t)(dt_t x_t)^T and y_t = C_t . S_t + (C_t . B_t) dt_t x_t, i.e. WKV-6 with
r = C, k = a_{t+1} B, v = dt x, per-head scalar decay, no bonus (u = 0); d_state 128 is two 64-wide halves."""
import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..rwkv_v6.wkv6_op import wkv6_forward

_EMPTY_CACHE = torch.zeros((0, 0, 0, 0))


class RMSNormGated(nn.Module):
    def __init__(self, d: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d))

    def forward(self, x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        x = (x * F.silu(z)).float()          # norm_before_gate = False: gate first
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps)
        return (x * self.weight.float()).to(z.dtype)


class Mamba2(nn.Module):
    def __init__(self, d_model: int, d_state: int = 128, d_conv: int = 4, expand: int = 2, headdim: int = 64,
                 ngroups: int = 1, A_init_range=(1, 16), dt_min: float = 0.001, dt_max: float = 0.1,
                 dt_init_floor: float = 1e-4, bias: bool = False, conv_bias: bool = True, chunk_size: int = 256):
        super().__init__()
        if headdim != 64 or d_state % 64 or ngroups != 1:
            raise NotImplementedError("scan kernel: headdim 64, d_state a multiple of 64, ngroups 1 (the paper's config)")
        self.d_model, self.d_state, self.d_conv, self.headdim, self.ngroups = d_model, d_state, d_conv, headdim, ngroups
        self.d_inner = expand * d_model
        self.nheads = self.d_inner // headdim
        d_in_proj = 2 * self.d_inner + 2 * ngroups * d_state + self.nheads
        self.in_proj = nn.Linear(d_model, d_in_proj, bias=bias)
        conv_dim = self.d_inner + 2 * ngroups * d_state
        self.conv1d = nn.Conv1d(conv_dim, conv_dim, d_conv, groups=conv_dim, padding=d_conv - 1, bias=conv_bias)
        dt = torch.exp(torch.rand(self.nheads) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))      # inverse softplus
        self.A_log = nn.Parameter(torch.log(torch.empty(self.nheads).uniform_(*A_init_range)))
        self.D = nn.Parameter(torch.ones(self.nheads))
        self.norm = RMSNormGated(self.d_inner, eps=1e-5)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias)
        self.fused_inference = True    # GPU inference: glue kernels (False: the op-by-op restatement below)
        self.ssd_kernel = True         # bf16: the dedicated SSD scan kernel (False: operand planes + the WKV-6 scan)
        import os
        self.scan_bf16_out = os.environ.get("PAFC_MAMBA_SCAN_F32") != "1"   # scan writes bf16(y + D x); else fp32 y + finish kernel

    def _forward_fused(self, u: torch.Tensor, reverse: bool = False) -> torch.Tensor:
        """Inference on the GPU: the same arithmetic with the glue in three kernels (conv1d + SiLU on the xBC slice of
        in_proj's output; the six scan operand planes in one pass; residual terms + gate + RMSNorm in one pass) instead
        of ~40 framework kernels and their (L, 1024) fp32 temporaries."""
        from .. import hip_ops
        lin = lambda t, m: hip_ops.linear_fused(t.contiguous(), m.weight, m.bias, "none")   # hand-written GEMM for bf16
        zxbcdt = lin(u, self.in_proj)                                                  # (B, L, 2 d_inner + 2 N + H)
        di, N, H = self.d_inner, self.d_state, self.nheads
        z = zxbcdt[..., :di]
        dt_raw = zxbcdt[..., 2 * di + 2 * N:]
        xbc = hip_ops.causal_conv_silu_cl(zxbcdt[..., di:2 * di + 2 * N], self.conv1d.weight, self.conv1d.bias, reverse)
        if u.dtype == torch.bfloat16 and self.ssd_kernel:
            # dedicated SSD scan: scalar decay per head and step, B / C shared by the heads -- no operand planes at all
            dt = F.softplus(dt_raw.float() + self.dt_bias.float()).contiguous()           # (B, L, H), tiny
            log_a = (dt * (-torch.exp(self.A_log.float()))).contiguous()
            if self.scan_bf16_out:
                y = hip_ops.mamba2_scan(xbc, dt, log_a, H, reverse, D=self.D.float())       # bf16, skip term inside
                y = hip_ops.mamba2_gate_norm(y, z, self.norm.weight, self.norm.eps)
            else:
                y = hip_ops.mamba2_scan(xbc, dt, log_a, H, reverse)
                y = hip_ops.mamba2_finish(y, None, xbc, dt_raw, z, self.dt_bias.float(), self.D.float(), self.norm.weight,
                                          self.norm.eps, di, diag=False)
            return lin(y, self.out_proj)
        r0, r1, k0, k1, v, w = hip_ops.mamba2_prep(xbc, dt_raw, self.dt_bias.float(), self.A_log.float(), di)
        u0 = torch.zeros(H, 64, dtype=torch.float32, device=u.device)
        y0 = wkv6_forward(r0, k0, v, w, u0, reverse=reverse)
        y1 = wkv6_forward(r1, k1, v, w, u0, reverse=reverse)
        y = hip_ops.mamba2_finish(y0, y1, xbc, dt_raw, z, self.dt_bias.float(), self.D.float(), self.norm.weight,
                                  self.norm.eps, di)
        # (the module call `self.out_proj(y)` = the framework's F.linear stood here until round 6: with two or three window batches
        #  in flight on side streams its fp32 GEMM never finishes -- DESIGN section 4 "the c2 stall"; found by the Mamba-2 sweep)
        return lin(y, self.out_proj)

    def fused_eligible(self, u: torch.Tensor) -> bool:
        return (self.fused_inference and u.is_cuda and not torch.is_grad_enabled() and self.d_state == 128
                and self.d_inner <= 1024 and u.dtype in (torch.float32, torch.bfloat16)
                and self.conv1d.weight.dtype == u.dtype)

    def forward(self, u: torch.Tensor, reverse: bool = False) -> torch.Tensor:
        """reverse: the block run right-to-left on the un-flipped sequence = flip(forward(flip(u))) without the copies
        (only the bf16 kernels take the flag; elsewhere the two flips are made)."""
        if self.fused_eligible(u) and (not reverse or (u.dtype == torch.bfloat16 and self.ssd_kernel)):
            return self._forward_fused(u, reverse)
        if reverse:
            return torch.flip(self.forward(torch.flip(u, [1])), [1])
        Bsz, L, _ = u.shape
        H, P, N = self.nheads, self.headdim, self.d_state
        zxbcdt = self.in_proj(u)
        z, xBC, dt = torch.split(zxbcdt, [self.d_inner, self.d_inner + 2 * N, H], dim=-1)
        if xBC.is_cuda and xBC.dtype in (torch.float32, torch.bfloat16) and xBC.shape[-1] % 2 == 0 and self.d_conv in (3, 4, 7, 15, 31):
            # channels-last depthwise kernels (forward and gradients) instead of the library's grouped convolution, which
            # JIT-compiles a kernel per shape
            from ..hip_ops import depthwise_conv1d_cl_autograd
            xBC = F.silu(depthwise_conv1d_cl_autograd(xBC.contiguous(), self.conv1d.weight, self.conv1d.bias,
                                                      self.d_conv - 1, L))
        else:
            xBC = F.silu(self.conv1d(xBC.transpose(1, 2))[..., :L].transpose(1, 2))    # causal: keep the first L
        x, Bm, Cm = torch.split(xBC, [self.d_inner, N, N], dim=-1)
        dt = F.softplus(dt.float() + self.dt_bias.float())                            # (B, L, H)
        A = -torch.exp(self.A_log.float())                                             # (H,)
        logdec = dt * A                                                                # log a_t  (< 0)
        # shifted decay a_{t+1}; the last step's value never reaches an output
        nxt = torch.cat([logdec[:, 1:], torch.zeros_like(logdec[:, :1])], dim=1)       # (B, L, H)
        a_next = torch.exp(nxt)
        w = torch.log((-nxt).clamp_min(1e-30))                                         # exp(-exp(w)) = a_{t+1}
        xf = x.float().view(Bsz, L, H, P)
        v = (xf * dt.unsqueeze(-1)).reshape(Bsz, L, H * P).contiguous()                # dt_t x_t
        wk = w.unsqueeze(-1).expand(Bsz, L, H, 64).reshape(Bsz, L, H * 64).contiguous()
        u0 = torch.zeros(H, 64, dtype=torch.float32, device=u.device)
        y = torch.zeros(Bsz, L, H * P, dtype=torch.float32, device=u.device)
        for half in range(N // 64):
            Bh = Bm[..., half * 64:(half + 1) * 64].float()                            # (B, L, 64) shared by all heads
            Ch = Cm[..., half * 64:(half + 1) * 64].float()
            k = (a_next.unsqueeze(-1) * Bh.unsqueeze(2)).reshape(Bsz, L, H * 64).contiguous()
            r = Ch.unsqueeze(2).expand(Bsz, L, H, 64).reshape(Bsz, L, H * 64).contiguous()
            y = y + wkv6_forward(r, k, v, wk, u0)
            y = y + ((Bh * Ch).sum(-1, keepdim=True).unsqueeze(-1) * v.view(Bsz, L, H, P)).reshape(Bsz, L, H * P)
        y = y + (xf * self.D.float().view(1, 1, H, 1)).reshape(Bsz, L, H * P)
        y = self.norm(y.to(z.dtype), z)
        return self.out_proj(y)


class Mamba2Bidirectional(nn.Module):
    """mamba2_bidirectional.py:38-145: (Mamba2_f(u) + flip(Mamba2_b(flip(u)))) / 2."""

    def __init__(self, d_model: int, headdim: int = 32, **kw):
        super().__init__()
        self.mamba_forward = Mamba2(d_model, headdim=headdim, **kw)
        self.mamba_backward = Mamba2(d_model, headdim=headdim, **kw)

    def forward(self, u: torch.Tensor) -> torch.Tensor:
        return (self.mamba_forward(u) + self.mamba_backward(u, reverse=True)) / 2


class MambaAttWrapper(nn.Module):
    """mamba_att_wrapper.py:6-52: same MHA-shaped forward as the RWKV wrappers; `cache` handed back untouched."""

    def __init__(self, head_size: int, dim_att: int, num_blocks: int, rnn_att_version: str = "mamba2",
                 rnn_att_direction: str = "bi", layer_id: int = 1):
        super().__init__()
        self.head_size, self.dim_att, self.num_blocks = head_size, dim_att, num_blocks
        self.rnn_att_version, self.rnn_att_direction, self.layer_id = rnn_att_version, rnn_att_direction, layer_id
        if rnn_att_version != "mamba2":
            raise NotImplementedError("only rnn_att_version: mamba2 (the paper's conf/mamba YAMLs)")
        self.mamba = (Mamba2Bidirectional if rnn_att_direction == "bi" else Mamba2)(dim_att, headdim=head_size)

    def forward(self, query: torch.Tensor, key=None, value=None, mask=None, pos_emb=None,
                cache: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.mamba(query), (cache if cache is not None else _EMPTY_CACHE.to(query.device))
