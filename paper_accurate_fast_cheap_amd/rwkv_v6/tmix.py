"""RWKV-v6 time-mix block on the MI355X.

Host-side mirror of RWKV_Tmix_x060c (reference: wenet/rwkv_v6/src/model.py:218-325): same constructor
arguments, same parameter names / shapes / initialisation (so `encoders.N.self_attn...tmix_block.*`
checkpoints load unchanged), same arithmetic and rounding points.  Differences that matter on gfx950:

  * `reverse=True` evaluates the block on the time-reversed sequence WITHOUT materialising a flip: the
    token shift reads x_{t+1} (zero at t = T-1) and the WKV scan walks t downward.  The reference flips the
    input and the output instead (rwkv_wrapper_bidirectional.py:44,48): two extra (B,T,C) round trips.
  * `mix_project` / `finish` are split so the bidirectional wrapper can run both directions' scans in ONE
    launch (wkv6_forward_bidir) between them.
  * no TorchScript, no import-time compilation, no RWKV_* environment variables (model.py:33-47); head size
    is a constructor argument, not a process-wide global.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..transformer.layer_norm import LayerNorm
from .wkv6_op import wkv6, wkv6_forward

D_MIX_LORA = 32
D_DECAY_LORA = 64


class RWKV_Tmix_x060c(nn.Module):
    def __init__(self, head_size: int, n_layers: int, n_embd: int, dim_att: int, layer_id: int):
        super().__init__()
        self.layer_id = layer_id
        self.head_size = head_size
        self.n_head = dim_att // head_size
        self.dim_att = dim_att
        self.n_embd = n_embd
        self.n_layers = n_layers
        assert dim_att % self.n_head == 0

        with torch.no_grad():
            # closed-form initialisation, model.py:232-260
            r01 = layer_id / max(n_layers - 1, 1)   # 0 -> 1 with depth
            r10 = 1.0 - layer_id / n_layers         # 1 -> ~0 with depth
            ddd = (torch.arange(n_embd, dtype=torch.float32) / n_embd).view(1, 1, n_embd)
            self.time_maa_x = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_r = nn.Parameter(1.0 - torch.pow(ddd, 0.5 * r10))
            self.time_maa_k = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_v = nn.Parameter(1.0 - (torch.pow(ddd, r10) + 0.3 * r01))
            self.time_maa_w = nn.Parameter(1.0 - torch.pow(ddd, r10))
            self.time_maa_rkvw_w1 = nn.Parameter(torch.zeros(n_embd, D_MIX_LORA * 4))
            self.time_maa_rkvw_w2 = nn.Parameter(torch.zeros(4, D_MIX_LORA, n_embd).uniform_(-0.01, 0.01))

            n = torch.arange(dim_att, dtype=torch.float32)
            decay_speed = -6 + 5 * (n / max(dim_att - 1, 1)) ** (0.7 + 1.3 * r01)
            self.time_decay = nn.Parameter(decay_speed.reshape(1, 1, dim_att))
            self.time_decay_w1 = nn.Parameter(torch.zeros(n_embd, D_DECAY_LORA))
            self.time_decay_w2 = nn.Parameter(torch.zeros(D_DECAY_LORA, dim_att).uniform_(-0.01, 0.01))

            zigzag = ((n + 1) % 3 - 1) * 0.1
            faaaa = r01 * (1 - n / max(dim_att - 1, 1)) + zigzag
            self.time_faaaa = nn.Parameter(faaaa.reshape(self.n_head, head_size))

        self.receptance = nn.Linear(n_embd, dim_att, bias=False)
        self.key = nn.Linear(n_embd, dim_att, bias=False)
        self.value = nn.Linear(n_embd, dim_att, bias=False)
        self.output = nn.Linear(dim_att, n_embd, bias=False)
        self.ln_x = LayerNorm(dim_att)

    # ------------------------------------------------------------------------------------------
    def mix_project(self, x: torch.Tensor, reverse: bool = False):
        """model.py:274-289: token shift, data-dependent mixes, r/k/v projections and the decay LoRA.
        Returns contiguous (r, k, v, w) in the block's dtype."""
        B, T, C = x.shape
        mm = self._matmul(x)
        lin = self._linear(x)
        if x.is_cuda and torch.is_grad_enabled():
            from .. import hip_ops
            if hip_ops.tmix_train_eligible(x) and self.time_maa_x.dtype == x.dtype:
                # GPU training step: the two element-wise groups as one kernel each, forward and backward
                xxx = hip_ops.shift_mix_train(x, self.time_maa_x, reverse)
                fold = mm is hip_ops.matmul_param       # tanh (and the decay's bias) in the GEMM epilogue: one launch instead of two (three)
                t = mm(xxx, self.time_maa_rkvw_w1, "tanh") if fold else torch.tanh(mm(xxx, self.time_maa_rkvw_w1))
                maas = (self.time_maa_r, self.time_maa_k, self.time_maa_v, self.time_maa_w)
                if hip_ops.lora_mix4_train_eligible(x, t, self.time_maa_rkvw_w2):
                    zr, zk, zv, zw = hip_ops.lora_mix4_train(x, t, self.time_maa_rkvw_w2, maas, reverse)   # own GEMMs each way
                else:
                    maa4 = torch.stack([m_.reshape(C) for m_ in maas])
                    m = torch.bmm(t.view(B * T, 4, -1).transpose(0, 1), self.time_maa_rkvw_w2).view(4, B, T, C)
                    zr, zk, zv, zw = hip_ops.mix4_train(x, m, maa4, reverse)
                ws = (self.receptance.weight, self.key.weight, self.value.weight)
                if hip_ops.linear_group_train_eligible((zr, zk, zv), ws):
                    r, k, v = hip_ops.linear_group_train((zr, zk, zv), ws)       # one batched launch each way
                else:
                    r, k, v = (lin(z_, w_, None) for z_, w_ in zip((zr, zk, zv), ws))
                if fold:
                    w = mm(mm(zw, self.time_decay_w1, "tanh"), self.time_decay_w2, "none", self.time_decay)
                else:
                    w = self.time_decay + mm(torch.tanh(mm(zw, self.time_decay_w1)), self.time_decay_w2)
                return r.contiguous(), k.contiguous(), v.contiguous(), w.contiguous()
        # model.py:262,274: ZeroPad2d((0,0,1,-1)) = x_{t-1}, zero at t=0; reversed time: x_{t+1}, zero at T-1
        prev = F.pad(x, (0, 0, -1, 1)) if reverse else F.pad(x, (0, 0, 1, -1))
        xx = prev - x
        xxx = x + xx * self.time_maa_x
        xxx = torch.tanh(mm(xxx, self.time_maa_rkvw_w1)).view(B * T, 4, -1).transpose(0, 1)
        xxx = torch.bmm(xxx, self.time_maa_rkvw_w2).view(4, B, T, C)
        mr, mk, mv, mw = xxx.unbind(dim=0)
        r = lin(x + xx * (self.time_maa_r + mr), self.receptance.weight, None)
        k = lin(x + xx * (self.time_maa_k + mk), self.key.weight, None)
        v = lin(x + xx * (self.time_maa_v + mv), self.value.weight, None)
        w = x + xx * (self.time_maa_w + mw)
        w = self.time_decay + mm(torch.tanh(mm(w, self.time_decay_w1)), self.time_decay_w2)
        return r.contiguous(), k.contiguous(), v.contiguous(), w.contiguous()

    def finish(self, y: torch.Tensor) -> torch.Tensor:
        """model.py:323-324: LayerNorm over all C channels (not per head), output projection."""
        return self._linear(y)(self.ln_x(y), self.output.weight, None)

    @staticmethod
    def _matmul(x: torch.Tensor):
        """x @ W for the LoRA parameters; in the GPU training step the variant with the hand-written weight gradient."""
        if x.is_cuda and torch.is_grad_enabled():
            from ..hip_ops import matmul_param
            return matmul_param
        return torch.matmul

    @staticmethod
    def _linear(x: torch.Tensor):
        """F.linear; in the GPU training step the variant whose weight gradient runs on the hand-written kernel."""
        if x.is_cuda and torch.is_grad_enabled():
            from ..hip_ops import linear
            return linear
        return F.linear

    def forward(self, x: torch.Tensor, reverse: bool = False) -> torch.Tensor:
        r, k, v, w = self.mix_project(x, reverse)
        y = wkv6(r, k, v, w, self.time_faaaa, reverse)
        return self.finish(y)

    # state-carrying step for streaming (uni-directional) decoding; see DESIGN.md "state carry"
    def forward_state(self, x: torch.Tensor, shift_in, s_in):
        """x (B, T, C); shift_in (B, 1, C) = last frame of the previous chunk (or None = zeros);
        s_in float32 (B, H, N, N) or None.  Returns (out, shift_out, s_out)."""
        B, T, C = x.shape
        if shift_in is None:
            shift_in = x.new_zeros(B, 1, C)
        prev = torch.cat([shift_in, x[:, :-1]], dim=1)
        xx = prev - x
        xxx = x + xx * self.time_maa_x
        xxx = torch.tanh(xxx @ self.time_maa_rkvw_w1).view(B * T, 4, -1).transpose(0, 1)
        xxx = torch.bmm(xxx, self.time_maa_rkvw_w2).view(4, B, T, C)
        mr, mk, mv, mw = xxx.unbind(dim=0)
        r = self.receptance(x + xx * (self.time_maa_r + mr))
        k = self.key(x + xx * (self.time_maa_k + mk))
        v = self.value(x + xx * (self.time_maa_v + mv))
        w = x + xx * (self.time_maa_w + mw)
        w = self.time_decay + torch.tanh(w @ self.time_decay_w1) @ self.time_decay_w2
        y, s_out = wkv6_forward(r.contiguous(), k.contiguous(), v.contiguous(), w.contiguous(),
                                self.time_faaaa.contiguous(), s_in=s_in, want_state=True)
        return self.finish(y), x[:, -1:].clone(), s_out
