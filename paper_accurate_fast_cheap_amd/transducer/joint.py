"""RNN-T joint network (reference: wenet/transducer/joint.py:8-94): out = W_out act(W_e enc + W_p pred),
additive join, the paper's config: 512 / 640 -> 640 -> V, tanh (conf/rwkv/*.yaml joint_conf)."""
from typing import Optional

import torch
from torch import nn

_ACT = {"tanh": nn.Tanh, "relu": nn.ReLU, "swish": nn.SiLU, "gelu": nn.GELU, "hardtanh": nn.Hardtanh, "selu": nn.SELU}


class TransducerJoint(nn.Module):
    def __init__(self, vocab_size: int, enc_output_size: int, pred_output_size: int, join_dim: int,
                 prejoin_linear: bool = True, postjoin_linear: bool = False, joint_mode: str = "add",
                 activation: str = "tanh", hat_joint: bool = False, dropout_rate: float = 0.1,
                 hat_activation: str = "tanh"):
        assert joint_mode in ["add"]
        if hat_joint:
            raise NotImplementedError("hat_joint is not used by the paper's configs")
        super().__init__()
        self.activatoin = _ACT[activation]()   # (sic) attribute name as in the reference; it holds no parameters
        self.prejoin_linear = prejoin_linear
        self.postjoin_linear = postjoin_linear
        self.joint_mode = joint_mode
        if not prejoin_linear and not postjoin_linear:
            assert enc_output_size == pred_output_size == join_dim
        self.enc_ffn: Optional[nn.Linear] = nn.Linear(enc_output_size, join_dim) if prejoin_linear else None
        self.pred_ffn: Optional[nn.Linear] = nn.Linear(pred_output_size, join_dim) if prejoin_linear else None
        self.post_ffn: Optional[nn.Linear] = nn.Linear(join_dim, join_dim) if postjoin_linear else None
        self.hat_joint = hat_joint
        self.vocab_size = vocab_size
        self.ffn_out = nn.Linear(join_dim, vocab_size)
        self.join_dim = join_dim

    def forward(self, enc_out: torch.Tensor, pred_out: torch.Tensor, pre_project: bool = True) -> torch.Tensor:
        """enc_out (B, T, E), pred_out (B, U, P) -> (B, T, U, V)."""
        if pre_project and self.prejoin_linear:
            enc_out = self.enc_ffn(enc_out)
            pred_out = self.pred_ffn(pred_out)
        if enc_out.ndim != 4:
            enc_out = enc_out.unsqueeze(2)
        if pred_out.ndim != 4:
            pred_out = pred_out.unsqueeze(1)
        out = enc_out + pred_out
        if self.postjoin_linear:
            out = self.post_ffn(out)
        return self.ffn_out(self.activatoin(out))

    def forward_optimized(self, enc_out: torch.Tensor, pred_out: torch.Tensor, enc_out_len: torch.Tensor,
                          pred_out_len: torch.Tensor, pre_project: bool = True) -> torch.Tensor:
        """joint.py:111-149: the layout the optimized transducer loss wants -- per utterance only the valid
        (T_n, U_n + 1) lattice, flattened and concatenated: (sum_n T_n (U_n + 1), V).  enc_out (B, T, E), pred_out
        (B, U + 1, P) (predictor over the blank-prepended targets)."""
        if not (pre_project and self.prejoin_linear):
            raise NotImplementedError("forward_optimized needs the pre-join projections (as in the reference)")
        rows = []
        for i in range(enc_out.size(0)):
            e = self.enc_ffn(enc_out[i, :int(enc_out_len[i])]).unsqueeze(1)          # (T_n, 1, J)
            d = self.pred_ffn(pred_out[i, :int(pred_out_len[i]) + 1]).unsqueeze(0)   # (1, U_n + 1, J)
            rows.append((e + d).reshape(-1, self.join_dim))
        out = torch.cat(rows)
        if self.postjoin_linear:
            out = self.post_ffn(out)
        return self.ffn_out(self.activatoin(out))
