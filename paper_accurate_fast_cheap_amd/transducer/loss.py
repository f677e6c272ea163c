"""RNN-T (transducer) loss -- PARITY UNPINNED.

The reference calls the third-party ``optimized_transducer.transducer_loss`` (Rev fork of csukuangfj/optimized_transducer,
version unpinned: requirements.txt:33, path.sh:9; call site wenet/transducer/transducer.py:506-523) with
``reduction="mean", from_log_softmax=False`` on the joint's concatenated output, falling back to
``torchaudio.functional.rnnt_loss`` (:565-570).  Neither package is in the reference tree or in this image, so the
arithmetic is restated from the published definition (A. Graves, "Sequence Transduction with Recurrent Neural
Networks", 2012, eq. 16-18) with the call site's argument layout:

  logits: (sum_n T_n * (U_n + 1), V), the per-utterance (T_n, U_n + 1, V) lattices flattened and concatenated
          (TransducerJoint.forward_optimized); targets (N, U_max) int; logit_lengths T_n; target_lengths U_n.
  alpha(0, 0) = 0;  alpha(t, u) = logaddexp(alpha(t-1, u) + blank(t-1, u), alpha(t, u-1) + y(t, u-1))
  loss_n = -(alpha(T_n - 1, U_n) + blank(T_n - 1, U_n))
  reduction: "mean" = sum_n loss_n / sum_n T_n ... see `reduction` below.

The lattice is walked along anti-diagonals (t + u = const), so the recursion is T + U vectorised steps of plain
torch ops and autograd provides the gradients; device-agnostic (host logic over framework ops, like the searches).
"""
from typing import List

import torch


def _lattice_nll(logp_blank: torch.Tensor, logp_label: torch.Tensor) -> torch.Tensor:
    """logp_blank (T, U+1), logp_label (T, U) [label u emitted at (t, u)] -> -log P(y | x) (scalar)."""
    T, U1 = logp_blank.shape
    U = U1 - 1
    # "log 0" is a large finite number, not -inf: logaddexp(-inf, -inf) has NaN gradients, and exp(-1e30 - m) is an
    # exact 0 in the forward and the backward pass alike
    neg_inf = torch.tensor(-1e30, dtype=logp_blank.dtype, device=logp_blank.device)
    # alpha on anti-diagonal d holds cells (t, u = d - t); kept as a length-(U+1) vector indexed by u
    alpha = torch.full((U1,), -1e30, dtype=logp_blank.dtype, device=logp_blank.device)
    alpha[0] = 0.0
    u_idx = torch.arange(U1, device=logp_blank.device)
    for d in range(1, T + U):
        t_idx = d - u_idx                                   # time of the cell with label count u on this diagonal
        valid = (t_idx >= 0) & (t_idx < T)
        # from (t-1, u) by blank: needs t-1 >= 0
        tb = (t_idx - 1).clamp(0, T - 1)
        from_blank = torch.where((t_idx - 1 >= 0) & (t_idx - 1 < T), alpha + logp_blank[tb, u_idx], neg_inf)
        # from (t, u-1) by label u-1: needs u >= 1
        tl = t_idx.clamp(0, T - 1)
        if U > 0:
            lab = torch.cat([neg_inf.reshape(1), alpha[:-1] + logp_label[tl[1:], u_idx[:-1]]])
        else:
            lab = neg_inf.reshape(1)
        new = torch.logaddexp(from_blank, lab)
        alpha = torch.where(valid, new, neg_inf)
    return -(alpha[U] + logp_blank[T - 1, U])


def transducer_loss(logits: torch.Tensor, targets: torch.Tensor, logit_lengths: torch.Tensor,
                    target_lengths: torch.Tensor, blank: int, reduction: str = "mean",
                    from_log_softmax: bool = False) -> torch.Tensor:
    """See the module docstring.  ``reduction``: "none" -> (N,) losses; "sum" -> their sum; "mean" -> their sum
    divided by the total number of frames sum_n T_n (optimized_transducer's documented convention for "mean")."""
    blank = int(blank)
    Ts: List[int] = [int(v) for v in logit_lengths.tolist()]
    Us: List[int] = [int(v) for v in target_lengths.tolist()]
    if logits.dim() != 2 or logits.shape[0] != sum(t * (u + 1) for t, u in zip(Ts, Us)):
        raise ValueError("logits must be (sum_n T_n * (U_n + 1), V)")
    if logits.dtype in (torch.float16, torch.bfloat16):
        logits = logits.float()                     # the lattice recursion runs in fp32 (fp64 inputs stay fp64)
    logp = logits if from_log_softmax else logits.log_softmax(-1)
    losses = []
    off = 0
    for n, (T, U) in enumerate(zip(Ts, Us)):
        lat = logp[off:off + T * (U + 1)].view(T, U + 1, -1)
        off += T * (U + 1)
        lb = lat[:, :, blank]
        if U > 0:
            y = targets[n, :U].to(torch.int64)
            ll = lat[:, :U, :].gather(2, y.view(1, U, 1).expand(T, U, 1)).squeeze(2)
        else:
            ll = lat.new_zeros((T, 0))
        losses.append(_lattice_nll(lb, ll))
    loss = torch.stack(losses)
    if reduction == "none":
        return loss
    if reduction == "sum":
        return loss.sum()
    if reduction == "mean":
        return loss.sum() / float(sum(Ts))
    raise ValueError(f"unknown reduction {reduction!r}")
