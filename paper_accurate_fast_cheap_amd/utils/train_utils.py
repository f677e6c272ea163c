"""The DDP training step of config c4, with the reference's step semantics (wenet/utils/train_utils.py:354-372,
609-729; wenet/utils/executor.py:185-205): forward -> loss / accum_grad -> backward (DDP's bucketed gradient
all-reduce overlaps here: RCCL over xGMI through torch.distributed backend "nccl") -> clip_grad_norm_ -> skip the
update when the norm is not finite -> optimizer step -> scheduler step.  One process per GPU; the trainer loop,
logging, snapshotting and DeepSpeed engines around it are out of scope."""
import contextlib
from typing import Optional

import torch
import torch.distributed as dist


def _rs_ag_hook(group, bucket):
    """Gradient averaging of one DDP bucket as reduce-scatter + all-gather.  On the 8-GPU xGMI mesh every GPU has a direct
    link to each of the other seven, so each phase moves bucket/8 per link on all seven links at once, where a ring
    all-reduce passes 2*(7/8) of the bucket over one link per hop ("Comm backend")."""
    pg = group if group is not None else dist.group.WORLD
    world = dist.get_world_size(pg)
    flat = bucket.buffer()
    n = flat.numel()
    per = (n + world - 1) // world
    padded = flat if per * world == n else torch.cat([flat, flat.new_zeros(per * world - n)])
    shard = torch.empty(per, dtype=flat.dtype, device=flat.device)

    def gather(_=None):
        shard.div_(world)
        out = torch.empty_like(padded)
        dist.all_gather_into_tensor(out, shard, group=pg)
        flat.copy_(out[:n])
        return flat
    work = dist.reduce_scatter_tensor(shard, padded, op=dist.ReduceOp.SUM, group=pg, async_op=True)
    try:
        return work.get_future().then(gather)      # RCCL: chained on the collective's stream, overlaps with backward
    except RuntimeError:                            # a backend without futures for this collective (gloo): finish it here
        work.wait()
        done = torch.futures.Future()
        done.set_result(gather())
        return done


def wrap_model_ddp(model: torch.nn.Module, device: Optional[torch.device] = None, find_unused_parameters: bool = False,
                   grad_sync: str = "allreduce", bucket_cap_mb: int = 25):
    """train_utils.py:354-372 (torch_ddp engine).  On MI355X the default bucket (25 MB) keeps each all-reduce far
    above the ~MB where the 7-link xGMI mesh saturates; gradient_as_bucket_view avoids one copy of the fp32 grads.

    grad_sync: "allreduce" (DDP's own bucketed all-reduce, RCCL picks the algorithm), "fp16" (the reference's
    `--fp16_grad_sync`: fp16_compress_hook, train_utils.py:368-372), "bf16" (same with bf16: no overflow at gradient
    scale), "rs_ag" (explicit reduce-scatter + all-gather per bucket, see _rs_ag_hook)."""
    if device is not None and device.type == "cuda":
        model = model.to(device)
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index],
                                                        find_unused_parameters=find_unused_parameters,
                                                        gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=find_unused_parameters,
                                                        bucket_cap_mb=bucket_cap_mb)
    if grad_sync in ("fp16", "bf16"):
        from torch.distributed.algorithms.ddp_comm_hooks import default as comm_hooks
        ddp.register_comm_hook(state=None, hook=comm_hooks.fp16_compress_hook if grad_sync == "fp16"
                               else comm_hooks.bf16_compress_hook)
    elif grad_sync == "rs_ag":
        ddp.register_comm_hook(state=None, hook=_rs_ag_hook)
    elif grad_sync != "allreduce":
        raise ValueError(f"grad_sync must be allreduce, fp16, bf16 or rs_ag, not {grad_sync!r}")
    return ddp


def _total_grad_norm(grads):
    """(L2 norm of all gradients as torch.nn.utils.clip_grad_norm_ forms it -- per-tensor norms by torch's own (device, dtype)
    grouping, stacked and summed in that order --, the groups)."""
    from torch.utils._foreach_utils import _group_tensors_by_device_and_dtype
    groups = {key: gs[0] for key, (gs, _) in _group_tensors_by_device_and_dtype([grads]).items()}
    first = grads[0].device
    # one stack PER dtype group, then one cat: torch.stack over the mixed list (fp32 norms of the fp32 gradients, bf16 norms of the
    # time-mix slot's) takes cat's slow path -- one 4-byte device-to-device copy per norm, 226 of the step's launches (round 6);
    # a same-dtype stack is one batched kernel per 128 norms.  Same values in the same order (bf16 -> fp32 is exact).
    parts = [torch.stack([n.to(first) for n in torch._foreach_norm(gs, 2.0)]) for gs in groups.values()]
    if len(parts) == 1:
        flat = parts[0]
    else:
        dt = parts[0].dtype
        for pk in parts[1:]:
            dt = torch.promote_types(dt, pk.dtype)         # what torch.stack over the mixed list promotes to
        flat = torch.cat([pk.to(dt) for pk in parts])
    return torch.linalg.vector_norm(flat, 2.0), groups


def total_grad_norm(params) -> torch.Tensor:
    """The norm clip_grad_norm_ would return, gradients untouched (train_step hands the clip coefficient to a fused optimizer
    as its `grad_scale` instead of rescaling every gradient first)."""
    grads = [p.grad for p in params if p.grad is not None]
    return _total_grad_norm(grads)[0] if grads else torch.zeros(())


def clip_grad_norm_(params, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (L2; the reference's call, wenet/utils/train_utils.py:700) with the
    same arithmetic in far fewer launches when gradient dtypes are mixed: torch multiplies every (device, dtype) group by the
    fp32 clip-coefficient TENSOR, and `_foreach_mul_` then falls to one kernel per tensor for each group that is not fp32 (the
    bf16 time-mix slot: ~400 launches per step).  Here every group is multiplied by the coefficient in the group's own dtype,
    which takes the multi-tensor kernel and is the same product (bit for bit: tests/test_train_step.py).  On the GPU the
    coefficient is never read back (round 6: rounds 3-5 did `float(coef)` to skip the multiplication when nothing is clipped --
    one host synchronisation per step, right between backward and the update: the host, 5 ms ahead of the GPU by then, waited,
    and the GPU then idled while the host queued the optimizer and the next forward); a coefficient of 1 multiplies by 1.
    Returns the total norm (fp32 tensor), like torch."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return torch.zeros(())
    total, groups = _total_grad_norm(grads)
    coef_t = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    on_gpu = grads[0].device.type == "cuda"
    if on_gpu or float(coef_t) < 1.0:        # nan compares false: nothing is scaled, the caller drops the update
        for (dev, dt), gs in groups.items():
            if dt == torch.float32:
                torch._foreach_mul_(gs, coef_t.to(dev))
            elif dev.type == "cuda":
                # on the GPU g.mul_(fp32 0-dim tensor) rounds the coefficient to g's dtype first (measured, torch 2.10): the
                # multi-tensor kernel with the coefficient already in that dtype is the same product
                torch._foreach_mul_(gs, coef_t.to(device=dev, dtype=dt))
            else:
                for g in gs:                       # the CPU kernel keeps the fp32 coefficient: torch's own per-tensor path
                    g.mul_(coef_t)
    return total


def train_step(model: torch.nn.Module, batch: dict, optimizer: torch.optim.Optimizer, device: torch.device,
               grad_clip: float = 0.1, accum_grad: int = 1, scheduler=None, step_index: int = 0,
               amp_dtype: Optional[torch.dtype] = None, scaler=None, clip_hard_maxvalue: float = float("inf"),
               clip_hard_warmup: float = float("inf")) -> dict:
    """One batch: returns {'loss', 'grad_norm', 'updated'}.  grad_clip 0.1 and Adam lr 1e-4 are the YAML's
    (conf/rwkv/*.yaml: grad_clip, optim_conf).

    Mixed precision as in batch_forward / batch_backward / update_parameter_and_lr (train_utils.py:609-729): with a
    `scaler` (the reference's `--use_amp`: fp16 autocast + GradScaler) the loss is scaled for backward, the gradients
    un-scaled before clipping and `scaler.step` does the skip-on-overflow; `amp_dtype=torch.bfloat16` without a scaler
    is the `dtype: bf16` autocast of the reference's other engine -- the natural one on MI355X (bf16 MFMA, no loss
    scaling) and the only one the bf16 time-mix slot accepts (GradScaler cannot un-scale bf16 gradients).  `clip_hard_maxvalue` / `clip_hard_warmup`
    (train_utils.py:683-684,712-716): after the warm-up batches an update whose gradient norm exceeds the hard
    maximum is dropped."""
    model.train()
    if scaler is not None and amp_dtype is None:
        amp_dtype = torch.float16
    from .. import hip_ops
    # bf16 copies of the fp32 master weights, refreshed by one launch instead of a cast per weight (GPU training kernels only)
    shadows = hip_ops.train_shadows() if (device.type == "cuda" and amp_dtype == torch.bfloat16) else contextlib.nullcontext()
    with shadows, torch.autocast(device_type=device.type, dtype=amp_dtype, enabled=amp_dtype is not None):
        out = model(batch, device)
    loss = out["loss"]
    scaled = loss / accum_grad
    (scaler.scale(scaled) if scaler is not None else scaled).backward()
    info = {"loss": loss.detach(), "grad_norm": None, "updated": False}
    if (step_index + 1) % accum_grad == 0:
        params = [p for p in model.parameters() if p.requires_grad]
        if scaler is not None:
            scaler.unscale_(optimizer)
            grad_norm = clip_grad_norm_(params, grad_clip)
            scale_before = scaler.get_scale()
            scaler.step(optimizer)              # skips the update itself when the un-scaled gradients overflowed
            scaler.update()
            info["updated"] = bool(torch.isfinite(grad_norm)) and scaler.get_scale() >= scale_before
        else:
            hard = not (clip_hard_maxvalue == float("inf") or step_index < clip_hard_warmup)
            on_device = (not hard and device.type == "cuda" and getattr(optimizer, "_step_supports_amp_scaling", False)
                         and getattr(optimizer, "found_inf", None) is None and getattr(optimizer, "grad_scale", None) is None)
            # a fused optimizer un-scales by `grad_scale` inside its own launches (the GradScaler protocol: g / grad_scale): the clip
            # g * min(1, max_norm / (norm + 1e-6)) rides there as grad_scale = max(1, (norm + 1e-6) / max_norm) -- no pass over the
            # gradients, no launch (clip_grad_norm_ itself costs ~40 multi-tensor launches per step on two dtype groups)
            grad_norm = total_grad_norm(params) if on_device else clip_grad_norm_(params, grad_clip)
            if on_device and grad_norm.is_cuda:
                # the skip-on-inf / nan decision (train_utils.py:702-711) taken ON THE DEVICE: a fused optimizer reads a
                # `found_inf` flag (the GradScaler protocol) and leaves parameters, moments and step counts alone when it is
                # set -- no host synchronisation between backward and the update, the host keeps queueing the next step.
                # info["updated"] is then a 0-dim bool tensor (bool() of it synchronises, like any tensor).
                finite = torch.isfinite(grad_norm)
                optimizer.found_inf = (~finite).to(torch.float32).reshape(())
                optimizer.grad_scale = torch.clamp((grad_norm + 1e-6) / grad_clip, min=1.0).to(torch.float32).reshape(())
                try:
                    optimizer.step()
                finally:
                    optimizer.found_inf = None
                    optimizer.grad_scale = None
                info["updated"] = finite
            elif torch.isfinite(grad_norm):     # train_utils.py:702-711: skip the update on inf / nan
                if not hard or float(grad_norm) <= clip_hard_maxvalue:
                    optimizer.step()
                    info["updated"] = True
        info["grad_norm"] = grad_norm.detach()
        hip_ops.bump_param_epoch()              # the parameters may have moved (fused optimizers leave Tensor._version alone)
        optimizer.zero_grad(set_to_none=True)
        if scheduler is not None:
            scheduler.step()
    return info


def reduce_seen_frames(num_frames: int, device: torch.device) -> int:
    """executor.py:362: dist.reduce of the frames seen, SUM to rank 0."""
    t = torch.tensor([num_frames], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    return int(t.item())
