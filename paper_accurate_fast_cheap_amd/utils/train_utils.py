"""The DDP training step of config c4, with the reference's step semantics (wenet/utils/train_utils.py:354-372,
609-729; wenet/utils/executor.py:185-205): forward -> loss / accum_grad -> backward (DDP's bucketed gradient
all-reduce overlaps here: RCCL over xGMI through torch.distributed backend "nccl") -> clip_grad_norm_ -> skip the
update when the norm is not finite -> optimizer step -> scheduler step.  One process per GPU; the trainer loop,
logging, snapshotting and DeepSpeed engines around it are out of scope."""
from typing import Optional

import torch
import torch.distributed as dist


def wrap_model_ddp(model: torch.nn.Module, device: Optional[torch.device] = None, find_unused_parameters: bool = False):
    """train_utils.py:354-372 (torch_ddp engine).  On MI355X the default bucket (25 MB) keeps each all-reduce far
    above the ~MB where the 7-link xGMI mesh saturates; gradient_as_bucket_view avoids one copy of the fp32 grads."""
    if device is not None and device.type == "cuda":
        model = model.to(device)
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index],
                                                         find_unused_parameters=find_unused_parameters,
                                                         gradient_as_bucket_view=True)
    return torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=find_unused_parameters)


def train_step(model: torch.nn.Module, batch: dict, optimizer: torch.optim.Optimizer, device: torch.device,
               grad_clip: float = 0.1, accum_grad: int = 1, scheduler=None, step_index: int = 0) -> dict:
    """One batch: returns {'loss', 'grad_norm', 'updated'}.  grad_clip 0.1 and Adam lr 1e-4 are the YAML's
    (conf/rwkv/*.yaml: grad_clip, optim_conf)."""
    model.train()
    out = model(batch, device)
    loss = out["loss"]
    (loss / accum_grad).backward()
    info = {"loss": loss.detach(), "grad_norm": None, "updated": False}
    if (step_index + 1) % accum_grad == 0:
        params = [p for p in model.parameters() if p.requires_grad]
        grad_norm = torch.nn.utils.clip_grad_norm_(params, grad_clip)
        info["grad_norm"] = grad_norm.detach()
        if torch.isfinite(grad_norm):       # train_utils.py:702-711: skip the update on inf / nan
            optimizer.step()
            info["updated"] = True
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
