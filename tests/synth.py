"""Deterministic synthetic weights for parity tests.

A golden fixture stores only a *spec* (key -> shape, dtype) and a seed; the
values are regenerated here, identically in the container that captured the
golden (where they were loaded into the reference's modules) and on the GPU
box (where they are loaded into ours).  The generator is torch's CPU mt19937,
same image on both sides; a checksum stored in the fixture guards against drift.

Ranges are chosen so that every branch of the time-mix matters: the reference's
own init leaves time_maa_rkvw_w1 / time_decay_w1 at zero (src/model.py:244,251),
which would switch the data-dependent LoRA paths off.
"""
import math
from typing import Dict, Tuple

import torch

_DT = {"float32": torch.float32, "bfloat16": torch.bfloat16}


def spec_of(state_dict) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    return {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state_dict.items()}


def _one(key: str, shape, gen) -> torch.Tensor:
    leaf = key.split(".")[-1]
    n = lambda s=1.0: torch.randn(shape, generator=gen) * s
    u = lambda a, b: torch.rand(shape, generator=gen) * (b - a) + a
    if leaf in ("time_maa_x", "time_maa_r", "time_maa_k", "time_maa_v", "time_maa_w"):
        return u(0.0, 1.0)
    if leaf == "time_maa_rkvw_w1":
        return n(0.08)
    if leaf == "time_maa_rkvw_w2":
        return u(-0.3, 0.3)
    if leaf == "time_decay":
        return u(-6.0, 0.5)
    if leaf in ("time_decay_w1", "time_decay_w2"):
        return n(0.1)
    if leaf == "time_faaaa":
        return n(0.4)
    if leaf == "mean":
        return n(1.0)
    if leaf == "istd":
        return u(0.5, 1.5)
    if leaf == "weight" and len(shape) == 1:  # LayerNorm gain
        return 1.0 + n(0.1)
    if leaf == "bias":
        return n(0.05)
    if leaf == "weight":
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return n(1.0 / math.sqrt(max(fan_in, 1)))
    return n(0.1)


def synth_state_dict(spec, seed: int) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for key in sorted(spec):
        shape, dt = spec[key]
        out[key] = _one(key, tuple(shape), gen).to(_DT[dt]).contiguous()
    return out


def checksum(sd) -> float:
    return float(sum(v.double().abs().sum() for v in sd.values()))


def randn(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale
