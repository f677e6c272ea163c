"""Checkpoint I/O in the reference's format (wenet/utils/checkpoint.py:29-80,114-190): a .pt holding either the
bare state_dict or {'model0': state_dict, 'optimizer0': ...}, plus a sidecar .yaml of run infos."""
import datetime
import os
import re

import torch
import yaml


def load_checkpoint(model: torch.nn.Module, path: str, encoder_only: bool = False, optimizer=None,
                    def_strict: bool = True) -> dict:
    blob = torch.load(path, map_location="cpu", weights_only=False)
    sd = blob["model0"] if isinstance(blob, dict) and "model0" in blob else blob
    if encoder_only:
        sd = {k: v for k, v in sd.items() if "encoder." in k}
    model.load_state_dict(sd, strict=def_strict and not encoder_only)
    if optimizer is not None and isinstance(blob, dict) and "optimizer0" in blob:
        optimizer.load_state_dict(blob["optimizer0"])
    info_path = re.sub(r"\.pt$", ".yaml", path)
    if os.path.exists(info_path):
        with open(info_path, "r") as fin:
            return yaml.load(fin, Loader=yaml.FullLoader) or {}
    return {}


def save_checkpoint(model: torch.nn.Module, path: str, infos=None, optimizer=None) -> None:
    if isinstance(model, torch.nn.parallel.DistributedDataParallel):
        model = model.module
    blob = {"model0": model.state_dict()}
    if optimizer is not None:
        blob["optimizer0"] = optimizer.state_dict()
    torch.save(blob, path)
    infos = dict(infos or {})
    infos["save_time"] = datetime.datetime.now().strftime("%d/%m/%Y %H:%M:%S")
    infos["includes_optimizer"] = optimizer is not None
    with open(re.sub(r"\.pt$", ".yaml", path), "w") as fout:
        fout.write(yaml.dump(infos))
