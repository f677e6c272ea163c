"""Driver-visible record of what the GPU parity tests measured (pytest -q drops their prints): per test max / mean |err|,
max |dlogp|, flipped frames and the largest reference margin among them.  Written at session end to
profiles/parity_r06.json and, because only gpurun_out/ travels back from a GPU box, to gpurun_out/parity_r06.json as
well; entries of earlier (partial) runs are kept, same-named ones replaced."""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REC = {}


def record(name: str, **values):
    ent = _REC.setdefault(name, {})
    for k, v in values.items():
        ent[k] = (round(float(v), 9) if isinstance(v, float) or hasattr(v, "item") else v)


def flush():
    if not _REC:
        return
    import torch
    meta = {"written": time.strftime("%Y-%m-%d %H:%M:%S"),
            "device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu"}
    for d in ("profiles", "gpurun_out"):
        path = os.path.join(ROOT, d, "parity_r06.json")
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            try:
                with open(path) as f:
                    doc = json.load(f)
            except (OSError, ValueError):
                doc = {"tests": {}}
            doc["meta"] = meta
            doc["tests"].update(_REC)
            with open(path, "w") as f:
                json.dump(doc, f, indent=1, sort_keys=True)
        except OSError:
            pass
