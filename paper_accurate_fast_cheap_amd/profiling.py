"""Opt-in per-op device timing with events on the launch stream (used by bench.py for the roofline object).

`with op_timer("wkv6_fwd") as t:` brackets a C-ABI call with two events recorded on torch's current stream --
the stream the kernels are launched on -- when profiling is enabled, and is free otherwise."""
from collections import defaultdict

import torch

_enabled = False
_records = defaultdict(list)
_calls = defaultdict(int)


def enable(flag: bool = True):
    global _enabled
    _enabled = flag
    if flag:
        _records.clear()
        _calls.clear()


def enable_recording(flag: bool):
    """Stop (or resume) recording without dropping what was recorded."""
    global _enabled
    _enabled = flag


class _Null:
    """The disabled timer: one shared object, nothing allocated per call (the op wrappers sit on the launch path of
    launch-bound batches)."""
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _Null()


class _Timed:
    __slots__ = ("name", "meta", "a", "b")

    def __init__(self, name, meta):
        self.name, self.meta = name, meta

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()
        return None

    def __exit__(self, *exc):
        self.b.record()
        _records[self.name].append((self.a, self.b, self.meta))
        return False


def op_timer(name: str, sample: int = 1, **meta):
    """sample = n: time only every n-th call of this name (frequent ops: the events themselves cost ~2 us each)."""
    if not _enabled:
        return _NULL
    _calls[name] += 1
    if sample > 1 and (_calls[name] - 1) % sample:
        return _NULL
    return _Timed(name, meta)


def summary():
    """{name: {"n": launches, "avg_ms": mean device time, "meta": last meta, "total_ms", "flops_total": sum of the timed
    launches' meta["flops"] (None when absent), "records": [(ms, meta), ...]}} -- call after a synchronize.  Ragged workloads
    (every launch another shape) divide the totals; fixed shapes can use avg_ms with the last meta."""
    out = {}
    for name, recs in _records.items():
        ms = [a.elapsed_time(b) for a, b, _ in recs]
        fl = [m.get("flops") for _, _, m in recs]
        out[name] = {"n": len(ms), "avg_ms": sum(ms) / len(ms), "meta": recs[-1][2], "total_ms": sum(ms),
                     "flops_total": sum(fl) if all(f is not None for f in fl) else None,
                     "records": [(t, m) for t, (_, _, m) in zip(ms, recs)]}
    return out
