#!/usr/bin/env python3
"""fp32 projections at a few thousand rows (a c2 decode batch, a batch of 2 000-frame windows) in the headline precision: the
library's exact fp32 GEMM (hip_ops.linear_bias_act, explicit plans) against the split-operand kernel (gemm_ph_ex(a_split), three bf16
products per fp32 product) at every tile height, and what a plan costs to create.  GPU box: python tools/bench_split_mid_rows.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paper_accurate_fast_cheap_amd import hip_ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    shapes = [("w_1 512->2048 silu", 512, 2048, "silu"), ("w_2 2048->512", 2048, 512, "none"), ("pw2 512->512", 512, 512, "none")]
    print(f"{'shape':22s} {'rows':>6s} {'fp32 library us':>16s} " + " ".join(f"{'split tm=' + str(t):>13s}" for t in (64, 128, 192, 256)))
    for name, K, N, act in shapes:
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev) * 0.1
        w3 = hip_ops.split_planes(w, triple=True)
        for rows in (1536, 3992, 8000, 12000, 16000, 24000):
            x = torch.randn(rows, K, device=dev)
            xp = hip_ops.split_planes(x)
            t0 = time.perf_counter()
            hip_ops.linear_bias_act(x, w, b, act)          # first call: creates the plan
            torch.cuda.synchronize()
            t_plan = (time.perf_counter() - t0) * 1e3
            lib = timeit(lambda: hip_ops.linear_bias_act(x, w, b, act))
            cols = []
            for tm in (64, 128, 192, 256):
                try:
                    cols.append(timeit(lambda: hip_ops.gemm_ph_ex(xp, w3, b, act, a_split=True, out_kind="f32" if act == "none" else "planes",
                                                                  tile_m=tm)))
                except Exception as e:      # noqa: BLE001 -- a tile height the kernel refuses is a table entry, not a failure
                    cols.append(float("nan"))
            ref = hip_ops.linear_bias_act(x, w, b, act)
            got = hip_ops.gemm_ph_ex(xp, w3, b, act, a_split=True, out_kind="f32") if act == "none" else None
            err = float((got - ref).abs().max()) if got is not None else float("nan")
            print(f"{name:22s} {rows:6d} {lib:16.1f} " + " ".join(f"{c:13.1f}" for c in cols) + f"   first call {t_plan:7.1f} ms  max|split - lib| {err:.2e}",
                  flush=True)


if __name__ == "__main__":
    main()
