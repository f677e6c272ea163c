"""Pure-torch reproduction of the round-5 c2 stall (no code of this package on the GPU): the framework's own fp32 GEMM
-- what hip_ops.linear_bias_act fell back to when the library offered no workspace-free kernel for a problem, e.g. w_2 of the
FFN, (rows, 2048) x (512, 2048)^T with beta = 1 -- issued from TWO streams at once.  tools/repro_plan_churn.py showed both
streams of a stalled pass sitting in exactly this product; with one stream the pass finishes in a second.

  python tools/micro/two_stream_linear.py [streams] [iters] [rows] [K] [N]     (watchdog: 40 s, then the Python stacks)
"""
import faulthandler
import sys
import time

import torch
import torch.nn.functional as F

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 27136
K = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
N = int(sys.argv[5]) if len(sys.argv) > 5 else 512
faulthandler.dump_traceback_later(40, exit=True)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
xs = [torch.randn(rows + 320 * i, K, device=dev, generator=g) for i in range(streams)]
ss = [torch.cuda.Stream() for _ in range(streams)] if streams > 1 else [torch.cuda.current_stream()]
torch.cuda.synchronize()
t0 = time.time()
outs = []
with torch.no_grad():
    for s in ss:
        s.wait_stream(torch.cuda.current_stream())
    for it in range(iters):
        for x, s in zip(xs, ss):
            with torch.cuda.stream(s):
                outs.append(F.linear(x, w).abs().mean())
        if it % 50 == 49:
            print(f"{time.strftime('%H:%M:%S')} queued {it + 1} products per stream", flush=True)
    for s in ss:
        torch.cuda.current_stream().wait_stream(s)
    total = float(torch.stack(outs).sum())
print(f"{streams} stream(s), {iters} x F.linear(({rows}+, {K}), ({N}, {K})) fp32: finished in {time.time() - t0:.2f} s, checksum {total:.4f}", flush=True)
