"""Is the single-window corner of the sweep (2 000 frames x 1) bound by the HOST side of hipGraphLaunch?
Measures, for the fp32 model + bf16 slot at one 2 000-frame window per forward:
  (a) host time of one graph replay (the call returns when the launch is queued) vs its GPU time;
  (b) 90 forwards from ONE host thread round-robin over 3 streams (decode_windows' schedule);
  (c) the same 90 forwards from 3 host threads, one per stream.
  python tools/micro/graph_launch_host_cost.py [chunk] [batch]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from paper_accurate_fast_cheap_amd import _lib
from paper_accurate_fast_cheap_amd.utils.longform import feats_batcher, _side_streams

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_lib.lib()
dev = torch.device("cuda", 0)
model, _ = bench.build_model("bf16slot", dev)
feats, _ = bench.front_end(bench.synthetic_waveform(bench.AUDIO_SECONDS, 777), dev)
batches = [b for b in feats_batcher(feats, chunk, bs, dev)]
full = [b for b in batches if b[0].shape[1] == chunk and b[0].shape[0] == bs]
enc = model.encoder
enc.graph_cache_size = 8
side = _side_streams(dev, 3)
main = torch.cuda.current_stream(dev)


@torch.no_grad()
def fwd(fb, lens):
    e, m = model._forward_encoder(fb, lens)
    return model.ctc_logprobs(e)


for s in side:
    s.wait_stream(main)
for rep in range(3):                         # capture: a shape is replayed from its third sighting on a stream
    for i, (fb, lens) in enumerate(full[:9]):
        with torch.cuda.stream(side[i % 3]):
            fwd(fb, lens)
torch.cuda.synchronize()

# (a) host time of a replay vs GPU time, one stream
fb, lens = full[0]
with torch.cuda.stream(side[0]):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    t0 = time.perf_counter()
    for _ in range(20):
        fwd(fb, lens)
    t_host = (time.perf_counter() - t0) / 20
    b.record()
    torch.cuda.synchronize()
    t_gpu = a.elapsed_time(b) / 20
print(f"(a) one forward of {bs} x {chunk} frames on one stream: host {t_host * 1e3:.3f} ms per call, GPU {t_gpu:.3f} ms per call", flush=True)


def run_single():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (fb, lens) in enumerate(full):
        with torch.cuda.stream(side[i % 3]):
            fwd(fb, lens)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def run_threads():
    torch.cuda.synchronize()

    def worker(k):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side[k]):
            for i in range(k, len(full), 3):
                fwd(*full[i])
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for name, fn in (("(b) one host thread, 3 streams", run_single), ("(c) three host threads, one per stream", run_threads)):
    fn()
    ts = sorted(fn() for _ in range(3))
    sec = len(full) * bs * chunk / 100.0
    print(f"{name}: {len(full)} forwards in {ts[1] * 1e3:.1f} ms = {ts[1] / len(full) * 1e3:.3f} ms per forward = {sec / ts[1]:.0f} audio-sec/sec", flush=True)
