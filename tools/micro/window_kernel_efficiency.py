"""Per-kernel cost of ONE forward at a window batch's row count against the one-sequence pass, per row: which kernels lose the
most at a few thousand rows.  Runs under rocprofv3:
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/micro/window_kernel_efficiency.py run <chunk> <batch>
  python3 tools/micro/window_kernel_efficiency.py report DIR <chunk> <batch>"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
chunk, bs = int(sys.argv[-2]), int(sys.argv[-1])
if mode == "run":
    import torch
    import bench
    from paper_accurate_fast_cheap_amd import _lib
    from paper_accurate_fast_cheap_amd.utils.longform import feats_batcher
    _lib.lib()
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model("bf16slot", dev)
    feats, _ = bench.front_end(bench.synthetic_waveform(bench.AUDIO_SECONDS, 777), dev)
    lens = torch.tensor([feats.shape[1]], dtype=torch.int32, device=dev)
    fb, fl = next(iter(feats_batcher(feats, chunk, bs, dev)))
    marker = torch.zeros(1, device=dev)
    with torch.no_grad():
        for _ in range(2):
            model.ctc_logprobs(model._forward_encoder(feats, lens)[0])
            model.ctc_logprobs(model._forward_encoder(fb, fl)[0])
        torch.cuda.synchronize()
        marker.cos_(); torch.cuda.synchronize()      # marker kernels delimit the two timed forwards in the trace
        model.ctc_logprobs(model._forward_encoder(feats, lens)[0]); torch.cuda.synchronize()
        marker.cos_(); torch.cuda.synchronize()
        model.ctc_logprobs(model._forward_encoder(fb, fl)[0]); torch.cuda.synchronize()
        marker.cos_(); torch.cuda.synchronize()
else:
    import collections
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cos_kernel" in r["Kernel_Name"]]
    m = marks[-3:]
    def agg(a, b):
        d = collections.defaultdict(lambda: [0, 0])
        for r in rows[a + 1:b]:
            k = r["Kernel_Name"][:96]
            d[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[k][1] += 1
        return d
    one, win = agg(m[0], m[1]), agg(m[1], m[2])
    r_one, r_win = 44998, bs * (((chunk - 1) // 2 - 1) // 2)
    t1, t2 = sum(v[0] for v in one.values()) / 1e3, sum(v[0] for v in win.values()) / 1e3
    print(f"one sequence: {t1:.0f} us for {r_one} rows; window batch {bs} x {chunk}: {t2:.0f} us for {r_win} rows = {t2 / r_win / (t1 / r_one):.2f} x the cost per row")
    print(f"{'window us':>10} {'calls':>5} {'x per row':>9}  kernel (x per row: against the kernels of the one-sequence pass with the same name, where there are any)")
    for k, v in sorted(win.items(), key=lambda kv: -kv[1][0])[:26]:
        ratio = (v[0] / r_win) / (one[k][0] / r_one) if k in one and one[k][0] else float("nan")
        print(f"{v[0] / 1e3:10.1f} {v[1]:5d} {ratio:9.2f}  {k}")
