"""Config c4: the DDP training step -- 32 utterances per GPU, T <= 2000 frames, fp32 master weights with
the bf16 time-mix slot (the YAML default), CTC objective, forward + backward (WKV-6 backward kernels) + gradient
all-reduce (RCCL when launched with torch.distributed.run) + clip + Adam.  Not the headline metric: a measurement of
the training path, printed as one JSON line.

  python tools/bench_train_step.py [--gpus N --steps 5 --warmup 2 --batch 32 --amp bf16 --grad-sync allreduce|rs_ag|bf16|fp16]

--gpus N > 1 starts N ranks itself (a child torchrun, before the parent touches the GPU), as bench.py does; under a
launcher (WORLD_SIZE set) it is one of the ranks.  With N > 1 the line also carries `exposed_allreduce_ms`: the step
timed again under DDP's no_sync() (same kernels, no collective) and subtracted.
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B   # noqa: E402  (model / waveform helpers of the headline bench)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-fused-adam", action="store_true", help="the for-each Adam (default: the fused multi-tensor update)")
    ap.add_argument("--grad-sync", default="allreduce", choices=["allreduce", "rs_ag", "bf16", "fp16"])
    ap.add_argument("--ddp", action="store_true", help="wrap in DistributedDataParallel even with one rank (RCCL init + "
                                                       "bucketing + the all-reduce kernels on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of N > 1 on one GPU with "
                                                           "PAFC_BENCH_ONE_GPU=1)")
    ap.add_argument("--amp", default="none", choices=["none", "bf16"],
                    help="bf16: forward under torch.autocast(bfloat16) (the reference's `dtype: bf16`), fp32 master weights")
    args = ap.parse_args()
    one_gpu = os.environ.get("PAFC_BENCH_ONE_GPU") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = torch.cuda.device_count()
        if have < args.gpus and not one_gpu:
            sys.exit(f"bench_train_step.py: --gpus {args.gpus} but this node shows {have} GPU(s)")
        sys.exit(B.spawn_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world != args.gpus:
        sys.exit(f"bench_train_step.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    use_ddp = world > 1 or args.ddp
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)                  # before the process group: RCCL binds its communicator to this device
    device = torch.device("cuda", local)
    if use_ddp:
        kw = dict(device_id=device) if args.dist_backend == "nccl" else {}
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29641")
            dist.init_process_group(args.dist_backend, rank=0, world_size=1, **kw)
        else:
            dist.init_process_group(args.dist_backend, **kw)
    from paper_accurate_fast_cheap_amd import _lib
    from paper_accurate_fast_cheap_amd.utils.train_utils import train_step, wrap_model_ddp
    _lib.lib()
    model, _ = B.build_model("fp32", device)          # fp32 parameters, bf16 slot: conf/rwkv/*.yaml as shipped
    if use_ddp:
        model = wrap_model_ddp(model, device, grad_sync=args.grad_sync)
    # the reference builds optim.Adam(model.parameters(), **conf) (wenet/utils/train_utils.py:init_optimizer_and_scheduler); the
    # fused implementation is the same update as one multi-tensor kernel instead of ~100 launches
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=not args.no_fused_adam)
    g = torch.Generator().manual_seed(777 + rank)
    lens = torch.randint(100, 2001, (args.batch,), generator=g)
    feats, _ = B.front_end(B.synthetic_waveform(60.0, 777 + rank), device)
    src = feats[0]
    fb = torch.zeros(args.batch, int(lens.max()), 80, device=device)
    for j, n in enumerate(lens.tolist()):
        off = (j * 7919) % (src.shape[0] - 2001)
        fb[j, :n] = src[off:off + n]
    tl = torch.randint(1, 161, (args.batch,), generator=g)
    tl = torch.minimum(tl, ((lens - 1) // 2 - 1) // 2 // 2).clamp(min=1)
    tgt = torch.randint(1, B.VOCAB, (args.batch, int(tl.max())), generator=g)
    batch = {"feats": fb, "feats_lengths": lens.to(device), "target": tgt.to(device), "target_lengths": tl.to(device)}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(); torch.cuda.synchronize()
    info = None
    amp = torch.bfloat16 if args.amp == "bf16" else None
    for i in range(args.warmup):
        info = train_step(model, batch, opt, device, step_index=i, amp_dtype=amp)
    sync(); t0 = time.perf_counter()
    for i in range(args.steps):
        info = train_step(model, batch, opt, device, step_index=i, amp_dtype=amp)
        if os.environ.get("PAFC_BENCH_TRACE_LOSS") == "1" and rank == 0:
            print(f"step {i}: loss {float(info['loss']):.3f} grad_norm {float(info['grad_norm']):.2f}", file=sys.stderr)
    t_issued = time.perf_counter() - t0          # the host has queued every step; what is left is the GPU draining its queue
    sync(); dt = time.perf_counter() - t0
    dt_nosync = None
    if use_ddp:        # the same steps without the gradient exchange: the difference is the all-reduce time not hidden by backward
        with model.no_sync():
            train_step(model, batch, opt, device, step_index=0, amp_dtype=amp)
            sync(); t0 = time.perf_counter()
            for i in range(args.steps):
                train_step(model, batch, opt, device, step_index=i, amp_dtype=amp)
            sync(); dt_nosync = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, dt_nosync], device=device if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); dt, dt_nosync = float(t[0]), float(t[1])
    nparam = sum(p.numel() for p in model.parameters() if p.requires_grad)
    frames = int(lens.sum()) * world * args.steps
    if rank == 0:
        print(json.dumps({"metric": "training audio-sec/sec (c4: fwd + bwd + all-reduce + clip + Adam, CTC objective)",
                          "value": round(frames / 100.0 / dt, 1), "unit": "audio-sec/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                          "host_issue_ms_per_step": round(t_issued / args.steps * 1e3, 2),
                          "utts_per_gpu": args.batch, "frames_per_gpu_step": int(lens.sum()),
                          "loss": float(info["loss"]), "grad_norm": float(info["grad_norm"]), "dtype": "fp32 + bf16 slot" + (", bf16 autocast" if amp else ""),
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                          "ddp": use_ddp, "grad_sync": args.grad_sync if use_ddp else None, "backend": args.dist_backend if use_ddp else None,
                          "gradient_message_MB": round(nparam * 4 / 1e6, 1),
                          "ms_per_step_no_sync": round(dt_nosync / args.steps * 1e3, 2) if dt_nosync else None,
                          "exposed_allreduce_ms": round((dt - dt_nosync) / args.steps * 1e3, 2) if dt_nosync else None}))
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
