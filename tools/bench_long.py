"""Capacity check: files longer than the headline's 30 minutes as ONE sequence through the bidirectional encoder (bf16)."""
import json, sys, time
import torch
import bench as B

dev = torch.device("cuda")
model, _ = B.build_model("bf16", dev)
out = []
for seconds in [float(a) for a in sys.argv[1:]] or [3600.0, 7200.0]:
    feats, _ = B.front_end(B.synthetic_waveform(seconds, 777), dev)
    feats = feats.to(torch.bfloat16)
    lens = torch.tensor([feats.shape[1]], device=dev)
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        enc, _ = model._forward_encoder(feats, lens); logp = model.ctc_logprobs(enc)
        torch.cuda.synchronize(); t0 = time.time(); n = 3
        for _ in range(n):
            enc, _ = model._forward_encoder(feats, lens); logp = model.ctc_logprobs(enc)
        torch.cuda.synchronize(); dt = (time.time() - t0) / n
    out.append({"audio_s": seconds, "frames": int(feats.shape[1]), "encoder_frames": int(enc.shape[1]),
                "ms": round(dt * 1e3, 1), "audio_sec_per_sec": round(seconds / dt, 1),
                "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                "finite": bool(torch.isfinite(logp.float()).all())})
    del feats, enc, logp
    torch.cuda.empty_cache()
print(json.dumps({"workload": "one file as one sequence, bidirectional RWKV-6 encoder + CTC log-softmax, bf16", "runs": out}))
