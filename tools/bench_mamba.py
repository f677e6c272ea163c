"""The Mamba-2 slot (registry key mamba_att, parity unpinned) on the headline workload: one 30-minute file as one sequence
through the 12-layer encoder with bidirectional Mamba-2 in the attention slot, bf16 model.  Prints one JSON line."""
import json, sys, time
import torch
import bench as B
from paper_accurate_fast_cheap_amd.utils.init_model import init_model

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1800.0
dev = torch.device("cuda")
torch.manual_seed(777)
conf = B.encoder_conf()
conf.update(selfattention_layer_type="mamba_att", rnn_att_version="mamba2", rnn_att_direction="bi")
configs = dict(encoder="conformer", encoder_conf=conf, input_dim=80, output_dim=B.VOCAB, ctc="ctc",
               ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})


class A:
    checkpoint = None


model, _ = init_model(A(), configs)
model = model.eval().to(torch.bfloat16).to(dev)
feats, _ = B.front_end(B.synthetic_waveform(seconds, 777), dev)
feats = feats.to(torch.bfloat16)
lens = torch.tensor([feats.shape[1]], device=dev)
with torch.no_grad():
    for _ in range(2):
        enc, _ = model._forward_encoder(feats, lens); logp = model.ctc_logprobs(enc)
    torch.cuda.synchronize(); t0 = time.time(); n = 3
    for _ in range(n):
        enc, _ = model._forward_encoder(feats, lens); logp = model.ctc_logprobs(enc)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(json.dumps({"workload": f"c3 with the Mamba-2 slot (mamba_att, bi), {seconds:.0f} s file as one sequence, bf16 model",
                  "ms_per_file": round(dt * 1e3, 1), "audio_sec_per_sec": round(seconds / dt, 1),
                  "finite": bool(torch.isfinite(logp.float()).all()), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
