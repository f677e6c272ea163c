"""Config c3/streaming (BASELINE configs[2]): 30-minute file through the uni-directional encoder chunk by chunk with
recurrent-state carry (forward_chunk_carry), vs the same model on the whole sequence.  Prints one JSON line."""
import json, sys, time
import torch
import bench as B
from paper_accurate_fast_cheap_amd.utils.init_model import init_model

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 64           # encoder frames per chunk (64 = 2.56 s)
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1800.0
use_graph = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 1         # concurrent streams per chunk step (serving batch)
dev = torch.device("cuda")
torch.manual_seed(777)
conf = B.encoder_conf()
conf.update(selfattention_layer_type="rwkv_tmix60", rnn_att_direction="uni", causal=True, cnn_module_kernel=15)
configs = dict(encoder="conformer", encoder_conf=conf, input_dim=80, output_dim=B.VOCAB, ctc="ctc",
               ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})


class A:
    checkpoint = None


model, _ = init_model(A(), configs)
model = model.eval().to(torch.bfloat16).to(dev)
enc = model.encoder
feats, _ = B.front_end(B.synthetic_waveform(seconds, 777), dev)
feats = feats.to(torch.bfloat16)
if streams > 1:      # every stream its own audio: the same file rotated by 17 s per stream
    feats = torch.cat([torch.roll(feats, 1700 * i, dims=1) for i in range(streams)], dim=0).contiguous()
T = feats.shape[1]
sub, ctx = enc.embed.subsampling_rate, enc.embed.right_context + 1
stride, window = sub * chunk, (chunk - 1) * sub + ctx
starts = list(range(0, T - ctx + 1, stride))


def run_stream():
    return enc.stream_chunks(feats, chunk, use_graph=use_graph)


with torch.no_grad():
    full, _ = enc(feats[:1], torch.tensor([T], device=dev))
    ys = run_stream()
    torch.cuda.synchronize(); t0 = time.time()
    ys = run_stream()
    torch.cuda.synchronize(); dt = time.time() - t0
    n = min(ys.shape[1], full.shape[1])
    err = (ys[:1, :n].float() - full[:, :n].float()).abs()
print(json.dumps({"workload": f"streaming forward_chunk with state carry, uni-directional 12-layer encoder (causal conv k=15), "
                              f"{seconds:.0f} s of audio, chunk {chunk} encoder frames = {chunk * 0.04:.2f} s, bf16",
                  "chunks": len(starts), "streams": streams, "graph": use_graph, "ms_total": round(dt * 1e3, 1),
                  "ms_per_chunk": round(dt * 1e3 / len(starts), 3), "audio_sec_per_sec": round(streams * seconds / dt, 1),
                  "chunked_vs_full_mean_abs": float(err.mean()), "chunked_vs_full_max_abs": float(err.max())}))
