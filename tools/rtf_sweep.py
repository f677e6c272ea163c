#!/usr/bin/env python3
"""The paper's encoder-RTF sweep on one MI355X: chunk size x batch size over one synthetic 30-minute file.

Reference: examples/gigaspeech/s0/local/go-run-encoder-rtf.single-gpu-3x3-g5.sh:59-61 sweeps
chunk_size in {2000, 4000, 9000, 15000, 20000, 40000, 60000, 100000, 200000} x batch_size in {4, 8, 1, 10, 12, 14} through
wenet/bin/encoder-rtf.py (warm-up 3 batches, :41), and tools/rtf/get-rtf-tables.py:9-27 reads `final_rtf`,
`minutes of audio processed per sec` and `max_vram` out of each run into per-model tables with chunk sizes as rows and batch
sizes as columns.  This tool runs the same grid in ONE process, through the package's own window scheduler
(utils.longform.decode_windows: encoder + CTC log-softmax + greedy tokens + stitching inside the timing, `--streams` window
batches in flight, hipGraph replay of the recurring batch shape), in the headline precision (fp32 model + bf16 time-mix slot;
`--dtype bf16` for the whole-model-bf16 mode), and writes

  <out>.jsonl   one record per point: ms per pass, audio-sec/sec, minutes of audio per second, final_rtf, max VRAM (MB),
                the share of the one-sequence rate, the token checksum of the pass and whether it equals the checksum of an
                eager one-stream pass over the same windows (no graph, no side stream)
  <out>.md      the tables, chunk sizes as rows and batch sizes as columns (the layout of get-rtf-tables.py's per-model tables)

The reference's sweep list (go-run-encoder-rtf.single-gpu-3x3-g5.sh:63-103) and how to name each model here:
  rwkv_uni_12L / 18L                 --direction uni [--num-blocks 18]
  rwkv_bi_12L / 18L / 24L / 30L      [--num-blocks N]
  mamba2bi_12L, mamba2_uni_12L       --slot mamba_att --direction bi|uni   (conf/mamba/giga.mamba{bi,}_ds4k31nc_12le.*.yaml: fp32
                                     parameters, no rwkv_do_bfloat16 key -- `--dtype bf16slot` runs them as the YAML ships, fp32)
  rwkvbi_12L_alt-only / -bi11 / -bi9-11 / -BiFirst / -BiLast6
                                     --slot dir_drop_both --dir-dropout-layers "-1" | "11" | "9,10,11" | "0" | "6,7,8,9,10,11"
                                     (conf/rwkv/giga.rwkvbi_dldb_*.yaml; RWKV_BIDIRECTIONAL_LAYERS, read by the wrapper at
                                     construction: rwkv_wrapper_bidirectional_direction_dropout_both.py:25-35).  The script exports
                                     RRWKV_ALT_DECODING=1 -- a name the wrapper does not read (it reads RWKV_ALT_DECODING), so as
                                     RUN the other layers were left-to-right only; `--alt-decoding` gives the intended alternation.
  (mha_* and the LA256-GT / LFXL checkpoints: the MHA baseline is out of scope; LFXL is rwkv_bi_12L's architecture.)

Usage (GPU box):  python tools/rtf_sweep.py [--dtype bf16slot|bf16] [--streams 3] [--passes 3] [--out profiles/r06_rtf_sweep]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

MODEL_NAME = "rwkv_bi_12L-GPU"
CHUNKS = [2000, 4000, 9000, 15000, 20000, 40000, 60000, 100000, 200000]
BATCHES = [1, 4, 8, 10, 12, 14]


def table(records, key, fmt, title):
    chunks = sorted({r["chunk_size"] for r in records})
    batches = sorted({r["batch_size"] for r in records})
    cell = {(r["chunk_size"], r["batch_size"]): r for r in records}
    w = 12
    lines = [f"### {MODEL_NAME} - {title}", "",
             "|" + "Chunk Size".center(w + 2) + "|" + "".join(f"BS {b}".center(w + 2) + "|" for b in batches),
             "|" + "-" * (w + 2) + "|" + "".join("-" * (w + 2) + "|" for _ in batches)]
    for c in chunks:
        row = "|" + str(c).center(w + 2) + "|"
        for b in batches:
            r = cell.get((c, b))
            row += (fmt(r[key]) if r is not None and r.get(key) is not None else "-").center(w + 2) + "|"
        lines.append(row)
    return "\n".join(lines) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16slot", choices=["bf16slot", "bf16"])
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", default=",".join(map(str, CHUNKS)))
    ap.add_argument("--batches", default=",".join(map(str, BATCHES)))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rtf_sweep"))
    ap.add_argument("--no-eager-check", action="store_true", help="skip the eager one-stream pass behind the token checksum")
    ap.add_argument("--num-blocks", type=int, default=12, help="encoder layers: the paper sweeps 12 / 18 / 24 / 30 (go-run-encoder-rtf...sh:63-70)")
    ap.add_argument("--direction", default="bi", choices=["bi", "uni"], help="bidirectional slot (rwkv_tmix60_bidirectional) or uni (rwkv_tmix60)")
    ap.add_argument("--slot", default="rwkv", choices=["rwkv", "mamba_att", "dir_drop", "dir_drop_both"],
                    help="attention slot: RWKV-v6 time-mix (default), Mamba-2 (mamba_att), or the direction-dropout wrappers' eval branches")
    ap.add_argument("--dir-dropout-layers", default=None,
                    help='RWKV_BIDIRECTIONAL_LAYERS for --slot dir_drop*: comma list of the layers that stay bidirectional ("-1": none)')
    ap.add_argument("--alt-decoding", action="store_true", help="RWKV_ALT_DECODING=1 for --slot dir_drop*: the other layers alternate l2r / r2l")
    ap.add_argument("--name", default=None, help="model name in the tables (default: derived from the options)")
    ap.add_argument("--merge-frames", type=int, default=0,
                    help="decode_windows(merge_frames=...): consecutive batches run as one launch of up to this many input frames "
                         "(0 = one forward per batch, the reference's literal schedule)")
    args = ap.parse_args()
    from paper_accurate_fast_cheap_amd import _lib
    from paper_accurate_fast_cheap_amd.utils.longform import decode_windows
    _lib.lib()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    over = dict(num_blocks=args.num_blocks)
    global MODEL_NAME
    MODEL_NAME = f"rwkv_{args.direction}_{args.num_blocks}L-GPU"
    if args.slot == "mamba_att":    # conf/mamba/giga.mamba{bi,}_ds4k31nc_12le.trans.shortform.yaml:21-23
        over.update(selfattention_layer_type="mamba_att", rnn_att_version="mamba2", rnn_att_direction=args.direction)
        MODEL_NAME = f"mamba2{'bi' if args.direction == 'bi' else '_uni'}_{args.num_blocks}L-GPU"
    elif args.slot.startswith("dir_drop"):   # conf/rwkv/giga.rwkvbi_dldb_ds4k31nc_12le.trans.shortform.yaml:21-25
        over.update(selfattention_layer_type="rwkv_tmix60_dir_layer_drop" + ("_both" if args.slot.endswith("both") else ""),
                    rnn_att_direction="bi")
        if args.dir_dropout_layers is not None:      # read by the wrappers when they are constructed
            os.environ["RWKV_BIDIRECTIONAL_LAYERS"] = args.dir_dropout_layers
        os.environ["RWKV_ALT_DECODING"] = "1" if args.alt_decoding else "0"
        MODEL_NAME = f"rwkvbi_{args.num_blocks}L_bi[{args.dir_dropout_layers or 'all'}]{'_alt' if args.alt_decoding else ''}-GPU"
    elif args.direction == "uni":   # conf/rwkv/giga.rwkv_uni_ds4k31nc_12le.*.yaml: same layer, one direction, non-causal conv k = 31
        over.update(selfattention_layer_type="rwkv_tmix60", rnn_att_direction="uni")
    if args.name:
        MODEL_NAME = args.name
    model, _ = bench.build_model(args.dtype, device, **over)
    wave = bench.synthetic_waveform(bench.AUDIO_SECONDS, 777)
    feats, _ = bench.front_end(wave, device)
    if args.dtype == "bf16":
        feats = feats.to(torch.bfloat16)
    frames = feats.shape[1]
    lens = torch.tensor([frames], dtype=torch.int32, device=device)
    precision = bench.PRECISION[args.dtype]
    if args.slot == "mamba_att" and args.dtype == "bf16slot":
        precision = "fp32 model, fp32 Mamba-2 slot (conf/mamba/*.yaml as shipped: fp32 parameters; exact fp32 products, no bf16 slot)"
    enc = model.encoder

    def one_sequence():
        model.ctc_logprobs(model._forward_encoder(feats, lens)[0])
    one_ms = bench.timed_passes(one_sequence, max(args.passes, 3), 2) * 1e3
    one_rate = frames / 100.0 / (one_ms * 1e-3)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    records = []
    with open(args.out + ".jsonl", "w") as fj:
        head = {"one_sequence_ms": round(one_ms, 3), "one_sequence_audio_sec_per_sec": round(one_rate, 1), "dtype": args.dtype,
                "precision": precision, "frames": frames, "streams": args.streams, "merge_frames": args.merge_frames,
                "model": MODEL_NAME,
                "device": torch.cuda.get_device_name(0)}
        fj.write(json.dumps(head) + "\n")
        fj.flush()
        for b in [int(v) for v in args.batches.split(",")]:
            for c in [int(v) for v in args.chunks.split(",")]:
                enc.graph_cache_size = 0
                enc._graphs.clear()
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats(device)
                last = {}

                def step():
                    last["out"] = decode_windows(model, feats, c, b, streams=args.streams, merge_frames=args.merge_frames)
                sec = bench.timed_passes(step, args.passes, args.warmup)
                vram = torch.cuda.max_memory_allocated(device) / 1024 / 1024
                checksum = bench.token_checksum([last["out"]["windows"]])
                rec = {"chunk_size": c, "batch_size": b, "batches_per_pass": -(-frames // (c * b)), "ms_per_pass": round(sec * 1e3, 3),
                       "audio_sec_per_sec": round(frames / 100.0 / sec, 1), "minutes_per_sec": round(frames / 100.0 / 60.0 / sec, 2),
                       "final_rtf": round(sec / (frames / 100.0), 9), "vram": round(vram, 2),
                       "share_of_one_sequence": round(frames / 100.0 / sec / one_rate, 4), "token_checksum": checksum,
                       "tokens": len(last["out"]["tokens"])}
                if not args.no_eager_check:
                    # the same schedule without graph replay and without side streams must decode the same tokens; with merged
                    # launches also: how many WINDOWS decode to the token list of the literal one-forward-per-batch pass (kernel
                    # choice -- GEMM family, scan chunking -- follows the rows per launch, so a near-tie may flip; on a random-init
                    # head most frames are near-ties)
                    enc.graph_cache_size = 0
                    enc._graphs.clear()
                    with torch.no_grad():
                        eager = decode_windows(model, feats, c, b, streams=1, graph_cache=False, merge_frames=args.merge_frames)
                        torch.cuda.synchronize()
                        rec["token_checksum_equals_eager_pass"] = bench.token_checksum([eager["windows"]]) == checksum
                        if args.merge_frames:
                            lit = decode_windows(model, feats, c, b, streams=1, graph_cache=False)
                            torch.cuda.synchronize()
                            same = sum(1 for x, y in zip(lit["windows"], last["out"]["windows"]) if x == y)
                            rec["windows"] = len(lit["windows"])
                            rec["windows_equal_to_literal_schedule"] = same
                            rec["tokens_literal_schedule"] = len(lit["tokens"])
                records.append(rec)
                fj.write(json.dumps(rec) + "\n")
                fj.flush()
                print(f"{time.strftime('%H:%M:%S')} chunk {c:6d} x batch {b:2d}: {rec['ms_per_pass']:8.2f} ms  "
                      f"{rec['audio_sec_per_sec']:9.0f} audio-sec/sec  {rec['share_of_one_sequence']:.2f} of one sequence  "
                      f"tokens == eager: {rec.get('token_checksum_equals_eager_pass')}", flush=True)
    enc.graph_cache_size = 0
    enc._graphs.clear()
    with open(args.out + ".md", "w") as fm:
        fm.write(f"# Encoder RTF sweep, {MODEL_NAME}, one MI355X, {precision}\n\n"
                 f"One synthetic 30-minute file ({frames} frames), `utils.longform.decode_windows` (encoder + CTC log-softmax + greedy "
                 f"tokens + stitching inside the timing), {args.streams} window batches in flight, "
                 + (f"consecutive batches merged into launches of up to {args.merge_frames} frames, " if args.merge_frames else
                    "one forward per batch (the reference's literal schedule), ")
                 + f"{args.warmup} warm-up + {args.passes} "
                 f"timed passes per point.  The same file as ONE sequence: {one_ms:.2f} ms = {one_rate:.0f} audio-sec/sec.  Grid and "
                 f"table layout: go-run-encoder-rtf.single-gpu-3x3-g5.sh:59-61, tools/rtf/get-rtf-tables.py.\n\n")
        fm.write(table(records, "minutes_per_sec", lambda v: f"{v:.2f}", "Minutes of Audio Processed per Second") + "\n")
        fm.write(table(records, "audio_sec_per_sec", lambda v: f"{v:.0f}", "audio-sec/sec (1/RTF)") + "\n")
        fm.write(table(records, "share_of_one_sequence", lambda v: f"{v:.2f}", "share of the one-sequence rate") + "\n")
        fm.write(table(records, "vram", lambda v: f"{v:.2f}", "Max VRAM Usage (MB)") + "\n")
        if not args.no_eager_check:
            fm.write(table(records, "token_checksum_equals_eager_pass", lambda v: "yes" if v else "NO",
                           "token checksum equals the eager one-stream pass") + "\n")
    print("wrote", args.out + ".jsonl", args.out + ".md")


if __name__ == "__main__":
    main()
