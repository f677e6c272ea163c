#!/usr/bin/env python3
"""Headline benchmark: audio-sec/sec (1/RTF) of the long-form encode on MI355X.

Metric (BASELINE.json / SURVEY.md 8(d)): sum(valid input frames)/100 / wall seconds over the timed region
cmvn -> Conv2dSubsampling4 -> 12 bidirectional-RWKV Conformer layers -> after_norm -> CTC log-softmax, i.e. what
wenet/bin/encoder-rtf.py:499-509 times, but with a device synchronize + barrier on both sides (the reference takes
t1 without one, :510).  One "step" = one pass over one synthetic 30-minute file (config c3: 179 998 frames), run
either as ONE sequence (default) or as the paper's window batches (--chunk-size / --batch-size,
encoder-rtf.py:354-385).  Inputs are resident in HBM before the timed region.

N > 1 (torchrun, one rank per GPU): every rank encodes its own file -- independent units, no data-path
collective; value = all ranks' audio seconds / max-over-ranks time => "scaling": "weak".

Prints ONE JSON line on rank 0 (schema in the task contract) with two extra objects:
  roofline     -- the hand-written WKV-6 scan (its three kernels per launch), timed live with events on the launch
                  stream; algorithmic bytes per DESIGN.md: B*T'*C*(4 reads + 1 write)*elem per direction.
  cpu_baseline -- the CPU oracle (reference graph in torch CPU fp32 ops + oracle/wkv6_oracle.c) on a bounded sample
                  of the same features, on this box's host cores (rank 0, N == 1 only).
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AUDIO_SECONDS = 1800.0
FRAMES = 1 + (int(AUDIO_SECONDS * 16000) - 400) // 160   # 179 998 (Kaldi snip_edges framing, 25 ms / 10 ms)
VOCAB = 5000
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
VALU_PEAK_TFLOPS = 157.3
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PFLOP/s headline figure is with 2:1 sparsity)
PRECISION = {"bf16slot": "fp32 model + bf16 time-mix slot (the YAML default, the reference's own precision)",
             "bf16": "whole-model bf16 (a mode the reference's bidirectional wrapper cannot run)"}


_T0 = time.time()


def stage(msg: str) -> None:
    """One line per stage on stderr (rank 0): a long run is never silent, and a stage that hangs is named."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def encoder_conf():
    # examples/gigaspeech/s0/conf/rwkv/giga.rwkvbi_ds4k31nc_12le.trans.shortform.yaml:4-25
    return dict(output_size=512, attention_heads=8, linear_units=2048, num_blocks=12, dropout_rate=0.1,
                positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="conv2d", normalize_before=True,
                cnn_module_kernel=31, use_cnn_module=True, cnn_module_norm="layer_norm", activation_type="swish",
                pos_enc_layer_type="rel_pos", selfattention_layer_type="rwkv_tmix60_bidirectional",
                rnn_att_version="rwkv", rnn_att_direction="bi", rwkv_ctx_len=2048, rwkv_do_bfloat16=True)


def synthetic_waveform(seconds: float, seed: int) -> torch.Tensor:
    """(1, S) float32 in int16 range: seeded band-limited Gaussian noise under a slow amplitude envelope (so that
    the CTC argmax is not constant), SURVEY.md 8(d) "Synthetic inputs"."""
    S = int(seconds * 16000)
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(S + 8, generator=g)
    x = (x[:-8] + 2 * x[2:-6] + 3 * x[4:-4] + 2 * x[6:-2] + x[8:]) / 4.36   # low-pass FIR, unit variance
    t = torch.arange(S, dtype=torch.float32)
    env = 0.15 + 0.85 * (0.5 + 0.5 * torch.sin(2 * math.pi * t / (16000 * 3.1))) * (0.5 + 0.5 * torch.sin(2 * math.pi * t / (16000 * 41.0)))
    return (x * env * 6000.0).round().clamp_(-32768, 32767).unsqueeze(0)


def front_end(wave: torch.Tensor, device):
    """HIP fbank of the whole file; returns ((1, T, 80) fp32 features on `device`, device milliseconds)."""
    from paper_accurate_fast_cheap_amd.dataset.fbank import fbank
    w = wave.to(device)
    fbank(w, num_mel_bins=80)   # warm-up (tables, attributes)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    feats = fbank(w, num_mel_bins=80, frame_length=25.0, frame_shift=10.0, dither=0.0, energy_floor=0.0,
                  sample_frequency=16000.0)
    b.record()
    torch.cuda.synchronize()
    return feats.unsqueeze(0), a.elapsed_time(b)


def windows(feats: torch.Tensor, chunk_size: int, batch_size: int):
    """feats_batcher of encoder-rtf.py:354-385: cut (1, T, 80) into batches of `batch_size` windows of `chunk_size`
    frames; the last window is zero-padded and its length shortened."""
    T = feats.shape[1]
    if chunk_size <= 0:
        yield feats, torch.tensor([T], dtype=torch.int32, device=feats.device)
        return
    per = chunk_size * batch_size
    for b in range(math.ceil(T / per)):
        fb = feats[:, b * per:(b + 1) * per]
        nb = math.ceil(fb.shape[1] / chunk_size)
        lens = torch.full((nb,), chunk_size, dtype=torch.int32)
        pad = nb * chunk_size - fb.shape[1]
        if pad > 0:
            lens[-1] -= pad
            fb = torch.nn.functional.pad(fb, (0, 0, 0, pad))
        yield fb.reshape(nb, chunk_size, 80), lens.to(feats.device)


def streaming_leg(feats32: torch.Tensor, device, seconds: float = 600.0, chunk: int = 64):
    """BASELINE configs[2] beside the headline: the first `seconds` of the same features streamed through the UNI-directional
    12-layer encoder (causal conv, k = 15) in 64-frame chunks (2.56 s) with recurrent-state carry, one stream, the step
    replayed from a hipGraph (encoder.stream_chunks) -- ms per chunk over the second of two passes, graph capture included."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    conf = encoder_conf()
    conf.update(selfattention_layer_type="rwkv_tmix60", rnn_att_direction="uni", causal=True, cnn_module_kernel=15)
    configs = dict(encoder="conformer", encoder_conf=conf, input_dim=80, output_dim=VOCAB, ctc="ctc",
                   ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None

    torch.manual_seed(777)
    model, _ = init_model(A(), configs)
    enc = model.eval().to(torch.bfloat16).to(device).encoder
    x = feats32[:, :int(seconds * 100)].to(device=device, dtype=torch.bfloat16)
    sub, ctx = enc.embed.subsampling_rate, enc.embed.right_context + 1
    nchunks = len(range(0, x.shape[1] - ctx + 1, sub * chunk))
    with torch.no_grad():
        enc.stream_chunks(x, chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = enc.stream_chunks(x, chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the same audio as ONE sequence through the same model: a stream with carried state must reproduce it (the
        # whole-sequence pass itself is pinned against the reference / oracle in tests/test_streaming_gpu.py)
        whole, _ = enc(x, torch.tensor([x.shape[1]], dtype=torch.int32, device=device))
        n = min(whole.shape[1], y.shape[1])
        diff = (y[:, :n].float() - whole[:, :n].float()).abs()
    return {"workload": f"streaming with state carry: uni-directional 12-layer encoder, {x.shape[1] / 100.0:.0f} s of the same audio in "
                        f"{chunk}-frame chunks ({chunk * 0.04:.2f} s), one stream, bf16",
            "chunks": nchunks, "ms_per_chunk": round(dt * 1e3 / nchunks, 3),
            "audio_sec_per_sec": round(x.shape[1] / 100.0 / dt, 1),
            "max_abs_vs_whole_sequence": round(float(diff.max()), 4), "mean_abs_vs_whole_sequence": round(float(diff.mean()), 5),
            "frames_compared": int(n), "finite": bool(torch.isfinite(y.float()).all())}


def streaming_lookahead_leg(feats32: torch.Tensor, device, seconds: float = 600.0, chunk: int = 64):
    """The same stream on the uni-directional model AS SHIPPED (non-causal conv, k = 31: giga.rwkv_uni_ds4k31nc_12le.*.yaml:14-16):
    every layer emits 15 frames behind its input (encoder.forward_chunk_lookahead; 12 x 15 frames = 7.2 s behind the audio,
    exact, nothing recomputed); the steady-state steps run on the fused chunk-step kernels and are replayed from a hipGraph
    (encoder.stream_chunks_lookahead), filling and draining the pipeline takes the module path -- ms per chunk over the
    second of two passes, graph capture included."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    conf = encoder_conf()
    conf.update(selfattention_layer_type="rwkv_tmix60", rnn_att_direction="uni", causal=False, cnn_module_kernel=31)
    configs = dict(encoder="conformer", encoder_conf=conf, input_dim=80, output_dim=VOCAB, ctc="ctc",
                   ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class A:
        checkpoint = None

    torch.manual_seed(777)
    model, _ = init_model(A(), configs)
    enc = model.eval().to(torch.bfloat16).to(device).encoder
    x = feats32[:, :int(seconds * 100)].to(device=device, dtype=torch.bfloat16)
    sub, ctx = enc.embed.subsampling_rate, enc.embed.right_context + 1
    nchunks = len(range(0, x.shape[1] - ctx + 1, sub * chunk))
    with torch.no_grad():
        enc.stream_chunks_lookahead(x, chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = enc.stream_chunks_lookahead(x, chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        whole, _ = enc(x, torch.tensor([x.shape[1]], dtype=torch.int32, device=device))
        diff = (y.float() - whole.float()).abs()
    return {"workload": f"streaming with state carry, the shipped uni YAML (NON-causal conv k = 31, 15 frames of look-ahead per layer): "
                        f"{x.shape[1] / 100.0:.0f} s in {chunk}-frame chunks, one stream, bf16, fused steady-state steps from a hipGraph",
            "algorithmic_latency_frames": 15 * len(enc.encoders), "chunks": nchunks, "ms_per_chunk": round(dt * 1e3 / nchunks, 3),
            "audio_sec_per_sec": round(x.shape[1] / 100.0 / dt, 1), "frames_out": int(y.shape[1]), "frames_whole": int(whole.shape[1]),
            "max_abs_vs_whole_sequence": round(float(diff.max()), 4), "mean_abs_vs_whole_sequence": round(float(diff.mean()), 5)}


def c2_batches(device, dtype, rank: int = 0, world: int = 1):
    """BASELINE configs[1]: 5715 utterances with lengths U[1 s, 20 s] (GigaSpeech DEV size and segment filter),
    cut from one long synthetic signal, sharded by length over the ranks, sorted by length, decode batches of 64
    (local/go-SF-dev-one-model-paper.sh:27), issued longest first.  Returns (batches, fbank ms charged to this shard, the source features on the CPU)."""
    from paper_accurate_fast_cheap_amd.utils.sharding import decode_batches, shard_units
    g = torch.Generator().manual_seed(777)
    lens_all = torch.randint(100, 2001, (5715,), generator=g).tolist()
    plan = decode_batches(shard_units(lens_all, rank, world), lens_all, 64)     # sorted by length, longest batch first
    mine = [i for ids in plan for i in ids]
    wave = synthetic_waveform(600.0, 777 + rank)
    long_feats, fbank_ms = front_end(wave, device)
    fbank_ms *= sum(lens_all[i] for i in mine) / float(long_feats.shape[1])
    src = long_feats[0].to(dtype)
    batches = []
    for ids in plan:
        L = [lens_all[i] for i in ids]
        fb = torch.zeros(len(ids), max(L), 80, dtype=src.dtype, device=device)
        for j, (i, n) in enumerate(zip(ids, L)):
            off = (i * 7919) % (src.shape[0] - 2001)
            fb[j, :n] = src[off:off + n]
        batches.append((fb, torch.tensor(L, dtype=torch.int32, device=device)))
    # (issue order: utils.sharding.decode_batches -- same box: 76 800-77 100 shortest first, 78 100-79 200 longest first)
    return batches, fbank_ms, long_feats.cpu()


def make_step(model, batches, device, nstreams: int = 1, greedy=None, progress=None):
    """One pass over `batches` through the package's own decode loop (utils.longform.greedy_decode_batches): encoder + CTC
    log-softmax (+ greedy tokens when `greedy` is given) per batch, `nstreams` batches in flight on HIP streams of their own.
    Returns (step function, the list the last pass's token lists are left in)."""
    from paper_accurate_fast_cheap_amd.utils.longform import greedy_decode_batches
    last_tokens = []

    def step():
        toks, logp = greedy_decode_batches(model, batches, streams=nstreams, want_tokens=greedy is not None)
        last_tokens[:] = toks or []          # c2 = encoder + CTC log-softmax + greedy tokens (search.py:106-121)
        if progress is not None:
            progress.write(f"{time.strftime('%H:%M:%S')} pass over {len(batches)} batches queued\n")
            progress.flush()
        return logp
    return step, last_tokens


def token_checksum(token_lists) -> str:
    """sha256 prefix over the greedy token lists of a pass (batch by batch, utterance by utterance): two runs that decode the
    same tokens print the same string."""
    import hashlib
    h = hashlib.sha256()
    for batch in token_lists:
        for r in batch:
            toks = getattr(r, "tokens", r)
            h.update((",".join(str(int(t)) for t in toks) + ";").encode())
    return h.hexdigest()[:16]


def timed_passes(step, passes: int, warmup: int = 1):
    with torch.no_grad():
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes


def c2_leg(model, device, dtype: str = "bf16slot"):
    """BASELINE configs[1] beside the headline, on the package's default dispatch: the full synthetic DEV-shaped set (5715
    utterances, decode batch 64), encoder + CTC + greedy tokens, two decode batches in flight, 1 warm-up + 2 timed passes,
    in the precision of `model` (dtype "bf16slot": fp32 features and model, bf16 time-mix slot; "bf16": everything bf16)."""
    from paper_accurate_fast_cheap_amd import profiling
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_greedy_search
    batches, _, _ = c2_batches(device, torch.bfloat16 if dtype == "bf16" else torch.float32)
    frames = int(sum(int(l.sum()) for _, l in batches))
    step, toks = make_step(model, batches, device, nstreams=2, greedy=ctc_greedy_search)
    profiling.enable_recording(False)
    sec = timed_passes(step, 2, 1)
    checksum = token_checksum(toks)
    profiling.enable(True)                    # one more pass with the per-kernel event timers, outside the timed passes
    with torch.no_grad():
        step()
    torch.cuda.synchronize()
    profiling.enable_recording(False)
    prof = profiling.summary()
    leg = {"workload": "c2: 5715 synthetic DEV-shaped utterances (1-20 s), decode batches of 64 sorted by length, encoder + CTC "
                       "log-softmax + greedy tokens, two batches in flight, package-default dispatch, " + PRECISION[dtype],
           "utterances": 5715, "batches": len(batches), "passes": 2, "frames_per_pass": frames,
           "ms_per_pass": round(sec * 1e3, 2), "audio_sec_per_sec": round(frames / 100.0 / sec, 1),
           "token_checksum": checksum, "tokens_total": int(sum(len(getattr(r, "tokens", r)) for b in toks for r in b))}
    rec = prof.get("wkv6_fwd_bidir") or prof.get("wkv6_fwd")
    if rec:     # the scan over ragged decode batches: algorithmic bytes of all its timed launches / their total time
        byts = sum(m["B"] * m["T"] * m["C"] * 5 * m["elem_bytes"] * m["ndir"] for _, m in rec["records"])
        ach = byts / (rec["total_ms"] * 1e-3) / 1e9
        leg["roofline"] = {"kernel": "wkv6 forward scan, both directions, every decode batch of one pass (B x T' from 64 x 24 to 64 x 499)",
                           "bound": "hbm", "launches": rec["n"], "avg_launch_us": round(rec["avg_ms"] * 1e3, 1),
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    leg["mfma"] = {name: {"avg_us": round(r["avg_ms"] * 1e3, 1), "launches": r["n"],
                          "frac": round(r["flops_total"] / (r["total_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
                   for name, r in sorted(prof.items()) if r.get("flops_total")}
    return leg


def windows_leg(model, feats, device, chunk: int = 2000, batch: int = 8, nstreams: int = 3, dtype: str = "bf16slot"):
    """The paper's sweep shape (local/go-run-encoder-rtf.single-gpu-3x3-g5.sh:58-62) beside the headline: the same 30-minute
    file as windows of `chunk` frames in batches of `batch`, encoder + CTC log-softmax + greedy tokens, hipGraph cache for the
    recurring batch shape, three window batches in flight on three streams (one box: 1 / 2 / 3 / 4 in flight = 38 000 / 53 300 /
    59 900 / 50 400 audio-sec/sec), package-default dispatch; 2 warm-up + 3 timed passes."""
    from paper_accurate_fast_cheap_amd.utils.longform import decode_windows
    nb = -(-feats.shape[1] // (chunk * batch))
    old = model.encoder.graph_cache_size
    last = {}

    def step():           # the package's own window scheduler, token lists fetched and stitched inside the timed region
        last["out"] = decode_windows(model, feats, chunk, batch, streams=nstreams)

    def step_merged():    # the same windows, consecutive batches merged into launches of up to the whole file
        last["merged"] = decode_windows(model, feats, chunk, batch, streams=nstreams, merge_frames=feats.shape[1] + chunk)
    try:
        sec = timed_passes(step, 3, 2)
        model.encoder._graphs.clear()
        sec_m = timed_passes(step_merged, 3, 2)
    finally:
        model.encoder.graph_cache_size = old
        model.encoder._graphs.clear()
    wl, wm = last["out"]["windows"], last["merged"]["windows"]
    return {"workload": f"the 30-minute file as windows of {chunk} frames x batch {batch} (encoder-rtf.py:354-385) through "
                        f"utils.longform.decode_windows: encoder + CTC + greedy tokens + stitching, hipGraph replay of the recurring "
                        f"batch shape, {nstreams} batches in flight, package-default dispatch, " + PRECISION[dtype],
            "batches": nb, "passes": 3, "ms_per_pass": round(sec * 1e3, 3),
            "audio_sec_per_sec": round(feats.shape[1] / 100.0 / sec, 1), "token_checksum": token_checksum([wl]),
            # batch_size is the reference's memory knob for a 24 GB card; the windows are independent: decode_windows(merge_frames=...)
            # runs them as one launch (same windows, same order).  Token lists follow the launch shape on a random-init head
            # (scan chunking and GEMM tiles follow the rows per launch): the equal windows are counted, not assumed
            "merged_launches": {"ms_per_pass": round(sec_m * 1e3, 3), "audio_sec_per_sec": round(feats.shape[1] / 100.0 / sec_m, 1),
                                "windows": len(wl), "windows_with_the_same_token_list": sum(1 for a, b in zip(wl, wm) if a == b),
                                "tokens": sum(len(w) for w in wm), "tokens_one_forward_per_batch": sum(len(w) for w in wl)}}


def two_files_leg(model, feats, device, dtype: str = "bf16slot", passes: int = 3):
    """Two 30-minute files in flight on two HIP streams (utils.longform.greedy_decode_batches with one file per "batch": the decode
    loop of recognize_wav2.py:323-351 over a list of files), against the same two files one after the other in the same run.  One
    file's step keeps the matrix cores ~38 % and HBM ~41 % busy: the memory-bound passes of one file can run under the GEMMs of the
    other.  Token lists of each file must be the ones the one-file-at-a-time pass decodes."""
    from paper_accurate_fast_cheap_amd.utils.longform import greedy_decode_batches
    wave_b = synthetic_waveform(AUDIO_SECONDS, 778)
    feats_b, _ = front_end(wave_b, device)
    feats_b = feats_b.to(feats.dtype)
    lens = torch.tensor([FRAMES], dtype=torch.int32, device=device)
    files = [(feats, lens), (feats_b, lens)]
    last = {}

    def run(streams):
        def step():
            toks, _ = greedy_decode_batches(model, files, streams=streams, want_tokens=True)
            last[streams] = toks
        return timed_passes(step, passes, 1)
    sec1 = run(1)
    sec2 = run(2)
    sums1 = [token_checksum([t]) for t in last[1]]
    sums2 = [token_checksum([t]) for t in last[2]]
    audio = 2 * FRAMES / 100.0
    return {"workload": "two 30-minute files per pass (seeds 777 / 778), each ONE sequence, encoder + CTC log-softmax + greedy tokens, "
                        "utils.longform.greedy_decode_batches, " + PRECISION[dtype],
            "passes": passes,
            "one_file_at_a_time": {"ms_per_pass": round(sec1 * 1e3, 3), "audio_sec_per_sec": round(audio / sec1, 1)},
            "two_files_in_flight": {"ms_per_pass": round(sec2 * 1e3, 3), "audio_sec_per_sec": round(audio / sec2, 1)},
            "speedup": round(sec1 / sec2, 4), "token_checksums_per_file": sums2,
            "token_checksums_equal_one_at_a_time": sums1 == sums2}


def build_model(dtype: str, device, **conf_overrides):
    """The bench's model; conf_overrides change encoder_conf keys (tools/rtf_sweep.py: the paper's other models -- num_blocks
    18 / 24 / 30, the uni-directional slot)."""
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    torch.manual_seed(777)  # the trainer's seed, wenet/bin/train.py:71
    configs = dict(encoder="conformer", encoder_conf=dict(encoder_conf(), **conf_overrides), input_dim=80, output_dim=VOCAB, ctc="ctc",
                   ctc_conf={"ctc_blank_id": 0}, model_conf={}, dataset_conf={})

    class Args:
        checkpoint = None

    model, _ = init_model(Args(), configs)
    model.eval()
    if dtype == "bf16":
        model = model.to(torch.bfloat16)   # encoder-rtf.py:424-426 (--bf16)
    return model.to(device), configs


def cpu_baseline(model, feats_cpu_f32, conf, sample_frames: int):
    """Reference CPU path = same graph, torch CPU fp32 ops + the C restatement of the WKV op."""
    from oracle import encoder_oracle as EO
    sd = {k: v.detach().float().cpu() for k, v in model.encoder.state_dict().items()}
    for k in list(sd):  # the slot stores bf16 parameters when rwkv_do_bfloat16 (rwkv_wrapper.py:53-54)
        if ".tmix_block." in k and conf.get("rwkv_do_bfloat16", True):
            sd[k] = sd[k].to(torch.bfloat16)
    csd = {"ctc." + k: v.detach().float().cpu() for k, v in model.ctc.state_dict().items()}
    xs = feats_cpu_f32[:, :sample_frames].contiguous()
    lens = torch.tensor([sample_frames])
    threads = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.time()
        out, _ = EO.encoder_forward(xs, lens, sd, conf, env={})
        EO.ctc_log_softmax(out, csd)
        dt = time.time() - t0
    return {"value": round(sample_frames / 100.0 / dt, 3), "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "precision": "f32 model + bf16 time-mix slot (the reference's YAML default; the GPU headline's precision with "
                         "--dtype bf16slot, the default)",
            "sample": f"first {sample_frames / 100:.0f} s of the same synthetic file, one sequence, fp32 graph with the "
                      f"bf16 time-mix slot, {threads} torch/OpenMP threads, {dt:.1f} s wall"}


def scan_source_sha() -> str:
    """sha256 prefix of the scan kernels' sources: a PMC traffic file is only quoted for the kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("wkv6.hip", "wkv6_mfma.inc"):
        with open(os.path.join(ROOT, "paper_accurate_fast_cheap_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def scan_traffic(meta):
    """HBM bytes per launch of the scan from the rocprofv3 PMC passes committed under profiles/ (tools/summarize_wkv_pmc.py),
    or None when no file was collected on the present kernel sources and shape."""
    import glob
    sha = scan_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_wkv6_bidir_*_hbm_traffic.json")), reverse=True):
        try:
            tj = json.load(open(path))
        except (OSError, ValueError):
            continue
        if tj.get("scan_source_sha") == sha and all(tj["shape"].get(k) == meta[k] for k in ("B", "T", "C", "ndir", "elem_bytes")):
            return tj["hbm_bytes_per_launch_corrected"], os.path.basename(path)
    return None, None


def scan_roofline(prof, copy_gbs):
    """The `roofline` object: the bidirectional WKV-6 scan (its three kernels per launch) from the event timers of the profiled
    steps; algorithmic bytes per DESIGN.md section 4: B*T'*C*(4 reads + 1 write)*elem per direction."""
    rec = prof.get("wkv6_fwd_bidir") or prof.get("wkv6_fwd")
    if not rec:
        return None
    # HBM bytes per launch from the rocprofv3 PMC passes of the same op and shape ON THE PRESENT KERNEL SOURCES, else null
    traffic, traffic_file = scan_traffic(rec["meta"])
    m = rec["meta"]
    alg_bytes = m["B"] * m["T"] * m["C"] * 5 * m["elem_bytes"] * m["ndir"]
    alg_flops = m["B"] * m["T"] * m["C"] * 448 * m["ndir"]
    sec = rec["avg_ms"] * 1e-3
    return {"kernel": "wkv6 forward scan, both directions (chunk_state + state_scan + chunk_output kernels)",
            "bound": "hbm", "achieved": round(alg_bytes / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": traffic_file,
            "launches": rec["n"], "avg_launch_us": round(rec["avg_ms"] * 1e3, 1),
            "algorithmic_bytes_per_launch": alg_bytes,
            "measured_copy_gbs": copy_gbs,
            "frac_of_measured_copy": round(alg_bytes / sec / 1e9 / copy_gbs, 4) if copy_gbs else None,
            "valu_tflops": round(alg_flops / sec / 1e12, 2),
            "valu_frac": round(alg_flops / sec / 1e12 / VALU_PEAK_TFLOPS, 4)}


def mfma_object(prof, dtype: str, one_sequence: bool, copy_gbs):
    """The `mfma` object: achieved TFLOP/s of every dense kernel of the profiled steps against the dense bf16 MFMA peak.  For the
    fp32 model (dtype "bf16slot") every fp32 product is three bf16 MFMAs and `achieved` counts the EXECUTED matrix flops."""
    mfma = {"peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "kernels": {},
            "note": None if dtype == "bf16" else
            "f32 model + bf16 slot: every fp32 product is three bf16 MFMAs (hi*hi + lo*hi + hi*lo); `achieved` counts the "
            "EXECUTED matrix flops (3 x 2 K N per row), i.e. the matrix cores' utilisation; algorithmic flops are a third"}
    for name, rec in sorted(prof.items()):
        fl = rec["meta"].get("flops") if isinstance(rec.get("meta"), dict) else None
        if not fl:
            continue
        tf = fl / (rec["avg_ms"] * 1e-3) / 1e12
        label = ("subsampling conv2 (hand-written implicit GEMM)" if name == "conv3x3s2"
                 else "subsampling conv2, split operands (hand-written implicit GEMM, 3 MFMAs per product)" if name == "conv3x3s2_split"
                 else "hand-written GEMM K x N [x batch] = " + name.split("_", 1)[1] if name.startswith("gemm_")
                 else "hand-written split-operand GEMM (3 MFMAs per fp32 product) K x N = " + name.split("_", 1)[1]
                 if name.startswith("gemm3_") else "hand-written fp32 GEMM (fp32 matrix cores: priced against the bf16 peak here) K x N = "
                 + name.split("_", 1)[1])
        ent = {"achieved": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
               "avg_us": round(rec["avg_ms"] * 1e3, 1), "launches": rec["n"]}
        # the other side of the roofline for the projections of ONE long sequence (c3): the activations cross HBM once
        # (in + out [+ residual]; the weights stay in L2), which for the N = 512 shapes takes longer than the
        # multiplies at the MFMA peak -- `frac` alone cannot reach 1 there ("Two-sided roofline")
        dims = name.split("_", 1)[1].split("x") if name.startswith("gemm_") else None
        if dims and one_sequence and copy_gbs and dtype == "bf16":
            K_, N_, Z_ = int(dims[0]), int(dims[1]), int(dims[2]) if len(dims) > 2 else 1
            rows_ = fl / (2.0 * K_ * N_ * Z_)
            n_out = N_ // 2 if (K_, N_) == (512, 1024) else N_                 # pointwise_conv1 + GLU writes half
            has_res = Z_ == 1 and N_ == 512 and K_ in (512, 1024, 2048)          # w_2, slot output, pointwise_conv2
            byts = rows_ * Z_ * (K_ + n_out + (N_ if has_res else 0)) * 2
            floor_mfma, floor_hbm = fl / (MFMA_PEAK_TFLOPS * 1e12), byts / (copy_gbs * 1e9)
            floor_spec = byts / (HBM_PEAK_GBS * 1e9)
            ent.update(hbm_bytes=int(byts), mfma_floor_us=round(floor_mfma * 1e6, 1),
                       hbm_floor_us_at_8tbs=round(floor_spec * 1e6, 1),
                       hbm_floor_us_at_measured_copy=round(floor_hbm * 1e6, 1),
                       # the roofline the contract prescribes: spec peaks on both sides (8 TB/s, 2.5 PFLOP/s) ...
                       frac_of_two_sided_roofline=round(max(floor_mfma, floor_spec) / (rec["avg_ms"] * 1e-3), 4),
                       # ... and against the copy rate this box sustains (what a perfect kernel could reach here)
                       frac_of_two_sided_roofline_at_measured_copy=round(max(floor_mfma, floor_hbm) / (rec["avg_ms"] * 1e-3), 4))
        mfma["kernels"][label] = ent
    return mfma


def whole_model_bf16_leg(feats32, device, steps: int, copy_gbs):
    """The same 30-minute file as one sequence through the whole-model-bf16 encoder (encoder-rtf.py --bf16): rounds 1-4's headline,
    kept as a secondary figure.  The reference's bidirectional wrapper cannot run this mode (it returns .float() into a bf16
    LayerNorm, rwkv_wrapper_bidirectional.py:55-56); here the slot returns the query dtype.  `steps` timed steps after 2 warm-up
    steps, then profiled steps for the scan's roofline and the GEMMs' matrix-core fractions in this mode."""
    from paper_accurate_fast_cheap_amd import profiling
    m2, _ = build_model("bf16", device)
    fb2 = feats32.to(device=device, dtype=torch.bfloat16)
    ln2 = torch.tensor([feats32.shape[1]], dtype=torch.int32, device=device)
    profiling.enable_recording(False)
    with torch.no_grad():
        for _ in range(2):
            m2.ctc_logprobs(m2._forward_encoder(fb2, ln2)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m2.ctc_logprobs(m2._forward_encoder(fb2, ln2)[0])
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / steps * 1e3
        profiling.enable(True)
        for _ in range(min(steps, 3)):
            m2.ctc_logprobs(m2._forward_encoder(fb2, ln2)[0])
        torch.cuda.synchronize()
        profiling.enable_recording(False)
    prof = profiling.summary()
    return {"workload": "the same file as one sequence, " + PRECISION["bf16"], "steps": steps, "ms_per_step": round(ms2, 3),
            "audio_sec_per_sec": round(feats32.shape[1] / 100.0 / (ms2 * 1e-3), 1),
            "roofline": scan_roofline(prof, copy_gbs), "mfma": mfma_object(prof, "bf16", True, copy_gbs)}


def spawn_ranks(n: int, argv, script: str = None) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as a child torchrun -- the parent has made
    no GPU call -- relay their output and return the child's exit code (wenet's recipe launches the same way,
    examples/gigaspeech/s0/run-pipeline-v3.sh:135-137)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16slot", choices=["bf16", "bf16slot"],
                    help="bf16slot (default): fp32 model with the bf16 time-mix slot -- the YAML default and what the paper's RTF sweep "
                         "ran (go-run-encoder-rtf.single-gpu-3x3-g5.sh:33-41 passes no --bf16), the precision the reference itself "
                         "can run; bf16: whole model bf16 (encoder-rtf.py --bf16), which the reference's bidirectional wrapper "
                         "cannot run (it returns .float() into a bf16 LayerNorm, rwkv_wrapper_bidirectional.py:55-56)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c2"],
                    help="c3 (default, the metric's workload): one 30-min file per GPU; c2: 5715 DEV-shaped utterances "
                         "(1-20 s) sharded over the GPUs, decode batch 64, CTC greedy tokens (parity-suite workload)")
    ap.add_argument("--chunk-size", type=int, default=0, help="frames per window; 0 = the whole file as one sequence")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--cpu-sample-frames", type=int, default=20000)
    ap.add_argument("--streams", type=int, default=0,
                    help="c2: decode batches in flight at once, each on a HIP stream of its own (0 = 2).  A batch of 64 short "
                         "utterances fills half the chip per kernel; independent batches side by side fill the rest -- same "
                         "kernels, same batches, same results")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the fp32-model / bf16-slot timing reported under 'extra'")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on one GPU)")
    args = ap.parse_args()

    one_gpu = os.environ.get("PAFC_BENCH_ONE_GPU") == "1"   # rehearsal of the N > 1 code path on a single-GPU box
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = torch.cuda.device_count()           # counts devices without initialising the GPU
        if have < args.gpus and not one_gpu:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (PAFC_BENCH_ONE_GPU=1 "
                     f"--dist-backend gloo rehearses the multi-rank path on one)")
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)              # before the process group: RCCL binds its communicator to THIS device
    device = torch.device("cuda", local_rank)
    rccl_ranks = None
    if world > 1:
        import torch.distributed as dist
        # RCCL; only used for the timing barrier and the max-reduce (wenet/utils/train_utils.py:208 is the reference's call)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)
        ones = torch.ones(1, device=device if args.dist_backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(ones)                      # proof for the record that the collective saw every rank
        rccl_ranks = int(ones.item())

    from paper_accurate_fast_cheap_amd import _lib, profiling
    _lib.lib()  # fail loudly, before anything else, if the HIP extension is missing
    if os.environ.get("PAFC_BENCH_WATCHDOG"):      # diagnostics: dump every thread's Python stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PAFC_BENCH_WATCHDOG"]), exit=True)

    stage("library loaded; building the model")
    model, configs = build_model(args.dtype, device)
    conf = configs["encoder_conf"]
    if args.workload == "c3" and args.chunk_size > 0:
        model.encoder.graph_cache_size = 2     # windowed long-form: dozens of identical (B, chunk) batches -> hipGraph replay
    greedy = None
    if args.workload == "c3":
        wave = synthetic_waveform(AUDIO_SECONDS, 777 + rank)
        feats, fbank_ms = front_end(wave, device)          # outside the timed region, as in encoder-rtf.py:347-353
        assert feats.shape == (1, FRAMES, 80)
        feats32 = feats.cpu()
        if args.dtype == "bf16":
            feats = feats.to(torch.bfloat16)
        batches = list(windows(feats, args.chunk_size, args.batch_size))   # resident in HBM before timing
    else:
        from paper_accurate_fast_cheap_amd.transformer.search import ctc_greedy_search as greedy
        batches, fbank_ms, feats32 = c2_batches(device, torch.bfloat16 if args.dtype == "bf16" else torch.float32, rank, world)
    frames_per_step = int(sum(int(l.sum()) for _, l in batches))
    progress = None
    if args.workload == "c2" and rank == 0:   # a long ragged run is never silent: one line per pass over the shard
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        progress = open(os.path.join(ROOT, "gpurun_out", "bench_c2_progress.log"), "a")

    # batches in flight: two for the ragged decode batches of c2, three for the window batches of --chunk-size (their launch-bound
    # stretches overlap; measured 1-4, see windows_leg); the one-sequence headline has one batch per step
    nstreams = (args.streams or (2 if args.workload == "c2" else 3)) if (args.workload == "c2" or args.chunk_size > 0) else 1
    if args.chunk_size > 0 and nstreams > 1:
        model.encoder.graph_cache_size = max(model.encoder.graph_cache_size, 2 * nstreams)
    step, last_tokens = make_step(model, batches, device, nstreams, greedy, progress)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def rss_mb():
        import resource
        return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0     # Linux: KiB

    stage(f"inputs resident ({frames_per_step} frames per step); warm-up")
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        stage("warm-up done; timed steps")
        torch.cuda.reset_peak_memory_stats(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        max_vram_mb = torch.cuda.max_memory_allocated(device) / 1024 / 1024    # encoder-rtf.py:544 (MB, this rank)
        # per-kernel event timers (roofline / mfma objects) run on extra steps OUTSIDE the timed region: recording events
        # on the hot stream costs a few microseconds per timed op, which the headline number should not carry
        profiling.enable(True)
        for _ in range(min(args.steps, 3)):
            step()
        torch.cuda.synchronize()
        profiling.enable_recording(False)
    prof = profiling.summary()
    stage(f"timed region {elapsed * 1e3 / args.steps:.3f} ms per step; profiled steps done")

    if world > 1:
        rdev = device if args.dist_backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if world > 1 and args.workload == "c2":   # ranks hold different shards: total frames = sum over ranks
        t = torch.tensor([frames_per_step], device=device if args.dist_backend == "nccl" else torch.device("cpu"),
                         dtype=torch.float64)
        dist.all_reduce(t)
        total_frames = float(t.item())
    else:
        total_frames = frames_per_step * world
    audio_s = total_frames / 100.0 * args.steps
    value = audio_s / elapsed

    # achievable HBM bandwidth on this box: a 1 GiB device-to-device copy (bytes read + written per second), so that the
    # scan's rate can be read against the measured ceiling as well as against the 8 TB/s specification
    copy_gbs = None
    if rank == 0:
        a = torch.empty(1 << 30, dtype=torch.uint8, device=device)
        bq = torch.empty_like(a)
        for _ in range(3):
            bq.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            bq.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = round(10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del a, bq

    roofline = scan_roofline(prof, copy_gbs)
    # MFMA utilisation of the dense kernels, from the same event timers: achieved TFLOP/s against the dense bf16 peak
    mfma = mfma_object(prof, args.dtype, args.workload == "c3" and args.chunk_size <= 0, copy_gbs)

    out = {
        "metric": "audio-sec/sec (1/RTF) GigaSpeech long-form encode",
        "value": round(value, 2), "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if args.workload == "c3" else "strong",
        "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32+bf16-slot", "data": "synthetic",
        "config": {"workload": ("c2: 5715 synthetic DEV-shaped utterances (1-20 s, HIP fbank features) sharded over the "
                                "GPUs by length, batches of 64, encoder + CTC log-softmax + greedy tokens, "
                                if args.workload == "c2" else
                                "c3: 30-min synthetic 16 kHz file per GPU -> HIP fbank (179998 x 80, outside the timed "
                                "region as in encoder-rtf.py) -> ")
                               + ("" if args.workload == "c2" else "encoded as one sequence B=1" if args.chunk_size <= 0
                                  else f"windows chunk_size={args.chunk_size} x batch {args.batch_size}")
                               + ", 12-layer bidirectional RWKV-v6 Conformer encoder (512d, 8x64 heads) + CTC(5000) "
                                 "log-softmax; random-init weights (seed 777)",
                   "frames_per_step": frames_per_step, "parallelism": f"dp{world} (independent files, no collective)"},
        # the reference harness's own quantities (wenet/bin/encoder-rtf.py:526-549): final_rtf = wall / audio seconds,
        # "minutes of audio processed per sec", max VRAM (torch allocator peak over the timed region) and max host RSS
        "rtf_harness": {"final_rtf": round(elapsed / audio_s, 9), "minutes_per_sec": round(audio_s / 60.0 / elapsed, 3),
                        "total_frames": int(total_frames * args.steps), "total_elapsed": round(elapsed, 6),
                        "max_vram_GB": round(max_vram_mb / 1024.0, 3), "max_vram_MB": round(max_vram_mb, 2),
                        "max_cpu_ram_MB": round(rss_mb(), 2)},
        "rccl_ranks": rccl_ranks,
        "streams": nstreams,
        "roofline": roofline, "mfma": mfma,
        "front_end": {"kernel": "fbank (HIP, fp32 MFMA DFT)", "ms_per_file": round(fbank_ms, 3),
                      "audio_sec_per_sec": round(frames_per_step / 100.0 / (fbank_ms * 1e-3), 1),
                      "audio_sec_per_sec_encoder_plus_fbank": round(
                          frames_per_step / 100.0 / (elapsed / args.steps + fbank_ms * 1e-3), 2)},
    }
    # secondary legs beside the headline (N = 1, c3 one-sequence only; --no-extra skips them): BASELINE configs[1] (c2) and the
    # paper's window shape in the HEADLINE precision on the package's defaults (no knob is set anywhere in this file), the
    # streaming legs (configs[2]; a uni-directional whole-bf16 model, which the reference's uni wrapper does run:
    # rwkv_wrapper.py:57-83 returns the query dtype), and the other precision mode's one-sequence figure
    out["extra"] = None
    failed_legs = []
    if rank == 0 and world == 1 and args.workload == "c3" and args.chunk_size <= 0 and not args.no_extra:
        del batches
        out["extra"] = {}

        def leg(name, fn, *a, **kw):      # a failure in a leg is reported in its place, never costs the headline line, and
            stage(f"leg {name}")
            try:                           # turns the exit code non-zero after the line is printed
                out["extra"][name] = fn(*a, **kw)
            except (RuntimeError, ValueError, AssertionError, KeyError, OSError) as e:
                out["extra"][name] = {"error": f"{type(e).__name__}: {e}"[:300]}
                failed_legs.append(name)
                try:
                    torch.cuda.synchronize()
                except RuntimeError:
                    pass
        other = "bf16" if args.dtype == "bf16slot" else "bf16slot"
        if other == "bf16":
            leg("whole_model_bf16", whole_model_bf16_leg, feats32, device, min(args.steps, 10), copy_gbs)
        leg("c2", c2_leg, model, device, args.dtype)
        wfeats = feats32.to(device=device, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
        leg("windows_2000x8", windows_leg, model, wfeats, device, dtype=args.dtype)
        leg("two_files_in_flight", two_files_leg, model, wfeats, device, dtype=args.dtype)
        del wfeats
        leg("streaming", streaming_leg, feats32, device)
        leg("streaming_lookahead", streaming_lookahead_leg, feats32, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        stage("cpu baseline")
        out["cpu_baseline"] = cpu_baseline(model, feats32, conf, min(args.cpu_sample_frames, FRAMES))
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if failed_legs:
        sys.exit(f"bench.py: secondary leg(s) failed: {', '.join(failed_legs)} (the headline line above is complete)")


if __name__ == "__main__":
    main()
