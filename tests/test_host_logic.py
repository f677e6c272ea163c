"""CPU-only tests of the host side: plugin surface, parameter names, initialisation, masks, loaders, sharding."""
import json
import os
import subprocess
import sys

import pytest
import torch

from oracle import encoder_oracle as EO
from tests.conftest import ROOT, load_golden


def test_full_encoder_state_dict_equals_reference_spec():
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    from tests import synth
    g = load_golden("encoder_full_spec")
    enc = ConformerEncoder(80, **g["conf"])
    assert synth.spec_of(enc.state_dict()) == g["spec"]          # names, shapes AND dtypes (bf16 slot parameters)
    assert sum(p.numel() for p in enc.parameters()) == g["n_params"] == 97461248
    assert enc.output_size() == 512 and enc.embed.subsampling_rate == 4 and enc.embed.right_context == 6


def test_registry_keys_and_constructor_signature():
    from paper_accurate_fast_cheap_amd.utils.class_utils import WENET_ATTENTION_CLASSES, install_into
    want = {"rwkv_tmix60", "rwkv_tmix60_bidirectional", "rwkv_tmix60_bidirectional2", "rwkv_tmix60_dir_layer_drop",
            "rwkv_tmix60_dir_layer_drop_both"}
    assert want <= set(WENET_ATTENTION_CLASSES)
    for k in want:  # positional ctor the reference encoder builds, encoder.py:553-561 + layer_id
        m = WENET_ATTENTION_CLASSES[k](64, 128, 2, "rwkv", "bi", 2048, True, 1)
        assert any(n.endswith("tmix_block.time_faaaa") for n, _ in m.named_parameters())

    class FakeRef:
        WENET_ATTENTION_CLASSES = {"selfattn": object}
    install_into(FakeRef)
    assert want <= set(FakeRef.WENET_ATTENTION_CLASSES) and "selfattn" in FakeRef.WENET_ATTENTION_CLASSES


def test_unsupported_slots_fail_loudly():
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    with pytest.raises(NotImplementedError):
        ConformerEncoder(80, output_size=128, attention_heads=2, selfattention_layer_type="rel_selfattn")
    with pytest.raises(NotImplementedError):
        ConformerEncoder(80, output_size=128, attention_heads=2, num_langs=3)


def test_tmix_init_matches_reference_closed_form():
    from paper_accurate_fast_cheap_amd.rwkv_v6.tmix import RWKV_Tmix_x060c
    g = load_golden("tmix_init")
    for layer_id, ref in g["init"].items():
        blk = RWKV_Tmix_x060c(g["head_size"], g["n_layers"], g["n_embd"], g["n_embd"], layer_id)
        sd = blk.state_dict()
        for k, v in ref.items():
            torch.testing.assert_close(sd[k], v, rtol=1e-6, atol=1e-6, msg=f"layer {layer_id} {k}")
        assert float(sd["time_maa_rkvw_w1"].abs().max()) == 0 and float(sd["time_decay_w1"].abs().max()) == 0


def test_masks_match_oracle():
    from paper_accurate_fast_cheap_amd.utils.mask import make_pad_mask
    lens = torch.tensor([5, 1, 9, 0])
    assert torch.equal(make_pad_mask(lens, 9), EO.make_pad_mask(lens, 9))
    assert torch.equal(make_pad_mask(lens), EO.make_pad_mask(lens))


def test_greedy_collapse_matches_oracle():
    from paper_accurate_fast_cheap_amd.transformer.search import remove_duplicates_and_blank
    g = torch.Generator().manual_seed(3)
    for _ in range(50):
        hyp = torch.randint(0, 4, (int(torch.randint(0, 30, (1,), generator=g)),), generator=g).tolist()
        assert remove_duplicates_and_blank(hyp, 0) == EO.remove_duplicates_and_blank(hyp, 0)
    assert remove_duplicates_and_blank([], 0) == []
    assert remove_duplicates_and_blank([0, 0, 3, 3, 0, 3, 2, 2], 0) == [3, 3, 2]


def test_cmvn_loaders(tmp_path):
    from paper_accurate_fast_cheap_amd.utils.cmvn import load_cmvn
    sums, sumsq, n = [2.0, 4.0, 0.0], [6.0, 10.0, 0.0], 2
    (tmp_path / "c.json").write_text(json.dumps({"mean_stat": sums, "var_stat": sumsq, "frame_num": n}))
    (tmp_path / "c.kaldi").write_text("[ 2 4 0 2\n 6 10 0 0 ]\n")
    for f, is_json in (("c.json", True), ("c.kaldi", False)):
        mean, istd = load_cmvn(str(tmp_path / f), is_json)
        assert mean.tolist() == [1.0, 2.0, 0.0]
        assert istd[0] == pytest.approx(1 / (3 - 1) ** 0.5) and istd[1] == pytest.approx(1.0)
        assert istd[2] == pytest.approx(1e10)  # variance floored at 1e-20 (utils/cmvn.py:38-39)


def test_checkpoint_roundtrip_reference_format(tmp_path):
    from paper_accurate_fast_cheap_amd.utils.checkpoint import load_checkpoint, save_checkpoint
    from paper_accurate_fast_cheap_amd.utils.init_model import init_model
    g = load_golden("encoder_reduced_bf16slot")
    cfg = lambda: dict(encoder="conformer", encoder_conf=dict(g["conf"]), input_dim=80, output_dim=50, ctc="ctc",
                       ctc_conf={"ctc_blank_id": 0}, model_conf={"ctc_weight": 0.3, "lsm_weight": 0.1}, dataset_conf={},
                       tokenizer_conf={"special_tokens": {"<blank>": 0, "<sos>": 2, "<eos>": 2}})

    class A:
        checkpoint = None
    m1, _ = init_model(A(), cfg())
    save_checkpoint(m1, str(tmp_path / "m.pt"), {"step": 7, "num_seen_frames": 123})
    blob = torch.load(str(tmp_path / "m.pt"), weights_only=False)
    assert set(blob) == {"model0"} and any(k.startswith("encoder.encoders.0.self_attn.rwkv_wrapper_forward.") for k in blob["model0"])
    A.checkpoint = str(tmp_path / "m.pt")
    torch.manual_seed(1)
    m2, c2 = init_model(A(), cfg())
    assert c2["step"] == 7 and c2["num_seen_frames"] == 123 and c2["init_infos"]["step"] == 7
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert m2.lsl_enc is False and m2.add_cat_embs is False and m2.cat_labels == [] and m2.sos == 2
    assert isinstance(load_checkpoint(m2, A.checkpoint), dict)


def test_bench_windows_follow_the_reference_batcher():
    sys.path.insert(0, ROOT)
    import bench
    feats = torch.arange(1, 2 * 23 * 2 + 1, dtype=torch.float32).view(1, 46, 2).repeat(1, 1, 40)  # (1, 46, 80)
    out = list(bench.windows(feats, 10, 2))
    assert [tuple(f.shape) for f, _ in out] == [(2, 10, 80), (2, 10, 80), (1, 10, 80)]
    assert [l.tolist() for _, l in out] == [[10, 10], [10, 10], [6]]
    assert torch.equal(out[2][0][0, :6], feats[0, 40:46]) and float(out[2][0][0, 6:].abs().max()) == 0
    assert bench.FRAMES == 179998
    whole = list(bench.windows(feats, 0, 8))
    assert len(whole) == 1 and whole[0][1].tolist() == [46]


def test_decode_batches_sorted_by_length_longest_batch_first():
    from paper_accurate_fast_cheap_amd.utils.sharding import decode_batches
    lens = [int(x) for x in torch.randint(100, 2001, (331,), generator=torch.Generator().manual_seed(5))]
    units = list(range(0, 331, 2))
    plan = decode_batches(units, lens, 64)
    assert sorted(i for b in plan for i in b) == units                                  # every unit once
    assert [len(b) for b in plan] == [len(units) % 64 or 64] + [64] * (len(units) // 64 - (len(units) % 64 == 0))
    flat = [lens[i] for b in reversed(plan) for i in b]
    assert flat == sorted(flat)                                                           # neighbours in length share a batch
    assert max(lens[i] for i in plan[0]) == max(lens[i] for i in units)                   # ... and the longest batch is issued first


def test_shard_units_balanced_and_complete():
    from paper_accurate_fast_cheap_amd.utils.sharding import shard_units
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(100, 2001, (5715,), generator=g).tolist()   # GigaSpeech DEV size, segment filter range
    for world in (1, 2, 4, 8):
        parts = [shard_units(lens, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= 2000
    assert shard_units([], 0, 2) == [] and shard_units([5], 1, 2) == []


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["PAFC_ROOT"])
from paper_accurate_fast_cheap_amd.utils.sharding import shard_units, gather_results
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lens = [(7 * i) % 13 + 1 for i in range(29)]
mine = shard_units(lens, rank, world)
local = {i: [lens[i], i * i] for i in mine}          # stand-in for per-unit token lists
t = torch.tensor([float(sum(lens[i] for i in mine))])
dist.all_reduce(t, op=dist.ReduceOp.MAX)              # the bench's max-over-ranks reduction
merged = gather_results(local, world)
if rank == 0:
    assert sorted(merged) == list(range(29)) and all(merged[i] == [lens[i], i * i] for i in merged)
    assert t.item() >= sum(lens) / world
    print("GLOO_OK")
else:
    assert merged is None
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, PAFC_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GLOO_OK" in out.stdout


def test_feats_batcher_windows_like_the_reference():
    """recognize_wav2.py:323-351: windows of chunk_size frames in batches of batch_size; only the last window is padded
    (with zeros) and only its length is shortened; concatenating the valid frames gives the input back."""
    from paper_accurate_fast_cheap_amd.utils.longform import feats_batcher, window_offsets_ms
    x = torch.arange(1 * 1037 * 3, dtype=torch.float32).view(1, 1037, 3) + 1.0
    batches = list(feats_batcher(x, 100, 4))
    assert [tuple(b.shape) for b, _ in batches] == [(4, 100, 3), (4, 100, 3), (3, 100, 3)]
    assert [l.tolist() for _, l in batches] == [[100] * 4, [100] * 4, [100, 100, 37]]
    assert all(l.dtype == torch.int32 for _, l in batches)
    back = torch.cat([b[i, :int(l[i])] for b, l in batches for i in range(b.shape[0])])
    assert torch.equal(back, x[0])
    assert float(batches[-1][0][2, 37:].abs().max()) == 0.0
    # exact multiples: no padding, full lengths; a file shorter than one window: one short window
    assert [l.tolist() for _, l in feats_batcher(x[:, :800], 100, 4)] == [[100] * 4, [100] * 4]
    assert [l.tolist() for _, l in feats_batcher(x[:, :42], 100, 4)] == [[42]]
    assert window_offsets_ms(3, 2000) == [0.0, 20000.0, 40000.0]
    # merged launches: whole multiples of the batch, at most merge_frames input frames, never below one batch
    from paper_accurate_fast_cheap_amd.utils.longform import merged_batch_size
    assert merged_batch_size(2000, 8, 0) == 8 and merged_batch_size(2000, 8, 180000) == 88
    assert merged_batch_size(2000, 1, 180000) == 90 and merged_batch_size(100000, 4, 180000) == 4
    assert merged_batch_size(9000, 14, 180000) == 14 and merged_batch_size(4000, 10, 200000) == 50


def test_few_rows_gemm_is_scoped_to_a_chunk_step():
    """hip_ops.skinny_ok: the few-rows kernel (csrc/gemm_skinny.hip) is offered only inside hip_ops.chunk_step() -- offline
    inputs that merely happen to be short keep the kernels their goldens were recorded with -- and only for shapes it takes."""
    from paper_accurate_fast_cheap_amd import hip_ops
    assert not hip_ops.skinny_ok(64, 512, 512)
    with hip_ops.chunk_step():
        assert hip_ops.skinny_ok(64, 512, 512) and hip_ops.skinny_ok(78, 1024, 512, glu=True)
        assert hip_ops.skinny_ok(hip_ops.SKINNY_MAX_ROWS, 512, 2048) and not hip_ops.skinny_ok(hip_ops.SKINNY_MAX_ROWS + 1, 512, 2048)
        assert not hip_ops.skinny_ok(64, 512, 48)            # K in whole 32-deep steps
        assert not hip_ops.skinny_ok(64, 1000, 512, glu=True) and not hip_ops.skinny_ok(0, 512, 512)
        with hip_ops.chunk_step():                           # re-entrant
            pass
        assert hip_ops.skinny_ok(64, 512, 512)
    assert not hip_ops.skinny_ok(64, 512, 512)


def test_inference_plans_follow_fused_optimizer_updates():
    """Derived weight copies of the inference plans are keyed on (storage, Tensor._version) AND the parameter epoch: a fused
    optimizer moves parameters without touching `_version` (checked here), so without the epoch a CV pass after a training
    step would multiply the old weights.  train() / eval() and train_step's optimizer step bump it."""
    from paper_accurate_fast_cheap_amd import hip_ops
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    from paper_accurate_fast_cheap_amd.transformer.fused import LayerPlan
    g = load_golden("encoder_reduced_f32")
    enc = ConformerEncoder(80, **g["conf"]).eval()
    layer = enc.encoders[0]
    plan = LayerPlan(layer)
    wo = layer.self_attn.rwkv_wrapper_forward.tmix_block.output.weight
    b2 = layer.feed_forward.w_2.bias
    assert torch.equal(plan.Wo[:, :128], wo * 0.5) and torch.equal(plan.b2, b2 * layer.ff_scale)
    enc.train()
    opt = torch.optim.Adam(enc.parameters(), lr=0.05, fused=True)
    v0 = wo._version
    for p in enc.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert wo._version == v0                         # the hazard: the update is invisible to the version counter
    assert not torch.equal(plan.Wo[:, :128], wo.detach() * 0.5)
    enc.eval()                                       # mode switch = new parameter epoch
    plan.refresh()
    assert torch.equal(plan.Wo[:, :128], wo.detach() * 0.5) and torch.equal(plan.b2, b2.detach() * layer.ff_scale)
    e0 = hip_ops.param_epoch()
    hip_ops.bump_param_epoch()
    assert hip_ops.param_epoch() == e0 + 1


def test_graph_cache_token_follows_every_kind_of_parameter_change():
    """BaseEncoder._weights_token -- what the hipGraph cache of `forward` is keyed on besides (shape, stream): it moves with an
    in-place update (Tensor._version), load_state_dict, .to() (new storage) and the parameter epoch; it stays put between two
    forwards that change nothing.  (The graphs themselves need a GPU: tests/test_encoder_gpu.py.)"""
    from paper_accurate_fast_cheap_amd import hip_ops
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    g = load_golden("encoder_reduced_f32")
    enc = ConformerEncoder(80, **g["conf"]).eval()
    t0 = enc._weights_token()
    assert enc._weights_token() == t0
    with torch.no_grad():
        enc.encoders[1].feed_forward.w_2.weight.mul_(2.0)
    t1 = enc._weights_token()
    assert t1 != t0
    enc.load_state_dict({k: v.clone() for k, v in enc.state_dict().items()})
    t2 = enc._weights_token()
    assert t2 != t1 and t2[0] > t1[0]                 # a load is a new epoch
    enc.to(torch.float64)
    t3 = enc._weights_token()
    assert t3[0] > t2[0]                              # new storage: new epoch, and the tensor list is looked up again
    assert all(t.dtype == torch.float64 for t in enc._wt_tensors if t.is_floating_point())
    hip_ops.bump_param_epoch()
    assert enc._weights_token()[0] == t3[0] + 1


def test_derived_fill_is_a_no_op_without_a_gpu():
    from paper_accurate_fast_cheap_amd import hip_ops
    f = hip_ops.DerivedFill()
    f.use()
    if not torch.cuda.is_available():
        assert f.event is None


def test_encoder_copies_and_pickles_without_its_runtime_state():
    """copy.deepcopy (EMA copies, model averaging), pickle (spawn) and torch.save of a model keep parameters and configuration and
    drop what is rebuilt on demand: captured graphs, the fused executor's plans (their DerivedFill events cannot be copied), the
    subsampling module's derived weight copies.  The load_state_dict post-hook is a module-level function."""
    import copy
    import io
    import pickle
    from paper_accurate_fast_cheap_amd import hip_ops
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    g = load_golden("encoder_reduced_f32")
    enc = ConformerEncoder(80, **g["conf"]).eval()
    enc._graphs = {"a shape": "seen"}
    enc._fused_plan = object()
    enc.embed._nhwc_fill = hip_ops.DerivedFill()
    enc.embed._w_lin = torch.zeros(3)
    for other in (copy.deepcopy(enc), pickle.loads(pickle.dumps(enc))):
        assert other._graphs == {} and other._fused_plan is None
        assert not hasattr(other.embed, "_w_lin") and not hasattr(other.embed, "_nhwc_fill")
        assert other.graph_cache_size == enc.graph_cache_size and other.fp32_split_operands == enc.fp32_split_operands
        for (ka, a), (kb, b) in zip(enc.state_dict().items(), other.state_dict().items()):
            assert ka == kb and torch.equal(a, b)
    buf = io.BytesIO()
    torch.save(enc, buf)
    assert enc._graphs == {"a shape": "seen"}                    # the original keeps its state
    f = hip_ops.DerivedFill()
    assert copy.deepcopy(f).event is None and pickle.loads(pickle.dumps(f)).seen == set()


def test_subsampled_length_is_the_mask_slicing():
    """Conv2dSubsampling4.subsampled_length (what the hipGraph cache keys its all-rows-full decision on) counts exactly what
    x_mask[:, :, 2::2][:, :, 2::2] keeps (subsampling.py:201-226), for every input length."""
    from paper_accurate_fast_cheap_amd.transformer.embedding import RelPositionalEncoding
    from paper_accurate_fast_cheap_amd.transformer.subsampling import Conv2dSubsampling4
    from paper_accurate_fast_cheap_amd.utils.mask import make_pad_mask
    sub = Conv2dSubsampling4(80, 16, 0.0, RelPositionalEncoding(16, 0.0))
    for n in range(0, 70):
        m = ~make_pad_mask(torch.tensor([n]), 70).unsqueeze(1)
        assert int(m[:, :, 2::2][:, :, 2::2].sum()) == sub.subsampled_length(n), n
    assert sub.subsampled_length(179998) == 44998 and sub.subsampled_length(179995) == 44998 and sub.subsampled_length(179994) == 44997


def test_multi_stream_safe_only_for_all_own_kernel_passes():
    """BaseEncoder.multi_stream_safe: batches may be in flight on several HIP streams only when every GEMM of the pass is one of the
    package's kernels (the framework's library GEMM never finishes when two streams issue it: DESIGN.md "the c2 stall").  True for
    the fused executor over eligible layers -- RWKV slots, the Mamba-2 block --, False for the module path (switched off, a
    batch-norm conv module, training mode); utils.longform then keeps one stream."""
    import bench
    from paper_accurate_fast_cheap_amd.transformer.encoder import ConformerEncoder
    from paper_accurate_fast_cheap_amd.utils import longform
    conf = dict(bench.encoder_conf(), num_blocks=2, output_size=128, attention_heads=2, linear_units=256)
    enc = ConformerEncoder(80, **conf).eval()
    assert enc.multi_stream_safe()
    enc.fused_inference = False
    assert not enc.multi_stream_safe()
    assert ConformerEncoder(80, **dict(conf, selfattention_layer_type="mamba_att", rnn_att_version="mamba2")).eval().multi_stream_safe()
    assert not ConformerEncoder(80, **dict(conf, cnn_module_norm="batch_norm")).eval().multi_stream_safe()
    assert not ConformerEncoder(80, **conf).train().multi_stream_safe()

    class M:
        encoder = enc
    assert not longform._multi_stream_safe(M()) and not longform._multi_stream_safe(object())


def test_grouped_weights_keep_names_values_checkpoints_and_optimizer_state(tmp_path):
    """hip_ops._weight_group (round 6: the r / k / v weights of a time-mix block as ONE batched operand): bf16 parameters are moved
    into one buffer (`p.data = group[i]`) the first time they are asked for together inside train_shadows().  The Parameter
    objects, their values and their state_dict keys stay; an in-place update (an optimizer step, load_state_dict) is seen through
    the group; a checkpoint round trip and a `.to()` that breaks the layout are mended on the next call; outside train_shadows()
    there is no group.  fp32 parameters get a grouped bf16 COPY that is refreshed when the context is entered."""
    from paper_accurate_fast_cheap_amd import hip_ops
    torch.manual_seed(0)
    m = torch.nn.ModuleDict({n: torch.nn.Linear(64, 64, bias=False) for n in ("receptance", "key", "value")}).to(torch.bfloat16)
    ws = [m[n].weight for n in ("receptance", "key", "value")]
    before = [w.detach().clone() for w in ws]
    ids = [id(w) for w in ws]
    assert hip_ops._weight_group(ws) is None                                  # no context: nothing vouches for a grouped view
    with hip_ops.train_shadows():
        g = hip_ops._weight_group(ws)
        assert g.shape == (3, 64, 64) and g.is_contiguous()
        assert [id(w) for w in ws] == ids and all(isinstance(w, torch.nn.Parameter) for w in ws)
        assert all(w.data_ptr() == g[i].data_ptr() and torch.equal(w.detach(), before[i]) for i, w in enumerate(ws))
        assert hip_ops._weight_group(ws).data_ptr() == g.data_ptr()           # found again, nothing re-made
        opt = torch.optim.SGD(m.parameters(), lr=0.5)
        for w in ws:
            w.grad = torch.ones_like(w)
        opt.step()                                                            # in place: the group sees it
        assert all(torch.equal(hip_ops._weight_group(ws)[i], w.detach()) and not torch.equal(w.detach(), before[i]) for i, w in enumerate(ws))
    assert sorted(m.state_dict()) == ["key.weight", "receptance.weight", "value.weight"]
    torch.save(m.state_dict(), tmp_path / "sd.pt")
    sd = torch.load(tmp_path / "sd.pt")
    m2 = torch.nn.ModuleDict({n: torch.nn.Linear(64, 64, bias=False) for n in ("receptance", "key", "value")}).to(torch.bfloat16)
    m2.load_state_dict(sd)
    assert all(torch.equal(m2[n].weight, m[n].weight) for n in ("receptance", "key", "value"))
    m.load_state_dict({k: torch.zeros_like(v) for k, v in sd.items()})        # copies in place: still one buffer
    with hip_ops.train_shadows():
        g = hip_ops._weight_group(ws)
        assert float(g.abs().max()) == 0.0 and all(w.data_ptr() == g[i].data_ptr() for i, w in enumerate(ws))
    m.to(torch.float32).to(torch.bfloat16)                                    # new storage per parameter: the layout is gone ...
    ws = [m[n].weight for n in ("receptance", "key", "value")]
    with hip_ops.train_shadows():
        g = hip_ops._weight_group(ws)                                         # ... and mended here
        assert all(w.data_ptr() == g[i].data_ptr() for i, w in enumerate(ws))
    # fp32 parameters: a grouped bf16 copy, refreshed on entry
    f = [torch.nn.Parameter(torch.randn(64, 64)) for _ in range(3)]
    with hip_ops.train_shadows():
        g = hip_ops._weight_group(f)
        assert g.dtype == torch.bfloat16 and all(torch.equal(g[i], w.detach().to(torch.bfloat16)) and w.data_ptr() != g[i].data_ptr() for i, w in enumerate(f))
    with torch.no_grad():
        f[1].mul_(2.0)
    with hip_ops.train_shadows():
        g2 = hip_ops._weight_group(f)
        assert g2.data_ptr() == g.data_ptr() and torch.equal(g2[1], f[1].detach().to(torch.bfloat16))
