"""CPU restatement of the reference's encoder hot path around the WKV op.

TEST INFRASTRUCTURE ONLY (see oracle/wkv6_oracle.c header for the rule): only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

Everything here is a *functional* restatement: plain functions over a
reference-format state_dict (key names of SURVEY.md section 8(b)), torch CPU
tensors as the arithmetic library, no nn.Module, no autograd, eval mode
(dropout = identity).  Each function cites the reference lines it follows
(paths relative to /root/reference).  The restatement is pinned against
goldens captured from the reference's own Python modules in this container
(tests/golden/make_goldens.py -> tests/golden/*.pt, checked by
tests/test_oracle_goldens.py); the WKV op inside is oracle/wkv6_oracle.c.

dtype behaviour mirrors the reference: a module's arithmetic runs in the dtype
its parameters are stored in, every torch op rounds its result to that dtype.
"""
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from oracle import wkv6_oracle

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------
# RWKV-v6 time-mix block
# ----------------------------------------------------------------------------
def tmix_x060c(x: torch.Tensor, sd: SD, p: str, head_size: int) -> torch.Tensor:
    """RWKV_Tmix_x060c.forward, wenet/rwkv_v6/src/model.py:271-325.

    x: (B, T, C) in the block's dtype.  p: key prefix ending in 'tmix_block.'.
    """
    B, T, C = x.shape
    g = lambda n: sd[p + n]
    # model.py:274  time_shift = ZeroPad2d((0,0,1,-1)): x_{t-1}, zero at t=0 (model.py:262)
    xx = F.pad(x, (0, 0, 1, -1)) - x
    # model.py:276-278
    xxx = x + xx * g("time_maa_x")
    xxx = torch.tanh(xxx @ g("time_maa_rkvw_w1")).view(B * T, 4, -1).transpose(0, 1)
    xxx = torch.bmm(xxx, g("time_maa_rkvw_w2")).view(4, B, T, C)
    # model.py:280-284
    mr, mk, mv, mw = xxx.unbind(dim=0)
    r = x + xx * (g("time_maa_r") + mr)
    k = x + xx * (g("time_maa_k") + mk)
    v = x + xx * (g("time_maa_v") + mv)
    w = x + xx * (g("time_maa_w") + mw)
    # model.py:286-289
    r = F.linear(r, g("receptance.weight"))
    k = F.linear(k, g("key.weight"))
    v = F.linear(v, g("value.weight"))
    w = g("time_decay") + torch.tanh(w @ g("time_decay_w1")) @ g("time_decay_w2")
    # model.py:296-299 -> WKV_6 / WKV_6_FP32 (model.py:108-133,161-187): contiguous, same dtype
    u = g("time_faaaa")
    assert C % head_size == 0 and u.shape == (C // head_size, head_size)
    y = wkv6_oracle.forward(r.contiguous(), k.contiguous(), v.contiguous(), w.contiguous(), u.contiguous())
    # model.py:323-324: ln_x is a plain LayerNorm over C (not per-head), then output proj
    y = F.layer_norm(y, (C,), g("ln_x.weight"), g("ln_x.bias"), 1e-5)
    return F.linear(y, g("output.weight"))


def rwkv_wrapper(query: torch.Tensor, sd: SD, p: str, head_size: int, do_bfloat16: bool,
                 cache: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """RWKV_TmixWrapper.forward, wenet/rwkv_v6/rwkv_wrapper.py:57-83.  p ends in 'self_attn.' (uni)
    or 'self_attn.rwkv_wrapper_forward.' etc."""
    qd = query.dtype
    if do_bfloat16:
        query = query.to(torch.bfloat16)
    y = tmix_x060c(query, sd, p + "tmix_block.", head_size)
    if do_bfloat16:
        y = y.to(qd)
    if cache is None:
        cache = torch.zeros((0, 0, 0, 0))
    return y, cache


def rwkv_wrapper_bidirectional(query, sd: SD, p: str, head_size: int, do_bfloat16: bool, cache=None,
                               out_as_query: bool = False):
    """RWKV_TmixWrapper_bidirectional.forward, rwkv_wrapper_bidirectional.py:30-64 (same arithmetic as
    rwkv_wrapper_bidirectional2.py:95-150).  The inner wrappers have do_bfloat16 forced False (:27-28);
    the flip is over the whole padded length (:44); the sum and /2 happen in bf16 (:49), then .float() (:55-56).

    out_as_query: the ONE deliberate departure, for the whole-model-bf16 mode of `encoder-rtf.py --bf16`
    (encoder-rtf.py:424-426): there the reference's `.float()` hands fp32 to a LayerNorm with bf16 parameters and
    raises, so the mode is defined as "the slot returns the query dtype" -- identical to the reference for an fp32
    query, and the only change that lets its own graph run for a bf16 one."""
    qd = query.dtype
    if do_bfloat16:
        query = query.to(torch.bfloat16)
    x = query
    xf = torch.flip(x, [1])
    a, _ = rwkv_wrapper(x, sd, p + "rwkv_wrapper_forward.", head_size, False, cache)
    b, _ = rwkv_wrapper(xf, sd, p + "rwkv_wrapper_backward.", head_size, False, cache)
    out = (a + torch.flip(b, [1])) / 2
    if do_bfloat16:
        out = out.to(qd) if out_as_query else out.float()
    if cache is None:
        cache = torch.zeros((0, 0, 0, 0))
    return out, cache


def draw_direction_dropout(both: bool, p: float = 0.2) -> Tuple[bool, Optional[bool]]:
    """The host RNG draws of the train-time branch, in the reference's order and from the same (default CPU)
    generator: `Bernoulli(1 - p).sample((1,))` for keep (rwkv_wrapper_bidirectional_direction_dropout.py:60-62) and,
    only when the right-to-left branch is not kept and only in the `_both` class, a second draw from the same
    distribution for which single direction runs (..._direction_dropout_both.py:63-64).
    Returns (keep, left_only): left_only is None when no second draw was made."""
    bern = torch.distributions.bernoulli.Bernoulli(1 - p)
    keep = bool(bern.sample((1,)) == 1)
    if keep or not both:
        return keep, None
    return keep, bool(bern.sample((1,)) <= 0.5)


def rwkv_wrapper_dir_dropout_train(query, sd: SD, p: str, head_size: int, do_bfloat16: bool, both: bool, keep: bool,
                                   left_only: Optional[bool], cache=None):
    """Train-time branch of RWKV_TmixWrapper_bidirectional_direction_dropout{,_both}.forward
    (rwkv_wrapper_bidirectional_direction_dropout.py:59-69, ..._both.py:55-71) for given draws (draw_direction_dropout):
    keep -> (left + flipped right) / 2; else left only, or in `_both` the direction the second draw chose."""
    x = query
    fwd = lambda: rwkv_wrapper(x, sd, p + "rwkv_wrapper_forward.", head_size, do_bfloat16, cache)[0]

    def bwd():
        xf = torch.flip(x, [1])
        return torch.flip(rwkv_wrapper(xf, sd, p + "rwkv_wrapper_backward.", head_size, do_bfloat16, cache)[0], [1])
    if keep:
        out = (fwd() + bwd()) / 2
    elif not both or left_only:
        out = fwd()
    else:
        out = bwd()
    if cache is None:
        cache = torch.zeros((0, 0, 0, 0))
    return out, cache


def rwkv_wrapper_dir_dropout_eval(query, sd: SD, p: str, head_size: int, do_bfloat16: bool, layer_id: int,
                                  cache=None, env: Optional[Dict[str, str]] = None):
    """Eval branches of RWKV_TmixWrapper_bidirectional_direction_dropout{,_both}.forward,
    rwkv_wrapper_bidirectional_direction_dropout.py:25-33 (env read at construction) and :70-92.
    The inner wrappers keep their own do_bfloat16 (cast in, cast back to the query dtype), so the
    average is taken in the query dtype."""
    env = os.environ if env is None else env
    alt = env.get("RWKV_ALT_DECODING", "0") == "1"
    bi_active = True
    if env.get("RWKV_BIDIRECTIONAL_LAYERS"):
        bi_active = layer_id in [int(s) for s in env["RWKV_BIDIRECTIONAL_LAYERS"].split(",")]
    x = query
    fwd = lambda: rwkv_wrapper(x, sd, p + "rwkv_wrapper_forward.", head_size, do_bfloat16, cache)[0]

    def bwd():
        xf = torch.flip(x, [1])
        return torch.flip(rwkv_wrapper(xf, sd, p + "rwkv_wrapper_backward.", head_size, do_bfloat16, cache)[0], [1])

    if bi_active:
        out = (fwd() + bwd()) / 2
    elif alt and layer_id % 2 == 0:
        out = fwd()
    elif alt and layer_id % 2 == 1:
        out = bwd()
    else:
        out = fwd()
    if cache is None:
        cache = torch.zeros((0, 0, 0, 0))
    return out, cache


ATTENTION = {
    # registry keys of wenet/utils/class_utils.py:77-89
    "rwkv_tmix60": "uni",
    "rwkv_tmix60_bidirectional": "bi",
    "rwkv_tmix60_bidirectional2": "bi",
    "rwkv_tmix60_dir_layer_drop": "dld",
    "rwkv_tmix60_dir_layer_drop_both": "dld",
}


def self_attn(x, sd: SD, p: str, kind: str, head_size: int, do_bfloat16: bool, layer_id: int, cache=None, env=None,
              out_as_query: bool = False):
    k = ATTENTION[kind]
    if k == "uni":
        return rwkv_wrapper(x, sd, p, head_size, do_bfloat16, cache)
    if k == "bi":
        return rwkv_wrapper_bidirectional(x, sd, p, head_size, do_bfloat16, cache, out_as_query=out_as_query)
    return rwkv_wrapper_dir_dropout_eval(x, sd, p, head_size, do_bfloat16, layer_id, cache, env)


# ----------------------------------------------------------------------------
# Conformer layer pieces
# ----------------------------------------------------------------------------
def layer_norm(x, sd: SD, p: str):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def positionwise_ff(x, sd: SD, p: str):
    """PositionwiseFeedForward.forward, wenet/transformer/positionwise_feed_forward.py:47-55, activation
    'swish' = torch.nn.SiLU (class_utils.py:44-50)."""
    return F.linear(F.silu(F.linear(x, sd[p + "w_1.weight"], sd[p + "w_1.bias"])), sd[p + "w_2.weight"], sd[p + "w_2.bias"])


def conv_module(x, mask_pad, sd: SD, p: str, kernel_size: int, causal: bool = False, cache=None):
    """ConvolutionModule.forward, wenet/transformer/convolution.py:89-144; cnn_module_norm = layer_norm, activation SiLU.
    x: (B, T, C); mask_pad: (B, 1, T) bool or (0,0,0).

    Non-causal (`causal: false`, every shipped conf/rwkv YAML): lorder = 0 (:56-60), the depthwise convolution pads
    (k - 1) / 2 frames either side, nothing is cached -- returns y.
    Causal (:49-55,113-126): lorder = k - 1, the depthwise convolution has no padding of its own; the INPUT of the module
    (after the pad mask, before pointwise_conv1) is extended on the left by `lorder` zero frames when `cache` is None or
    has no frames (:114-115), otherwise by `cache` (B, C, cache_t) (:117-119); the new cache is the last `lorder` frames
    of that extended input (:121) -- returns (y, new_cache (B, C, lorder))."""
    x = x.transpose(1, 2)
    if mask_pad.size(2) > 0:
        x = x.masked_fill(~mask_pad, 0.0)
    new_cache = None
    if causal:
        lorder = kernel_size - 1
        if cache is None or cache.size(2) == 0:
            x = F.pad(x, (lorder, 0), "constant", 0.0)
        else:
            assert cache.size(0) == x.size(0) and cache.size(1) == x.size(1)
            x = torch.cat((cache, x), dim=2)
        assert x.size(2) > lorder
        new_cache = x[:, :, -lorder:]
    x = F.conv1d(x, sd[p + "pointwise_conv1.weight"], sd[p + "pointwise_conv1.bias"])
    x = F.glu(x, dim=1)
    C = x.shape[1]
    x = F.conv1d(x, sd[p + "depthwise_conv.weight"], sd[p + "depthwise_conv.bias"],
                 padding=0 if causal else (kernel_size - 1) // 2, groups=C)
    x = x.transpose(1, 2)
    x = F.silu(F.layer_norm(x, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5))
    x = x.transpose(1, 2)
    x = F.conv1d(x, sd[p + "pointwise_conv2.weight"], sd[p + "pointwise_conv2.bias"])
    if mask_pad.size(2) > 0:
        x = x.masked_fill(~mask_pad, 0.0)
    return (x.transpose(1, 2), new_cache) if causal else x.transpose(1, 2)


def conformer_layer(x, mask_pad, sd: SD, p: str, conf: dict, layer_id: int, env=None, cnn_cache=None,
                    return_cnn_cache: bool = False):
    """ConformerEncoderLayer.forward, wenet/transformer/encoder_layer.py:165-261: macaron FFN (ff_scale 0.5,
    :149-151), conv module, eval-mode dropout; normalize_before (the paper's configs) puts each LayerNorm in front of its
    branch, otherwise behind the residual add (:203-208, :212-213/:233-234, :241-247, :251-256).  With `causal: true`
    (encoder.py:572-573 -> convolution.py:49-55) the conv module takes `cnn_cache` (B, C, cache_t) and hands back its new
    left context (encoder_layer.py:243), returned as the second value when `return_cnn_cache`."""
    head_size = conf["output_size"] // conf["attention_heads"]
    pre = bool(conf.get("normalize_before", True))
    causal = bool(conf.get("causal", False))
    side = {}

    def slot(h):
        return self_attn(h, sd, p + "self_attn.", conf["selfattention_layer_type"], head_size,
                         conf.get("rwkv_do_bfloat16", True), layer_id, env=env,
                         out_as_query=bool(conf.get("oracle_slot_out_as_query", False)))[0]

    def conv(h):
        y = conv_module(h, mask_pad, sd, p + "conv_module.", conf["cnn_module_kernel"], causal, cnn_cache)
        if causal:
            y, side["cnn"] = y
        return y
    for norm, scale, fn in (("norm_ff_macaron.", 0.5, lambda h: positionwise_ff(h, sd, p + "feed_forward_macaron.")),
                            ("norm_mha.", None, slot),
                            ("norm_conv.", None, conv),
                            ("norm_ff.", 0.5, lambda h: positionwise_ff(h, sd, p + "feed_forward."))):
        if not pre and norm == "norm_conv." and mask_pad.size(2) > 0:
            # post-norm hands the residual stream itself to the conv module, whose masked_fill_ is IN PLACE on a view of
            # its input (convolution.py:105-109): the residual loses its padded frames as well
            x = x.masked_fill(~mask_pad.transpose(1, 2), 0.0)
        r = x
        y = fn(layer_norm(x, sd, p + norm) if pre else x)
        x = r + (y if scale is None else scale * y)
        if not pre:
            x = layer_norm(x, sd, p + norm)
    x = layer_norm(x, sd, p + "norm_final.")
    if return_cnn_cache:
        return x, side.get("cnn", torch.zeros((0, 0, 0), dtype=x.dtype))
    return x


# ----------------------------------------------------------------------------
# Encoder front: cmvn, subsampling, masks
# ----------------------------------------------------------------------------
def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """wenet/utils/mask.py:200-226."""
    max_len = max_len if max_len > 0 else int(lengths.max())
    return torch.arange(max_len, dtype=torch.int64)[None, :] >= lengths.to(torch.int64)[:, None]


def global_cmvn(x, sd: SD, p: str = "global_cmvn."):
    """GlobalCMVN.forward, wenet/transformer/cmvn.py:36-47."""
    if p + "mean" not in sd:
        return x
    return (x - sd[p + "mean"]) * sd[p + "istd"]


def abs_position_rows(offset: int, size: int, d_model: int):
    """Rows [offset, offset + size) of PositionalEncoding's sin/cos table, wenet/transformer/embedding.py:39-56."""
    pos = torch.arange(offset, offset + size, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(size, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def conv2d_subsampling4(x, x_mask, sd: SD, p: str = "embed.", pos_enc: str = "rel_pos", offset: int = 0):
    """Conv2dSubsampling4.forward, wenet/transformer/subsampling.py:201-226, followed by
    RelPositionalEncoding.forward, embedding.py:133-147 (x * sqrt(d_model); pos_emb is returned by the
    reference but never read by the RWKV slot, so it is not produced here) or, for pos_enc 'abs_pos',
    PositionalEncoding.forward, embedding.py:58-77 (x * sqrt(d_model) + pe[offset : offset + T'])."""
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd[p + "conv.2.weight"], sd[p + "conv.2.bias"], stride=2))
    b, c, t, f = x.shape
    x = F.linear(x.transpose(1, 2).contiguous().view(b, t, c * f), sd[p + "out.0.weight"], sd[p + "out.0.bias"])
    x = x * math.sqrt(x.shape[-1])
    if pos_enc in ("abs_pos", "embed"):
        x = x + abs_position_rows(offset, x.shape[1], x.shape[-1]).to(x.dtype)
    return x, x_mask[:, :, 2::2][:, :, 2::2]


def encoder_forward(xs, xs_lens, sd: SD, conf: dict, env=None, return_layers: bool = False):
    """BaseEncoder.forward, wenet/transformer/encoder.py:117-149 for ConformerEncoder (encoder.py:453-602)
    with num_langs = 0, no dynamic/static chunking (add_optional_chunk_mask is the identity for the paper's
    configs, utils/mask.py:126-197).  sd keys are relative to the encoder ('embed.', 'encoders.N.', ...)."""
    T = xs.size(1)
    masks = ~make_pad_mask(xs_lens, T).unsqueeze(1)
    xs = global_cmvn(xs, sd)
    xs, masks = conv2d_subsampling4(xs, masks, sd, pos_enc=conf.get("pos_enc_layer_type", "rel_pos"))
    layers = []
    for i in range(conf["num_blocks"]):
        xs = conformer_layer(xs, masks, sd, f"encoders.{i}.", conf, i, env)
        layers.append(xs)
    if conf.get("normalize_before", True):       # encoder.py:145-146
        xs = layer_norm(xs, sd, "after_norm.")
    return (xs, masks, layers) if return_layers else (xs, masks)


def encoder_forward_chunk(xs, sd: SD, conf: dict, env=None, offset: int = 0, cnn_cache=None):
    """BaseEncoder.forward_chunk, encoder.py:231-339, as it behaves with an RWKV slot: B == 1, all-ones masks, the slot
    hands `att_cache` back untouched so it comes back (0,0,0,0) (rwkv_wrapper.py:81): the
    recurrence restarts from S = 0 and the token shift from a zero frame in every chunk.
    Non-causal conv module: cnn_cache comes back (num_blocks,0,0,0) -- an independent full-context pass over the chunk.
    Causal (`causal: true`): layer i takes cnn_cache[i] (B=1, C, cache_t) when cnn_cache has layers, else the empty
    cache (encoder.py:311-318), and the new left contexts come back stacked (num_blocks, 1, C, lorder) (:322,335)."""
    assert xs.size(0) == 1
    masks = torch.ones(1, 1, xs.size(1), dtype=torch.bool)
    xs = global_cmvn(xs, sd)
    xs, _ = conv2d_subsampling4(xs, masks, sd, pos_enc=conf.get("pos_enc_layer_type", "rel_pos"), offset=offset)
    empty_mask = torch.ones((0, 0, 0), dtype=torch.bool)
    new_cnn = []
    for i in range(conf["num_blocks"]):
        cc = cnn_cache[i] if (cnn_cache is not None and cnn_cache.size(0) > 0) else None
        xs, nc = conformer_layer(xs, empty_mask, sd, f"encoders.{i}.", conf, i, env, cnn_cache=cc, return_cnn_cache=True)
        new_cnn.append(nc.unsqueeze(0))
    if conf.get("normalize_before", True):       # encoder.py:328-329
        xs = layer_norm(xs, sd, "after_norm.")
    return xs, torch.zeros((0, 0, 0, 0)), torch.cat(new_cnn, dim=0)


def encoder_forward_chunk_by_chunk(xs, decoding_chunk_size: int, sd: SD, conf: dict, env=None):
    """BaseEncoder.forward_chunk_by_chunk, encoder.py:341-402: overlapping input windows of
    (chunk - 1) * subsampling + right_context + 1 frames every subsampling * chunk frames (subsampling_rate 4,
    right_context 6: subsampling.py:197-199), each through forward_chunk, outputs concatenated, all-ones mask.
    With the recurrent slot and the non-causal conv module the caches stay empty, so every window is an independent
    full-context pass; a causal conv module's cnn_cache is threaded from window to window
    (encoder.py:392-397)."""
    assert decoding_chunk_size > 0 and xs.size(0) == 1
    subsampling, context = 4, 6 + 1
    stride = subsampling * decoding_chunk_size
    window = (decoding_chunk_size - 1) * subsampling + context
    outs = []
    offset = 0          # encoder.py:377,399: the running count of output frames is the next window's positional offset
    cnn_cache = None
    for cur in range(0, xs.size(1) - context + 1, stride):
        y, _, cnn_cache = encoder_forward_chunk(xs[:, cur:min(cur + window, xs.size(1))], sd, conf, env, offset, cnn_cache)
        outs.append(y)
        offset += y.size(1)
    ys = torch.cat(outs, 1)
    return ys, torch.ones((1, 1, ys.size(1)), dtype=torch.bool)


# ----------------------------------------------------------------------------
# CTC head + greedy search
# ----------------------------------------------------------------------------
def ctc_log_softmax(enc_out, sd: SD, p: str = "ctc."):
    """CTC.log_softmax, wenet/transformer/ctc.py:106-114 (via ASRModel.ctc_logprobs, asr_model.py:324-335,
    blank_penalty = 0)."""
    return F.log_softmax(F.linear(enc_out, sd[p + "ctc_lo.weight"], sd[p + "ctc_lo.bias"]), dim=2)


def remove_duplicates_and_blank(hyp: List[int], blank_id: int = 0) -> List[int]:
    """wenet/utils/ctc_utils.py:22-32."""
    out: List[int] = []
    cur = 0
    while cur < len(hyp):
        if hyp[cur] != blank_id:
            out.append(hyp[cur])
        prev = cur
        while cur < len(hyp) and hyp[cur] == hyp[prev]:
            cur += 1
    return out


def ctc_greedy_search(ctc_probs, ctc_lens, blank_id: int = 0) -> List[List[int]]:
    """wenet/transformer/search.py:106-121: topk(1), padded frames -> blank, collapse."""
    B, maxlen = ctc_probs.shape[:2]
    idx = ctc_probs.topk(1, dim=2)[1].view(B, maxlen)
    idx = idx.masked_fill(make_pad_mask(ctc_lens, maxlen), blank_id)
    return [remove_duplicates_and_blank(h.tolist(), blank_id) for h in idx]


def encoder_param_count(sd: SD) -> int:
    return sum(v.numel() for v in sd.values())
