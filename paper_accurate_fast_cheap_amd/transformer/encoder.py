"""BaseEncoder / ConformerEncoder with the recurrent attention slot, on the MI355X.

Reference: wenet/transformer/encoder.py:38-402 (BaseEncoder: forward, forward_chunk, forward_chunk_by_chunk)
and :453-602 (ConformerEncoder: slot constructor arguments :545-569, layer list :589-602).  Same constructor
keys as conf/rwkv/*.yaml `encoder_conf`, same sub-module names (embed, encoders.N, after_norm, global_cmvn),
same return shapes.  Scope: num_langs == 0 and the recurrent slot keys; the MHA baseline, the LSL variant and
the plain TransformerEncoder are not part of the accelerated path.
"""
import os
from .layer_norm import LayerNorm
from typing import List, Optional, Tuple

import torch

from ..utils.class_utils import (WENET_ACTIVATION_CLASSES, WENET_ATTENTION_CLASSES, WENET_EMB_CLASSES,
                                 WENET_SUBSAMPLE_CLASSES)
from ..utils.mask import add_optional_chunk_mask, make_pad_mask
from .convolution import ConvolutionModule
from .encoder_layer import ConformerEncoderLayer
from .positionwise_feed_forward import PositionwiseFeedForward


def _bump_epoch():
    from .. import hip_ops
    hip_ops.bump_param_epoch()


def _post_load_bump(module, incompatible_keys):
    """load_state_dict post-hook (a module-level function: a lambda would make the module unpicklable)."""
    _bump_epoch()


_CAPTURE_STREAMS = {}


def _capture_stream(device) -> torch.cuda.Stream:
    """The side stream this module's graph captures run on (one per device, as torch.cuda.graph keeps one of its own)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CAPTURE_STREAMS:
        _CAPTURE_STREAMS[idx] = torch.cuda.Stream(device=device)
    return _CAPTURE_STREAMS[idx]


def _abandon_capture(stream: torch.cuda.Stream, device) -> None:
    """After a capture the runtime refused: let the device drain, taking the runtime's echo of the error here (an invalidated
    capture reports itself once more through the next synchronising call) rather than in the caller's next unrelated operation.
    Best effort: on ROCm 7.0's HIP inside torch 2.10 an invalidated capture keeps failing every later call of the process with
    hipErrorStreamCaptureInvalidated -- ending the capture on its stream by hand (hipStreamEndCapture + hipGetLastError through
    ctypes) was tried in round 6 and changes nothing, and is not done here: a second copy of the HIP runtime could get loaded for
    it.  tests/test_encoder_gpu.py::test_refused_capture_is_never_silent_in_a_child_process records which way a runtime
    behaves; what cannot be cleared surfaces at the caller's next call, naming the capture."""
    for _ in range(2):
        try:
            torch.cuda.synchronize(device)
            break
        except RuntimeError as again:
            if not _capture_refused(again):
                raise


def _capture_refused(e: BaseException) -> bool:
    """Is this the runtime refusing an operation under stream capture (hipErrorStreamCapture* -- a synchronising call, an
    allocation the graph pool cannot serve, a capture-unsafe library call), as opposed to an error of the work itself?"""
    from .._lib import PafcError
    if isinstance(e, PafcError):
        return False
    msg = str(e).lower()
    return "captur" in msg


class BaseEncoder(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 attention_dropout_rate: float = 0.0, input_layer: str = "conv2d", pos_enc_layer_type: str = "abs_pos",
                 normalize_before: bool = True, static_chunk_size: int = 0, use_dynamic_chunk: bool = False,
                 global_cmvn: torch.nn.Module = None, use_dynamic_left_chunk: bool = False,
                 gradient_checkpointing: bool = False):
        super().__init__()
        self._output_size = output_size
        self.global_cmvn = global_cmvn
        self.embed = WENET_SUBSAMPLE_CLASSES[input_layer](
            input_size, output_size, dropout_rate,
            WENET_EMB_CLASSES[pos_enc_layer_type](output_size, positional_dropout_rate))
        self.normalize_before = normalize_before
        self.after_norm = LayerNorm(output_size, eps=1e-5)
        self.static_chunk_size = static_chunk_size
        self.use_dynamic_chunk = use_dynamic_chunk
        self.use_dynamic_left_chunk = use_dynamic_left_chunk
        self.gradient_checkpointing = gradient_checkpointing
        # inference executor (transformer/fused.py); PAFC_DISABLE_FUSED=1 keeps the op-by-op module path
        self.fused_inference = os.environ.get("PAFC_DISABLE_FUSED", "0") != "1"
        self._fused_plan = None
        self._carry_last_fused = False
        # opt-in hipGraph cache for inference batches of a recurring (B, T) shape (windowed long-form decoding runs dozens
        # of identical batches whose ~300 short kernels are launch-bound): the N most recent shapes keep a captured
        # graph of the whole forward; 0 = off.  Each graph pins its activations, hence opt-in and bounded.
        self.graph_cache_size = 0
        self._graphs = {}
        self._graphs_token = None              # the parameter state the cached graphs were captured on (_weights_token)
        self._wt_epoch, self._wt_tensors = None, []
        # a checkpoint load rewrites parameters in place (Tensor._version moves) and may be followed by anything: new epoch
        self.register_load_state_dict_post_hook(_post_load_bump)

    # runtime state that is rebuilt on demand and must not travel with copy.deepcopy / pickle / torch.save of the module
    # (captured hipGraphs, the fused executor's plans with their events and derived weight copies)
    _TRANSIENT = dict(_fused_plan=None, _graphs=None, _graphs_token=None, _wt_epoch=None, _wt_tensors=None, _carry_last_fused=False)

    def __getstate__(self):
        st = self.__dict__.copy()
        for k, v in self._TRANSIENT.items():
            if k in st:
                st[k] = {} if k == "_graphs" else [] if k == "_wt_tensors" else v
        return st

    def _fused(self, xs: torch.Tensor):
        """The fused executor's plan when this call may use it (no autograd, GPU, eligible layers), else None."""
        if not self.fused_inference or torch.is_grad_enabled() or not xs.is_cuda or self.training:
            return None
        from . import fused
        if self._fused_plan is None:
            if not (self.normalize_before and all(fused.eligible(l) for l in self.encoders)):
                self._fused_plan = False           # remembered: these layers take the module path in forward()
                return None
            self._fused_plan = fused.EncoderPlan(self.encoders)
        return self._fused_plan or None

    def output_size(self) -> int:
        return self._output_size

    def multi_stream_safe(self) -> bool:
        """May several forward passes of this encoder be in flight on different HIP streams?  Only when every GEMM of the pass is
        one of this package's kernels: the framework's library GEMM (F.linear, torch.bmm: what the op-by-op module path calls)
        stalls for good when two streams issue it concurrently (DESIGN.md section 4 "the c2 stall").  True for the fused
        executor over eligible layers (RWKV slots, or the Mamba-2 block on its fused kernels); the schedulers of utils.longform
        keep everything on ONE stream otherwise."""
        if not (self.fused_inference and self.normalize_before and not self.training):
            return False
        from . import fused
        from .mamba2 import MambaAttWrapper
        for layer in self.encoders:
            if not fused.eligible(layer):
                return False
            slot = layer.self_attn
            if isinstance(slot, MambaAttWrapper):
                blocks = [slot.mamba] if not hasattr(slot.mamba, "mamba_forward") else [slot.mamba.mamba_forward, slot.mamba.mamba_backward]
                if not all(b.fused_inference and b.d_state == 128 and b.d_inner <= 1024 for b in blocks):
                    return False
        return True

    def _apply(self, fn, *args, **kwargs):
        """model.to() / .cuda() / .half(): the parameters move to new storage -- every derived copy and captured graph is stale."""
        _bump_epoch()
        return super()._apply(fn, *args, **kwargs)

    def _weights_token(self):
        """Cheap identity of the parameter state a captured graph depends on: the process-wide parameter epoch (train_step,
        train() / eval(), .to(), load_state_dict) and the sum of the in-place versions of every parameter and buffer (an
        optimizer step that is not fused, `p.add_()`, `copy_()`).  The tensor list is looked up once per epoch."""
        from .. import hip_ops
        ep = hip_ops.param_epoch()
        if self._wt_epoch != ep:
            self._wt_tensors = list(self.parameters()) + list(self.buffers())
            self._wt_epoch = ep
        v = 0
        for t in self._wt_tensors:
            v += t._version
        return ep, v

    def train(self, mode: bool = True):
        """train() / eval(): the inference plans' derived weight copies are keyed on Tensor._version, which fused optimizers
        do not touch -- switching mode invalidates them (hip_ops.bump_param_epoch), so a CV pass after training steps
        rebuilds them from the current parameters."""
        from .. import hip_ops
        hip_ops.bump_param_epoch()
        return super().train(mode)

    def forward(self, xs: torch.Tensor, xs_lens: torch.Tensor, decoding_chunk_size: int = 0,
                num_decoding_left_chunks: int = -1, cat_embs: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(B, T, F) padded features + (B,) lengths -> (B, T', C), (B, 1, T') bool mask.  encoder.py:117-149."""
        if (self.graph_cache_size > 0 and xs.is_cuda and not torch.is_grad_enabled() and not self.training
                and cat_embs is None and decoding_chunk_size <= 0):
            out = self._forward_graphed(xs, xs_lens)
            if out is not None:
                return out
        xs, masks, _ = self.forward_return_layers(xs, xs_lens, decoding_chunk_size, num_decoding_left_chunks, cat_embs)
        return xs, masks

    def _forward_graphed(self, xs: torch.Tensor, xs_lens: torch.Tensor):
        """Replay the forward of this (B, T, dtype) shape from a captured hipGraph; the first sighting of a shape runs
        eagerly (returns None), the second one captures.  Outputs are copies: the graph's own buffers are reused."""
        # keyed by the issuing stream as well: a graph replays into its own buffers, so batches in flight on two streams
        # (two decode batches overlapping each other's launch-bound stretches) need a graph each
        # A captured graph holds the ADDRESSES of the plans' derived weight tensors (stacked / folded / split copies) and never
        # calls plan.refresh(): after any parameter change the graphs of the old state are dropped (their derived tensors have been
        # freed or are about to be), and the shape is captured again on its second sighting.
        token = self._weights_token()
        if token != self._graphs_token:
            self._graphs.clear()
            self._graphs_token = token
        # Long batches whose rows are all full length run the schedule without padding masks (fused.encoder_layers_forward), a
        # decision that takes a host read of the lengths -- not possible inside a capture.  It is taken HERE, before the capture,
        # and is part of the key: a replay runs the schedule the eager pass of the same batch would run (equal-length windows: the
        # unmasked one; the file's last, padded batch: another graph).
        full = None
        if self._long_batch(xs):
            # in SUBSAMPLED frames, as the eager pass decides it (fused.encoder_layers_forward compares the subsampled mask's sums
            # with T'): an input 1-3 frames short of T has as many output frames as a full one
            sub_len = getattr(self.embed, "subsampled_length", lambda n: n)
            full = bool(sub_len(int(xs_lens.min())) == sub_len(xs.shape[1]))
        key = (tuple(xs.shape), xs.dtype, xs_lens.dtype, torch.cuda.current_stream(xs.device).cuda_stream, full)
        ent = self._graphs.get(key)
        if ent is None:
            self._graphs[key] = "seen"
            self._trim_graphs()
            return None
        if ent == "seen":
            try:
                sx, sl = xs.clone(), xs_lens.clone()
                graph = torch.cuda.CUDAGraph()
                cap_stream = _capture_stream(xs.device)        # ours, so that a refused capture can be ended by hand (below)
                with torch.cuda.graph(graph, stream=cap_stream):
                    oy, om, _ = self.forward_return_layers(sx, sl, all_full=full)
                ent = self._graphs[key] = (graph, sx, sl, oy, om)
                self._graphs[key] = self._graphs.pop(key)     # newest last, then drop the oldest graphs beyond the bound
                self._trim_graphs()
            except RuntimeError as e:              # a REFUSED capture (an operation the capture mode does not permit): this shape
                if not _capture_refused(e):        # stays eager; anything else -- a failing launch, a PafcError -- surfaces
                    raise
                _abandon_capture(cap_stream, xs.device)
                self._graphs[key] = "eager"
                return None
        if ent == "eager":
            return None
        graph, sx, sl, oy, om = ent
        sx.copy_(xs)
        sl.copy_(xs_lens)
        graph.replay()
        self._graphs[key] = self._graphs.pop(key)  # most recently used last
        return oy.clone(), om.clone()

    def _long_batch(self, xs: torch.Tensor) -> bool:
        """Is this input long enough for the unmasked schedule to be a possibility (rows after subsampling against the schedule's
        threshold)?  Cheap and conservative: the exact eligibility is re-checked where the schedule is chosen."""
        from . import fused
        sub = getattr(self.embed, "subsampling_rate", 1)
        return xs.size(0) * (xs.size(1) // max(1, sub)) >= fused._LN_FOLD_MIN_ROWS

    def _trim_graphs(self):
        live = [k for k, v in self._graphs.items() if isinstance(v, tuple)]
        while len(live) > self.graph_cache_size:
            self._graphs.pop(live.pop(0))
        if len(self._graphs) > 4096:               # bookkeeping of shapes seen once stays bounded too
            for k in [k for k, v in self._graphs.items() if not isinstance(v, tuple)][:2048]:
                self._graphs.pop(k)

    def forward_return_layers(self, xs, xs_lens, decoding_chunk_size: int = 0, num_decoding_left_chunks: int = -1,
                              cat_embs=None, want_layers: bool = False, all_full: Optional[bool] = None):
        """all_full: the caller has read the lengths and knows whether every row is full length after subsampling (the hipGraph
        cache, before it captures); None: the fused executor finds out itself."""
        T = xs.size(1)
        masks = ~make_pad_mask(xs_lens, T).unsqueeze(1)
        if self.global_cmvn is not None:
            xs = self.global_cmvn(xs)
        xs, pos_emb, masks = self.embed(xs, masks)
        mask_pad = masks
        chunk_masks = add_optional_chunk_mask(xs, masks, self.use_dynamic_chunk, self.use_dynamic_left_chunk,
                                              decoding_chunk_size, self.static_chunk_size, num_decoding_left_chunks)
        plan = self._fused(xs)
        if plan is not None:
            from . import fused
            xs, layer_outs = fused.encoder_layers_forward(plan, xs, mask_pad, self.after_norm, want_layers,
                                                          all_full=all_full)
            return xs, masks, layer_outs
        layer_outs: List[torch.Tensor] = []
        for layer in self.encoders:
            xs, chunk_masks, _, _ = layer(xs, chunk_masks, pos_emb, mask_pad, cat_embs=cat_embs)
            if want_layers:
                layer_outs.append(xs)
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, masks, layer_outs

    def forward_chunk(self, xs: torch.Tensor, offset: int, required_cache_size: int,
                      att_cache: torch.Tensor = torch.zeros(0, 0, 0, 0),
                      cnn_cache: torch.Tensor = torch.zeros(0, 0, 0, 0),
                      att_mask: torch.Tensor = torch.ones((0, 0, 0), dtype=torch.bool),
                      cat_embs: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """One chunk, reference semantics (encoder.py:231-339): B == 1; with the recurrent slot the wrappers hand
        `cache` back untouched, so r_att_cache is (0,0,0,0) and, conv being non-causal in the paper's configs,
        r_cnn_cache is (num_blocks,0,0,0): every chunk is an independent full-context pass over its own frames.  With
        `causal: true` r_cnn_cache is (num_blocks, 1, C, lorder), to be handed to the next call (the recurrence still
        restarts per chunk, as in the reference).
        State-carrying streaming (what the reference lacks) is forward_chunk_carry()."""
        assert xs.size(0) == 1
        tmp_masks = torch.ones(1, 1, xs.size(1), device=xs.device, dtype=torch.bool)
        if self.global_cmvn is not None:
            xs = self.global_cmvn(xs)
        xs, pos_emb, _ = self.embed(xs, tmp_masks, offset)
        elayers = att_cache.size(0)
        plan = self._fused(xs)
        # a causal conv module threads its left context from call to call (encoder.py:311-337, convolution.py:113-126): the
        # layer-by-layer path below hands the caches on; the fused executor serves the cache-free (non-causal) chunk
        causal = any(l.conv_module is not None and l.conv_module.lorder > 0 for l in self.encoders)
        if plan is not None and elayers == 0 and cnn_cache.size(0) == 0 and not causal:
            from . import fused
            from .. import hip_ops
            with hip_ops.chunk_step():      # a chunk's few rows: the launch-bound regime (csrc/gemm_skinny.hip)
                xs, _ = fused.encoder_layers_forward(plan, xs, att_mask[:0], self.after_norm)
            n = len(self.encoders)
            return (xs, torch.zeros((0, 0, 0, 0), device=xs.device),
                    torch.zeros((n, 0, 0, 0), dtype=xs.dtype, device=xs.device))
        r_att_cache, r_cnn_cache = [], []
        for i, layer in enumerate(self.encoders):
            xs, _, new_att_cache, new_cnn_cache = layer(
                xs, att_mask, pos_emb, cat_embs=cat_embs,
                att_cache=att_cache[i:i + 1] if elayers > 0 else att_cache,
                cnn_cache=cnn_cache[i] if cnn_cache.size(0) > 0 else cnn_cache)
            r_att_cache.append(new_att_cache)
            r_cnn_cache.append(new_cnn_cache.unsqueeze(0))
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, torch.cat(r_att_cache, dim=0), torch.cat(r_cnn_cache, dim=0)

    def _carry_step_is_fused(self, xs: torch.Tensor) -> bool:
        """Whether forward_chunk_carry serves THIS call on the fused chunk-step kernels (decided once per call, from the
        call's own input: inference, bf16 streams on the GPU, every layer carry-eligible).  The outcome is kept in
        `_carry_last_fused`: stream_chunks hands the step the in-place "cx" carry, which only that path understands,
        only when the stream's own warm-up windows took it."""
        if not (self.fused_inference and not torch.is_grad_enabled() and not self.training and xs.is_cuda
                and xs.dtype == torch.bfloat16):
            return False
        from . import fused
        return all(fused.carry_eligible(l, xs) for l in self.encoders)

    @torch.no_grad()
    def forward_chunk_carry(self, xs: torch.Tensor, offset: int = 0, state: Optional[list] = None, in_place: bool = False
                            ) -> Tuple[torch.Tensor, list]:
        """Streaming step with recurrent-state carry (BASELINE config c3 "recurrent-state carry, no KV cache").
        xs: (B, time, F) input window -- windows overlap exactly as in forward_chunk_by_chunk (encoder.py:379-391,
        window (chunk-1)*4+7, stride 4*chunk) because the subsampling convolutions keep no cache.  state: list of
        per-layer carries from the previous call (None to start).  Uni-directional slot only; exact (chunked == full
        sequence) when the conv module is causal, otherwise the conv sees zeros past the chunk edge like the
        reference's forward_chunk does.  in_place (fused kernels only): the tensors in `state` are updated where they lie
        and `state` itself is returned -- the form a captured hipGraph step needs (stream_chunks)."""
        masks = torch.ones(xs.size(0), 1, xs.size(1), device=xs.device, dtype=torch.bool)
        if self.global_cmvn is not None:
            xs = self.global_cmvn(xs)
        state = state or [None] * len(self.encoders)
        new_state = []
        self._carry_last_fused = self._carry_step_is_fused(xs)      # (after the CMVN: its output dtype is what the layers see)
        if self._carry_last_fused:                  # bf16 streams, causal conv: fused kernels
            from . import fused
            from .. import hip_ops
            with hip_ops.chunk_step():      # few rows: the launch-bound regime (csrc/gemm_skinny.hip)
                xs, _, _ = self.embed(xs, masks, offset)
                if getattr(self, "_carry_plans", None) is None:
                    self._carry_plans = [fused.LayerPlan(l) for l in self.encoders]
                plans = self._carry_plans
                n = len(plans)
                tail = self.after_norm if self.normalize_before else None
                h = None
                pending = [] if in_place else None     # the carries' small refreshes: one multi-tensor copy at the end
                for i, carry in enumerate(state):
                    plans[i].refresh()
                    nxt = plans[i + 1].layer.norm_ff_macaron if i + 1 < n else tail
                    xs, c, h = fused.layer_forward_carry(plans[i], xs, carry, h0=h, next_norm=nxt,
                                                         in_place=in_place and carry is not None, pending=pending)
                    new_state.append(c)
                if pending:
                    torch._foreach_copy_([d for d, _ in pending], [s_ for _, s_ in pending])
                return (h if tail is not None else xs), new_state
        else:
            xs, _, _ = self.embed(xs, masks, offset)
        for layer, carry in zip(self.encoders, state):
            xs, c = layer.forward_carry(xs, carry)
            new_state.append(c)
        if self.normalize_before:
            xs = self.after_norm(xs)
        return xs, new_state

    @torch.no_grad()
    def stream_chunks(self, xs: torch.Tensor, decoding_chunk_size: int, use_graph: bool = True) -> torch.Tensor:
        """A whole utterance (B, T, F) through forward_chunk_carry, window by window (the windows of
        forward_chunk_by_chunk), returning the concatenated outputs (B, T', D).  A chunk step is ~25 small kernels per
        layer -- launch-bound -- and every full window has the same shape, so on the GPU the step is captured once into
        a hipGraph over fixed input / state / output buffers and replayed per window; the first windows (which also
        let the causal-conv cache reach its final length) and a shorter last window run eagerly."""
        assert decoding_chunk_size > 0
        sub, ctx = self.embed.subsampling_rate, self.embed.right_context + 1
        stride, window = sub * decoding_chunk_size, (decoding_chunk_size - 1) * sub + ctx
        T = xs.size(1)
        starts = list(range(0, T - ctx + 1, stride))
        full = [c for c in starts if c + window <= T]
        outs: List[torch.Tensor] = []
        state: Optional[list] = None

        def eager(c):
            nonlocal state
            y, state = self.forward_chunk_carry(xs[:, c:min(c + window, T)], 0, state)
            outs.append(y)

        lorder = max([getattr(l.conv_module, "lorder", 0) or 0 for l in self.encoders if l.conv_module is not None] + [0])
        warm = max(2, -(-lorder // decoding_chunk_size) + 1)
        done = 0
        if use_graph and xs.is_cuda and len(full) >= warm + 4:
            side = torch.cuda.Stream(device=xs.device)
            side.wait_stream(torch.cuda.current_stream(xs.device))
            with torch.cuda.stream(side):
                for c in full[:warm]:
                    eager(c)
            torch.cuda.current_stream(xs.device).wait_stream(side)
            done = warm
            static_in = xs[:, full[warm]:full[warm] + window].clone()
            static_state = [{k: v.clone() for k, v in st.items()} for st in state]
            if xs.size(0) == 1 and self._carry_last_fused:
                # one stream and THIS stream's steps ran on the fused kernels (the decision forward_chunk_carry just made
                # for the warm-up windows, same shapes and dtypes): the conv module's input buffer lives across steps
                for st in static_state:
                    cnn = st.get("cnn")
                    if cnn is not None and cnn.dim() == 3 and decoding_chunk_size >= cnn.size(2) > 0:
                        cx = cnn.new_zeros(1, cnn.size(2) + decoding_chunk_size, cnn.size(1))
                        cx[:, :cnn.size(2)] = cnn.transpose(1, 2)
                        st["cx"] = cx
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    y_static, new_state = self.forward_chunk_carry(static_in, 0, static_state, in_place=True)
                    for st, nw in zip(static_state, new_state):
                        if nw is not st:            # (the fused step updates its carries where they lie)
                            for k in st:
                                st[k].copy_(nw[k])
            except RuntimeError as e:
                # only "this step cannot be captured here" (an operation the stream capture refuses) falls back to the eager
                # loop -- nothing ran during the failed capture, so `state` is still the state after the warm-up windows;
                # any other error is a real one and is raised
                torch.cuda.synchronize(xs.device)
                if "captur" not in str(e).lower():
                    raise
                graph = None
            if graph is not None:                   # errors from here on are genuine kernel / launch errors: not swallowed
                for c in full[warm:]:
                    static_in.copy_(xs[:, c:c + window])
                    graph.replay()
                    outs.append(y_static.clone())
                    done += 1
                for st in static_state:            # back to the public carries
                    cx = st.pop("cx", None)
                    if cx is not None:
                        st["cnn"] = cx[:, :st["cnn"].size(2)].transpose(1, 2).contiguous()
                state = static_state
        for c in starts[done:]:
            eager(c)
        return torch.cat(outs, 1)

    @torch.no_grad()
    def forward_chunk_lookahead(self, xs: torch.Tensor, state: Optional[list] = None, final: bool = False
                                ) -> Tuple[torch.Tensor, list]:
        """Streaming step with state carry for the uni-directional model AS SHIPPED -- non-causal conv module, k = 31
        (conf/rwkv/giga.rwkv_uni_ds4k31nc_12le.trans-longutts.yaml:14-16) -- where forward_chunk_carry's causal-conv cache
        does not apply: every layer finalises a frame only when the 15 frames behind it have arrived
        (ConformerEncoderLayer.forward_lookahead), so layer l runs 15 l frames behind the input and the encoder emits
        frames 15 x num_blocks (= 180 frames, 7.2 s for 12 layers) behind it; nothing is recomputed, nothing approximated:
        over a whole stream the concatenated outputs equal forward() of the whole utterance (the reference's own
        forward_chunk restarts the recurrence per chunk and zero-pads the conv at every chunk edge, encoder.py:231-339).
        xs: (B, time, F) input window (the windows of forward_chunk_by_chunk; may be None with final=True to drain);
        returns (the output frames finalised by this call (B, v, D) -- v = 0 while the pipeline fills --, state)."""
        state = state or [None] * len(self.encoders)
        if xs is not None:
            masks = torch.ones(xs.size(0), 1, xs.size(1), device=xs.device, dtype=torch.bool)
            if self.global_cmvn is not None:
                xs = self.global_cmvn(xs)
            xs, _, _ = self.embed(xs, masks, 0)
        else:
            ref = state[0]["cu"]
            xs = ref.new_zeros(ref.size(0), 0, ref.size(2))
        new_state = []
        for layer, carry in zip(self.encoders, state):
            xs, c = layer.forward_lookahead(xs, carry, final)
            new_state.append(c)
        if self.normalize_before and xs.size(1) > 0:
            xs = self.after_norm(xs)
        return xs, new_state

    @torch.no_grad()
    def _lookahead_fused_step(self, xs: torch.Tensor, state: list) -> torch.Tensor:
        """One STEADY-STATE look-ahead step on the fused chunk-step kernels (fused.layer_forward_lookahead): every layer takes
        T frames and emits T frames, the carries are the fixed buffers "U" / "X2" / "shift" / "wkv" updated where they lie
        (the form a captured hipGraph needs).  xs: (1, window, F) input window; returns the (1, T, D) frames finalised."""
        from . import fused
        from .. import hip_ops
        masks = torch.ones(1, 1, xs.size(1), device=xs.device, dtype=torch.bool)
        if self.global_cmvn is not None:
            xs = self.global_cmvn(xs)
        with hip_ops.chunk_step():
            xs, _, _ = self.embed(xs, masks, 0)
            if getattr(self, "_carry_plans", None) is None:
                self._carry_plans = [fused.LayerPlan(l) for l in self.encoders]
            plans = self._carry_plans
            n = len(plans)
            tail = self.after_norm if self.normalize_before else None
            h, pending = None, []
            for i, carry in enumerate(state):
                plans[i].refresh()
                nxt = plans[i + 1].layer.norm_ff_macaron if i + 1 < n else tail
                xs, h = fused.layer_forward_lookahead(plans[i], xs, carry, h, nxt, pending)
            torch._foreach_copy_([d for d, _ in pending], [s_ for _, s_ in pending])
        return h if tail is not None else xs

    @torch.no_grad()
    def stream_chunks_lookahead(self, xs: torch.Tensor, decoding_chunk_size: int, use_graph: bool = True) -> torch.Tensor:
        """A whole utterance (B, T, F) through forward_chunk_lookahead window by window, drained at the end: (B, T', D), equal
        to forward() of the utterance.  One bf16 stream on the GPU: once the pipeline is full (every layer holds its 30 + 15
        carried frames: after ceil(15 L / chunk) + 1 windows) each full window is a STEADY-STATE step -- T frames in, T frames
        out of every layer -- which runs on the fused chunk-step kernels over fixed carry buffers and is replayed from a
        captured hipGraph (as stream_chunks does for the causal model); the windows that fill the pipeline, a shorter last
        window and the final drain take the module path."""
        assert decoding_chunk_size > 0
        sub, ctx = self.embed.subsampling_rate, self.embed.right_context + 1
        stride, window = sub * decoding_chunk_size, (decoding_chunk_size - 1) * sub + ctx
        T = xs.size(1)
        starts = list(range(0, T - ctx + 1, stride))
        outs, state = [], None
        i = 0
        n = len(starts)
        halves = [((l.conv_module.kernel_size - 1) // 2 if l.conv_module is not None else 0) for l in self.encoders]
        fill = -(-sum(halves) // decoding_chunk_size) + 1          # windows until every layer emits a full chunk per step
        from . import fused
        fused_ok = (use_graph and xs.is_cuda and self.fused_inference and not self.training and xs.dtype == torch.bfloat16
                    and decoding_chunk_size >= 2 * max(halves + [0]) and all(fused.lookahead_eligible(l, xs) for l in self.encoders))
        steady = [j for j in range(n - 1) if starts[j] + window <= T]          # full windows that are not the last one
        if fused_ok and len(steady) >= fill + 4:
            while i < fill:
                y, state = self.forward_chunk_lookahead(xs[:, starts[i]:starts[i] + window], state)
                outs.append(y)
                i += 1
            ok = all(st["cu"].size(1) == 2 * hf and st["x2"].size(1) == hf for st, hf in zip(state, halves))
            if ok:
                Tc = decoding_chunk_size
                bufs = []
                for st, hf in zip(state, halves):        # the public carries -> fixed buffers with room for a chunk behind them
                    U = st["cu"].new_zeros(1, 2 * hf + Tc, st["cu"].size(2))
                    U[:, :2 * hf] = st["cu"]
                    X2 = st["x2"].new_zeros(1, hf + Tc, st["x2"].size(2))
                    X2[:, :hf] = st["x2"]
                    bufs.append({"U": U, "X2": X2, "shift": st["shift"].clone().contiguous(), "wkv": st["wkv"].clone().contiguous()})
                static_in = xs[:, starts[i]:starts[i] + window].clone()
                side = torch.cuda.Stream(device=xs.device)               # one eager step on a side stream warms every kernel up
                side.wait_stream(torch.cuda.current_stream(xs.device))
                with torch.cuda.stream(side):
                    outs.append(self._lookahead_fused_step(static_in, bufs).clone())
                torch.cuda.current_stream(xs.device).wait_stream(side)
                i += 1
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph):
                        y_static = self._lookahead_fused_step(static_in, bufs)
                except RuntimeError as e:                 # only a refused capture falls back to eager fused steps
                    torch.cuda.synchronize(xs.device)
                    if "captur" not in str(e).lower():
                        raise
                    graph = None
                while i in steady:
                    static_in.copy_(xs[:, starts[i]:starts[i] + window])
                    if graph is not None:
                        graph.replay()
                        outs.append(y_static.clone())
                    else:
                        outs.append(self._lookahead_fused_step(static_in, bufs).clone())
                    i += 1
                state = [{"cu": b["U"][:, :2 * hf].clone(), "x2": b["X2"][:, :hf].clone(), "shift": b["shift"], "wkv": b["wkv"]}
                         for b, hf in zip(bufs, halves)]
        while i < n:
            y, state = self.forward_chunk_lookahead(xs[:, starts[i]:min(starts[i] + window, T)], state, final=(i == n - 1))
            outs.append(y)
            i += 1
        return torch.cat(outs, 1)

    def _windows_independent(self, xs: torch.Tensor):
        """The fused plan when the windows of forward_chunk_by_chunk do not depend on each other: recurrent slot (its
        att_cache stays empty), a non-causal conv module (its cnn_cache stays empty) and a positional encoding that
        leaves xs independent of the window's offset (rel_pos / no_pos; abs_pos adds pe[offset : offset + T] into xs,
        embedding.py:64-77, so its windows keep the window-by-window path with the running offset) -- the paper's configs."""
        from .embedding import NoPositionalEncoding, PositionalEncoding, RelPositionalEncoding
        pos = getattr(self.embed, "pos_enc", None)
        if isinstance(pos, PositionalEncoding) or not isinstance(pos, (RelPositionalEncoding, NoPositionalEncoding)):
            return None
        plan = self._fused(xs)
        if plan is None or any(l.conv_module is not None and l.conv_module.lorder > 0 for l in self.encoders):
            return None
        return plan

    def forward_chunk_by_chunk(self, xs: torch.Tensor, decoding_chunk_size: int, num_decoding_left_chunks: int = -1,
                               cat_embs: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(1, T, F) -> ((1, T', C), all-ones (1, 1, T') mask), the function of encoder.py:341-402: the utterance is cut
        into overlapping windows of (chunk - 1) * subsampling + right_context + 1 input frames, one every
        subsampling * chunk frames, each encoded by forward_chunk with the caches of the previous one, outputs joined.

        With the recurrent slot and the non-causal conv module both caches stay empty, so the windows are independent
        full-context passes: here the equal-length ones go through the layers as ONE batch (B = number of windows; same
        arithmetic per window, no padding, hundreds of launches instead of hundreds per window) and only a shorter
        last window runs alone."""
        assert decoding_chunk_size > 0
        sub, ctx = self.embed.subsampling_rate, self.embed.right_context + 1
        stride, window = sub * decoding_chunk_size, (decoding_chunk_size - 1) * sub + ctx
        T = xs.size(1)
        starts = list(range(0, T - ctx + 1, stride))
        outputs: List[torch.Tensor] = []
        done = 0
        plan = self._windows_independent(xs) if (xs.size(0) == 1 and cat_embs is None) else None
        if plan is not None:
            from . import fused
            full = [c for c in starts if c + window <= T]
            per_batch = max(1, 400_000 // window)                  # bound the activations of one batch of windows
            for b0 in range(0, len(full), per_batch):
                cs = full[b0:b0 + per_batch]
                wb = torch.stack([xs[0, c:c + window] for c in cs])                       # (nw, window, F)
                ones = torch.ones(wb.size(0), 1, window, device=xs.device, dtype=torch.bool)
                if self.global_cmvn is not None:
                    wb = self.global_cmvn(wb)
                wb, _, _ = self.embed(wb, ones, 0)
                y, _ = fused.encoder_layers_forward(plan, wb, ones[:0], self.after_norm)
                outputs.append(y.reshape(1, -1, y.size(-1)))
            done = len(full)
        att_cache = torch.zeros((0, 0, 0, 0), device=xs.device)
        cnn_cache = torch.zeros((0, 0, 0, 0), device=xs.device)
        offset = sum(o.size(1) for o in outputs)
        required_cache_size = decoding_chunk_size * num_decoding_left_chunks
        for cur in starts[done:]:
            y, att_cache, cnn_cache = self.forward_chunk(xs[:, cur:min(cur + window, T), :], offset, required_cache_size,
                                                         att_cache, cnn_cache, cat_embs=cat_embs)
            outputs.append(y)
            offset += y.size(1)
        ys = torch.cat(outputs, 1)
        return ys, torch.ones((1, 1, ys.size(1)), device=ys.device, dtype=torch.bool)


class ConformerEncoder(BaseEncoder):
    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048,
                 num_blocks: int = 6, dropout_rate: float = 0.1, positional_dropout_rate: float = 0.1,
                 attention_dropout_rate: float = 0.0, input_layer: str = "conv2d", pos_enc_layer_type: str = "rel_pos",
                 normalize_before: bool = True, static_chunk_size: int = 0, use_dynamic_chunk: bool = False,
                 global_cmvn: torch.nn.Module = None, use_dynamic_left_chunk: bool = False,
                 positionwise_conv_kernel_size: int = 1, macaron_style: bool = True,
                 selfattention_layer_type: str = "rwkv_tmix60_bidirectional", activation_type: str = "swish",
                 use_cnn_module: bool = True, cnn_module_kernel: int = 15, causal: bool = False,
                 cnn_module_norm: str = "batch_norm", key_bias: bool = True, gradient_checkpointing: bool = False,
                 lora_rank: int = 8, lora_alpha: int = 8, lora_dropout: float = 0.0, lora_list=None,
                 num_langs: int = 0, rwkv_ctx_len: int = 2048, rwkv_do_bfloat16: bool = True,
                 rnn_att_version: str = "", rnn_att_direction: str = "", att_context_size=(500, 500),
                 global_tokens: int = 0, global_tokens_spacing: int = 1, global_attn_separate: bool = False):
        super().__init__(input_size, output_size, attention_heads, linear_units, num_blocks, dropout_rate,
                         positional_dropout_rate, attention_dropout_rate, input_layer, pos_enc_layer_type,
                         normalize_before, static_chunk_size, use_dynamic_chunk, global_cmvn, use_dynamic_left_chunk,
                         gradient_checkpointing)
        if num_langs != 0:
            raise NotImplementedError("language-specific layers (num_langs > 0) are outside the accelerated path")
        if selfattention_layer_type not in WENET_ATTENTION_CLASSES:
            raise NotImplementedError(
                f"selfattention_layer_type={selfattention_layer_type!r}: only the recurrent slot keys "
                f"{sorted(WENET_ATTENTION_CLASSES)} are implemented here (the MHA baseline is out of scope)")
        activation = WENET_ACTIVATION_CLASSES[activation_type]()
        self.num_langs = num_langs
        # encoder.py:545-561: (head_size, dim_att, num_blocks, version, direction, ctx_len, do_bfloat16) + layer_id
        slot_args = (output_size // attention_heads, output_size, num_blocks, rnn_att_version, rnn_att_direction,
                     rwkv_ctx_len, rwkv_do_bfloat16)
        if selfattention_layer_type == "mamba_att":   # encoder.py:563-569
            slot_args = (output_size // attention_heads, output_size, num_blocks, rnn_att_version, rnn_att_direction)
        ff_args = (output_size, linear_units, dropout_rate, activation)
        conv_args = (output_size, cnn_module_kernel, activation, cnn_module_norm, causal)
        self.encoders = torch.nn.ModuleList([
            ConformerEncoderLayer(
                output_size,
                WENET_ATTENTION_CLASSES[selfattention_layer_type](*slot_args, layer_id),
                PositionwiseFeedForward(*ff_args),
                PositionwiseFeedForward(*ff_args) if macaron_style else None,
                ConvolutionModule(*conv_args) if use_cnn_module else None,
                dropout_rate, normalize_before,
            ) for layer_id in range(num_blocks)
        ])
        # fp32 products outside the layers (the subsampling Linear; the CTC head via init_model): split operands on the bf16 matrix
        # cores (~2^-16 relative) only in a model whose slot rounds to bf16 anyway -- a pure-fp32 model (rwkv_do_bfloat16: False,
        # the 1e-3 parity configuration) keeps exact fp32 products everywhere (hip_ops.DISPATCH: split_gemm_min_rows)
        self.fp32_split_operands = bool(selfattention_layer_type != "mamba_att" and rwkv_do_bfloat16)
        self.embed.fp32_split_operands = self.fp32_split_operands
