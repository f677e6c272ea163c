"""GPU-resident CTC greedy search (include/pafc_search.h) against the oracle's restatement of the reference steps
(topk(1) -> padded frames = blank -> remove_duplicates_and_blank).  Bit-exact bar: token ids."""
import pytest
import torch

from oracle import encoder_oracle as EO

pytestmark = pytest.mark.gpu


def _scores(B, T, V, dtype, seed, peaky=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=g)
    if peaky:   # long runs of the same id and many blanks, like real CTC posteriors
        ids = torch.randint(0, V, (B, (T + 6) // 7), generator=g).repeat_interleave(7, dim=1)[:, :T]
        ids[torch.rand(B, T, generator=g) < 0.5] = 0
        x.scatter_(2, ids[..., None], 6.0)
    return x.log_softmax(-1).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,V", [(1, 1, 2), (3, 17, 50), (8, 499, 5000), (2, 300, 4999), (5, 64, 7)])
def test_greedy_kernel_matches_oracle(hip, dtype, B, T, V):
    from paper_accurate_fast_cheap_amd.hip_ops import ctc_greedy
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_greedy_search
    x = _scores(B, T, V, dtype, seed=B * 1000 + T)
    lens = torch.randint(1, T + 1, (B,), generator=torch.Generator().manual_seed(T))
    lens[0] = T
    want = EO.ctc_greedy_search(x.float(), lens, 0)
    xd = x.cuda()
    got = [r.tokens for r in ctc_greedy_search(xd, lens.cuda(), 0)]
    assert got == want
    # raw kernel outputs: counts, first-frame indices, untouched tail
    tokens, ntok, frames = ctc_greedy(xd, lens.cuda(), 0, want_frames=True)
    best = xd.float().argmax(-1).cpu()
    for b in range(B):
        n = int(ntok[b])
        assert tokens[b, :n].tolist() == want[b]
        fr = frames[b, :n].tolist()
        assert all(int(best[b, f]) == tok for f, tok in zip(fr, want[b]))
        assert fr == sorted(fr) and all(f < int(lens[b]) for f in fr)


def test_greedy_ties_resolve_to_lowest_index_and_all_blank(hip):
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_greedy_search
    x = torch.full((2, 6, 300), -5.0)
    x[0, :, 0] = -1.0                          # utterance 0: blank everywhere -> no tokens
    x[1, 0, [7, 130, 299]] = -0.5              # three-way tie -> 7
    x[1, 1, [130, 299]] = -0.5                 # tie -> 130
    x[1, 2, 130] = -0.5                        # repeat of 130: collapsed
    x[1, 3, 0] = -0.1                          # blank separates
    x[1, 4, 130] = -0.5                        # 130 again: emitted again
    x[1, 5, 5] = -0.2                          # beyond lens -> ignored
    lens = torch.tensor([6, 5])
    for dt in (torch.float32, torch.bfloat16):
        got = [r.tokens for r in ctc_greedy_search(x.to(dt).cuda(), lens.cuda(), 0)]
        # ties: first maximal index (torch.argmax's documented rule; topk(1) leaves ties unspecified)
        assert got == [[], [7, 130, 130]]
    allneg = torch.full((1, 3, 40), float("-inf"))   # degenerate rows resolve to index 0 = blank -> nothing emitted
    assert [r.tokens for r in ctc_greedy_search(allneg.cuda(), torch.tensor([3]).cuda(), 0)] == [[]]


def test_greedy_long_form_sequence(hip):
    """c3 size: one 30-minute sequence (T' = 44 998, V = 5000, bf16): tile-crossing runs and offsets."""
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_greedy_search
    T, V = 44998, 5000
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, ((T + 4) // 5,), generator=g).repeat_interleave(5)[:T]
    ids[torch.rand(T, generator=g) < 0.4] = 0
    x = torch.full((1, T, V), -9.0, dtype=torch.bfloat16, device="cuda")
    x[0, torch.arange(T), ids.cuda()] = -0.01
    got = ctc_greedy_search(x, torch.tensor([T], device="cuda"), 0)[0].tokens
    assert got == EO.remove_duplicates_and_blank(ids.tolist(), 0)


def test_greedy_bad_arguments(hip):
    from paper_accurate_fast_cheap_amd._lib import PafcError
    from paper_accurate_fast_cheap_amd.hip_ops import ctc_greedy
    with pytest.raises(PafcError):
        ctc_greedy(torch.zeros(2, 3, 4), None)            # host tensor
    with pytest.raises(PafcError):
        ctc_greedy(torch.zeros(2, 3, 4, device="cuda"), None, blank_id=9)
    with pytest.raises(PafcError):
        ctc_greedy(torch.zeros(2, 3, 4, device="cuda", dtype=torch.float16), None)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,V", [(1, 1), (7, 50), (300, 5000), (33, 4999), (5, 9000), (3, 70000)])
def test_log_softmax_rows(hip, dtype, rows, V):
    """One-pass log-softmax vs the fp32 definition: register path, unaligned rows, rows too long for registers."""
    from paper_accurate_fast_cheap_amd.hip_ops import log_softmax_rows
    x = (torch.randn(rows, V, generator=torch.Generator().manual_seed(V)) * 4).to(dtype)
    x[0, 0] = 30.0                                           # a dominant entry: log-prob ~ 0, the others ~ -30
    want = x.float().log_softmax(-1)
    got = log_softmax_rows(x.cuda())
    tol = dict(rtol=2 ** -7, atol=2 ** -6) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got.cpu().float(), want, **tol)
    assert got.dtype == dtype
    if dtype == torch.float32:
        assert torch.allclose(got.exp().sum(-1).cpu(), torch.ones(rows), atol=1e-4)
    buf = x.cuda().clone()
    assert log_softmax_rows(buf, inplace=True).data_ptr() == buf.data_ptr()
    assert torch.equal(buf, got)
    if dtype == torch.float32:   # same argmax as the input (greedy search may run on either)
        assert torch.equal(got.argmax(-1).cpu(), x.argmax(-1))


def _host_beam(logp, lens, beam):
    """the product's host bookkeeping on CPU tensors (itself pinned to the reference by tests/test_search.py)"""
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_prefix_beam_search
    return ctc_prefix_beam_search(logp.cpu(), lens.cpu(), beam)


@pytest.mark.parametrize("B,T,V,beam", [(1, 1, 5, 3), (4, 37, 50, 10), (3, 120, 5000, 8), (2, 60, 12, 16), (5, 25, 7, 10)])
def test_gpu_prefix_beam_search_matches_host(hip, B, T, V, beam):
    """GPU-resident CTC prefix beam search: n-best token lists identical to the host loop (= the reference's), scores
    equal to float64 round-off; ragged lengths, beams wider than the vocabulary, peaky and flat posteriors."""
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_prefix_beam_search
    for peaky in (True, False):
        logp = _scores(B, T, V, torch.float32, seed=B * 100 + T + int(peaky), peaky=peaky)
        lens = torch.randint(1, T + 1, (B,), generator=torch.Generator().manual_seed(T + 3))
        lens[0] = T
        want = _host_beam(logp, lens, beam)
        got = ctc_prefix_beam_search(logp.cuda(), lens.cuda(), beam)
        for w, g in zip(want, got):
            assert [tuple(n) for n in g.nbest] == [tuple(n) for n in w.nbest]
            assert tuple(g.tokens) == tuple(w.tokens)
            assert g.nbest_scores == pytest.approx(w.nbest_scores, rel=1e-12, abs=1e-9)


def test_gpu_prefix_beam_search_golden_c5(hip):
    """The reference's own output for config c5's CTC search (tests/golden, captured from wenet/transformer/search.py)."""
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_prefix_beam_search
    from tests.conftest import load_golden
    g = load_golden("search_c5")
    res = ctc_prefix_beam_search(g["logp"].cuda(), g["enc_lens"].cuda(), 8)
    for r, want in zip(res, g["ctc_prefix"]):
        assert list(r.tokens) == want["tokens"]
        assert [list(n) for n in r.nbest] == want["nbest"]
        assert r.nbest_scores == pytest.approx(want["nbest_scores"], rel=1e-12, abs=1e-9)


def test_gpu_prefix_beam_search_edge_cases(hip):
    from paper_accurate_fast_cheap_amd._lib import PafcError
    from paper_accurate_fast_cheap_amd.hip_ops import ctc_prefix_beam
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_prefix_beam_search
    logp = torch.log_softmax(torch.randn(2, 6, 7, generator=torch.Generator().manual_seed(3)), -1)
    lens = torch.tensor([6, 0])
    r = ctc_prefix_beam_search(logp.cuda(), lens.cuda(), 3)
    h = _host_beam(logp, lens, 3)
    assert list(r[1].tokens) == [] and r[1].nbest == [()] and r[1].score == 0.0      # empty utterance
    assert [tuple(n) for n in r[0].nbest] == [tuple(n) for n in h[0].nbest] and len(r[0].nbest) == 3
    allblank = torch.full((1, 5, 4), -20.0)
    allblank[..., 0] = 0.0
    assert list(ctc_prefix_beam_search(allblank.cuda(), torch.tensor([5]).cuda(), 2)[0].tokens) == []
    with pytest.raises(PafcError):   # beam beyond the kernel's limit is refused, not truncated
        ctc_prefix_beam(torch.zeros(1, 2, 17, device="cuda"), torch.zeros(1, 2, 17, dtype=torch.int64, device="cuda"), None, 17)


def test_end_to_end_decode_example(hip, capsys):
    """tools/decode_example.py: waveform -> HIP fbank -> encoder -> CTC search on the device -> tokenizer -> WER report,
    for both CTC decode modes (random weights: the chain and its interfaces are what is checked)."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "decode_example.py")
    spec = importlib.util.spec_from_file_location("decode_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for mode in ("ctc_greedy_search", "ctc_prefix_beam_search"):
        res = mod.main(["--mode", mode, "--beam_size", "4"])
        out = capsys.readouterr().out
        assert len(res) == 3 and all(isinstance(t, int) for r in res for t in r.tokens)
        assert out.count("utt") == 3 and "Overall ->" in out and "N=" in out


def test_decode_windows_stitches_long_form(hip):
    """utils/longform.decode_windows: window batches through the encoder + CTC greedy search on the GPU, token lists
    stitched in order with window / token start times; equal to decoding every window by itself."""
    import bench
    from paper_accurate_fast_cheap_amd.utils.longform import decode_windows, feats_batcher
    model, _ = bench.build_model("bf16", torch.device("cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    feats = (torch.randn(1, 2450, 80, device="cuda", generator=g) * 2 + 8).to(torch.bfloat16)
    out = decode_windows(model, feats, 600, 2)
    assert len(out["windows"]) == 5 and out["window_start_ms"] == [0.0, 6000.0, 12000.0, 18000.0, 24000.0]
    assert out["tokens"] == [t for w in out["windows"] for t in w]
    assert len(out["token_start_ms"]) == len(out["tokens"]) and out["token_start_ms"] == sorted(out["token_start_ms"])
    singles = []
    with torch.no_grad():
        for fb, lens in feats_batcher(feats, 600, 2):
            singles += [list(r.tokens) for r in model.decode(["ctc_greedy_search"], fb, lens)["ctc_greedy_search"]]
    assert singles == out["windows"]
    beam = decode_windows(model, feats, 600, 2, mode="ctc_prefix_beam_search", beam_size=4)
    assert len(beam["windows"]) == 5 and beam["token_start_ms"] is None
    # batches in flight on streams (default 3): same token lists and times as one batch at a time; a file with many batches
    # of a recurring shape also goes through the encoder's hipGraph replays (one graph per shape and stream)
    one = decode_windows(model, feats, 600, 2, streams=1)
    assert one["windows"] == out["windows"] and one["token_start_ms"] == out["token_start_ms"]
    long_feats = (torch.randn(1, 600 * 2 * 9 + 250, 80, device="cuda", generator=g) * 2 + 8).to(torch.bfloat16)   # 10 batches
    a = decode_windows(model, long_feats, 600, 2, streams=3)
    b = decode_windows(model, long_feats, 600, 2, streams=1)
    assert len(a["windows"]) == 19 and a["windows"] == b["windows"] and a["token_start_ms"] == b["token_start_ms"]
    assert model.encoder.graph_cache_size == 6                  # (two shapes) x (three streams), kept for the next file ...
    c = decode_windows(model, long_feats, 600, 2, streams=3)    # ... which is replayed from the graphs of the first
    assert c["windows"] == a["windows"]
    assert sum(isinstance(v, tuple) for v in model.encoder._graphs.values()) >= 3
    assert decode_windows(model, feats, 600, 2, streams=2, graph_cache=False)["windows"] == out["windows"]


def test_batches_in_flight_on_two_streams_decode_to_the_same_tokens(hip):
    """bench.py --workload c2 keeps two decode batches in flight, each on a stream of its own, and fetches the greedy
    tokens once per pass (ctc_greedy_search(defer=True)): batch for batch the same token lists as the one-stream,
    fetch-per-batch loop -- the batches, the kernels and therefore the padded-frame semantics are untouched."""
    import bench
    from paper_accurate_fast_cheap_amd.transformer.search import ctc_greedy_search
    dev = torch.device("cuda")
    model, _ = bench.build_model("bf16", dev)
    g = torch.Generator(device="cuda").manual_seed(7)
    batches = []
    for B, T in ((5, 331), (7, 203), (4, 1203), (6, 97), (3, 611)):
        lens = torch.randint(T // 2, T + 1, (B,), generator=torch.Generator().manual_seed(T))
        lens[0] = T
        fb = (torch.randn(B, T, 80, device=dev, generator=g) * 2 + 8).to(torch.bfloat16)
        batches.append((fb, lens.to(dev)))

    def one(fb, lens, defer):
        enc, mask = model._forward_encoder(fb, lens)
        return ctc_greedy_search(model.ctc_logprobs(enc), mask.squeeze(1).sum(1), 0, defer=defer)
    with torch.no_grad():
        want = [[list(r.tokens) for r in one(fb, lens, False)] for fb, lens in batches]
        main = torch.cuda.current_stream(dev)
        side = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for s in side:
            s.wait_stream(main)
        pending = []
        for i, (fb, lens) in enumerate(batches):
            with torch.cuda.stream(side[i % 2]):
                pending.append(one(fb, lens, True))
        for s in side:
            main.wait_stream(s)
        got = [[list(r.tokens) for r in f()] for f in pending]
    assert got == want and any(len(t) > 0 for b in want for t in b)
    # the package's decode loop (what bench.py times): one, two and three batches in flight give the same lists
    from paper_accurate_fast_cheap_amd.utils.longform import greedy_decode_batches
    for n in (1, 2, 3):
        toks, logp = greedy_decode_batches(model, batches, streams=n)
        assert [[list(r.tokens) for r in b] for b in toks] == want and logp.shape[0] == batches[-1][0].shape[0]
    assert greedy_decode_batches(model, batches, streams=2, want_tokens=False)[0] is None
