"""CPU restatements of the decoders on the path (config c5).  TEST INFRASTRUCTURE ONLY.

Deliberately the simplest possible per-utterance, per-candidate Python (what the reference does), so that it can
check the product's batched / one-copy-per-frame implementations:
  ctc_prefix_beam_search            wenet/transformer/search.py:124-248 (tokens, scores, n-best; no time stamps)
  rnnt_prefix_beam_search_batch     wenet/transducer/search/prefix_beam_search.py:428-574
  joint / predictor_step            wenet/transducer/joint.py:64-94, wenet/transducer/predictor.py:185-206
Pinned by tests/golden/search_c5.pt (captured from the reference modules)."""
import math
from collections import defaultdict

import torch
import torch.nn.functional as F


def log_add(args):
    """wenet/utils/common.py:355-363."""
    if all(a == -float("inf") for a in args):
        return -float("inf")
    m = max(args)
    return m + math.log(sum(math.exp(a - m) for a in args))


def ctc_prefix_beam_search(ctc_probs, ctc_lens, beam_size, blank_id=0):
    out = []
    for i in range(ctc_probs.shape[0]):
        cur = [(tuple(), [0.0, -float("inf")])]           # prefix -> [blank-ending, non-blank-ending]
        for t in range(int(ctc_lens[i])):
            logp = ctc_probs[i][t]
            nxt = defaultdict(lambda: [-float("inf"), -float("inf")])
            for u in logp.topk(beam_size)[1]:
                u = u.item()
                prob = logp[u].item()
                for prefix, (s, ns) in cur:
                    last = prefix[-1] if prefix else None
                    tot = log_add([s, ns])
                    if u == blank_id:
                        nxt[prefix][0] = log_add([nxt[prefix][0], tot + prob])
                    elif u == last:
                        nxt[prefix][1] = log_add([nxt[prefix][1], ns + prob])
                        nxt[prefix + (u,)][1] = log_add([nxt[prefix + (u,)][1], s + prob])
                    else:
                        nxt[prefix + (u,)][1] = log_add([nxt[prefix + (u,)][1], tot + prob])
            cur = sorted(nxt.items(), key=lambda kv: log_add(kv[1]), reverse=True)[:beam_size]
        out.append(dict(tokens=list(cur[0][0]), score=log_add(cur[0][1]), nbest=[list(p) for p, _ in cur],
                        nbest_scores=[log_add(v) for _, v in cur]))
    return out


def joint(enc, pred, sd, p="joint."):
    e = F.linear(enc, sd[p + "enc_ffn.weight"], sd[p + "enc_ffn.bias"]).unsqueeze(2)
    q = F.linear(pred, sd[p + "pred_ffn.weight"], sd[p + "pred_ffn.bias"]).unsqueeze(1)
    return F.linear(torch.tanh(e + q), sd[p + "ffn_out.weight"], sd[p + "ffn_out.bias"])


def _lstm_step(x, h, c, sd, p, layers):
    """one time step of a `layers`-deep LSTM, batch_first, eval mode (torch.nn.LSTM gate order i, f, g, o)."""
    hs, cs = [], []
    for l in range(layers):
        g = (F.linear(x, sd[f"{p}rnn.weight_ih_l{l}"], sd[f"{p}rnn.bias_ih_l{l}"])
             + F.linear(h[l], sd[f"{p}rnn.weight_hh_l{l}"], sd[f"{p}rnn.bias_hh_l{l}"]))
        i, f, gg, o = g.chunk(4, dim=-1)
        cn = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
        x = torch.sigmoid(o) * torch.tanh(cn)
        hs.append(x)
        cs.append(cn)
    return x, torch.stack(hs), torch.stack(cs)


def predictor_step(tokens, h, c, sd, p="predictor."):
    """tokens (n,), h/c (layers, n, H) -> (n, 1, out), h', c'."""
    layers = h.shape[0]
    x = F.embedding(tokens, sd[p + "embed.weight"])
    y, h2, c2 = _lstm_step(x, h, c, sd, p, layers)
    return F.linear(y, sd[p + "projection.weight"], sd[p + "projection.bias"]).unsqueeze(1), h2, c2


def rnnt_prefix_beam_search_batch(enc_outs, enc_lens, ctc_probs, sd, beam_size, ctc_weight=0.3, transducer_weight=0.7,
                                  blank=0):
    layers = sum(1 for k in sd if k.startswith("predictor.rnn.weight_ih_l"))
    H = sd["predictor.rnn.weight_hh_l0"].shape[1]
    results = []
    B = enc_outs.shape[0]
    beams = [[dict(hyp=[blank], score=0.0, h=torch.zeros(layers, 1, H), c=torch.zeros(layers, 1, H))] for _ in range(B)]
    for t in range(int(enc_lens.max())):
        for i in range(B):
            if t >= int(enc_lens[i]):
                continue
            bs = beams[i]
            toks = torch.tensor([b["hyp"][-1] for b in bs])
            h = torch.cat([b["h"] for b in bs], 1)
            c = torch.cat([b["c"] for b in bs], 1)
            pred, h2, c2 = predictor_step(toks, h, c, sd)
            logp = joint(enc_outs[i, t].expand(len(bs), 1, -1), pred, sd).log_softmax(-1).squeeze(1).squeeze(1)
            logp = torch.log(transducer_weight * torch.exp(logp) + ctc_weight * torch.exp(ctc_probs[i, t].unsqueeze(0)))
            top_p, top_i = logp.topk(beam_size)
            scores = (torch.tensor([b["score"] for b in bs]).unsqueeze(1) + top_p).flatten()
            order = torch.argsort(scores, descending=True)
            out, seen = [], set()
            for k in order.tolist():
                j, tok, sc = k // beam_size, int(top_i.flatten()[k]), scores[k].item()
                base = bs[j]
                hyp = list(base["hyp"]) if tok == blank else base["hyp"] + [tok]
                if tuple(hyp) in seen:
                    for ex in out:
                        if ex["hyp"] == hyp:
                            ex["score"] = log_add([ex["score"], sc])
                            break
                else:
                    seen.add(tuple(hyp))
                    keep_old = tok == blank
                    out.append(dict(hyp=hyp, score=sc, h=base["h"] if keep_old else h2[:, j:j + 1],
                                    c=base["c"] if keep_old else c2[:, j:j + 1]))
                    if len(out) >= beam_size:
                        break
            out.sort(key=lambda d: d["score"], reverse=True)
            beams[i] = out[:beam_size]
    for bs in beams:
        results.append(dict(tokens=bs[0]["hyp"][1:], score=bs[0]["score"], nbest=[b["hyp"][1:] for b in bs],
                            nbest_scores=[b["score"] for b in bs]))
    return results
