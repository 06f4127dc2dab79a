"""CPU ORACLE of the text-generation row -- TEST INFRASTRUCTURE ONLY.

Reference: LlamaMultiModal.generate_batch (scripts/train.py:577-654) hands HF `generate` the image tokens + prompt
embeddings and samples with temperature 0.9, top-k 40, top-p 0.9, repetition penalty 1.2, no-repeat-3-grams
(:628-642).  Its embedding monkey-patch (:612-626) is not a well-defined function of the inputs once the KV cache is
on (every cached step is embedded as the first image token), so the semantics restated here -- and implemented by the
HIP path -- are the ones the reference evidently intends: prefix = [16 image tokens | the prompt's valid tokens], every
generated token is embedded like a text token (embed_tokens(id) + text_modality_embedding, train.py:526-527) at the
next position, logits = lm_head(final RMSNorm(hidden)) with lm_head tied to embed_tokens.

  * greedy_generate: the oracle decoder (oracle/forward.py) re-run on the growing sequence, one sample at a time.
  * process_logits / select_token: HF's logits processors (RepetitionPenalty, NoRepeatNGram, Temperature, TopK, TopP, in
    that order) and the token draw, restated in numpy exactly as csrc/generate.hip computes them (same ordering of ties,
    same Philox uniform); pinned against transformers' own processor classes by tests/golden/tiny_generation.npz.
"""
import numpy as np
import torch

from . import forward as O
from . import philox


def prefix_embeds(W, cfg, vision, input_ids, r, b, n_text):
    """[1, Nq + n_text, H]: image tokens + the first n_text prompt tokens of sample b (train.py:519-528)."""
    W = O.as_torch(W)
    r = O._rounder(r)
    img = O.linear(O.qformer(W, cfg, vision[b:b + 1], r), W, "mllm.q_proj", r.scoped("qf")) + W["mllm.vision_modality_embedding"]
    txt = r(W[O.LLAMA + "embed_tokens.weight"][input_ids[b:b + 1, :n_text]], "emb") + W["mllm.text_modality_embedding"]
    return torch.cat([img, txt], dim=1)


def token_embed(W, tok, r):
    W = O.as_torch(W)
    return (O._rounder(r)(W[O.LLAMA + "embed_tokens.weight"][tok], "emb") + W["mllm.text_modality_embedding"].reshape(-1))


def next_logits(W, cfg, embeds, r):
    """logits of the position after the last one of `embeds` [1, n, H] (no padding)."""
    W = O.as_torch(W)
    r = O._rounder(r)
    mask = torch.ones(1, embeds.shape[1], dtype=torch.int64)
    fh = O.llama_decoder(W, cfg, embeds, mask, r)
    return (r(fh[0, -1], "fh") @ r(W[O.LLAMA + "embed_tokens.weight"], "emb").T)


def greedy_generate(W, cfg, vision, input_ids, attention_mask, max_new_tokens, contract="fp32", forced=None):
    """-> (tokens [B, N] int64, margins [B, N] = top-1 minus top-2 logit).  forced [B, N] (optional): teacher forcing --
    the sequence continues with forced[b, i] whatever the arg-max says (margins / tokens of every step stay comparable
    after a near-tie)."""
    B = input_ids.shape[0]
    toks = torch.zeros(B, max_new_tokens, dtype=torch.int64)
    margins = torch.zeros(B, max_new_tokens)
    with torch.no_grad():
        for b in range(B):
            n_text = int(attention_mask[b].sum())
            seq = prefix_embeds(W, cfg, vision, input_ids, contract, b, n_text)
            for i in range(max_new_tokens):
                lg = next_logits(W, cfg, seq, contract)
                top = torch.topk(lg, 2)
                toks[b, i] = int(top.indices[0])
                margins[b, i] = float(top.values[0] - top.values[1])
                nxt = toks[b, i] if forced is None else forced[b, i]
                seq = torch.cat([seq, token_embed(W, nxt, contract)[None, None]], dim=1)
    return toks, margins


# ---------------------------------------------------------------------------------------------
# logits processors + selection, as csrc/generate.hip (sample_kernel) computes them
# ---------------------------------------------------------------------------------------------
def process_logits(scores, history, repetition_penalty=1.2, no_repeat_ngram_size=3):
    """RepetitionPenaltyLogitsProcessor + NoRepeatNGramLogitsProcessor on one row (numpy fp32, returns a copy)."""
    x = np.array(scores, dtype=np.float32, copy=True)
    hist = [int(t) for t in history]
    if repetition_penalty != 1.0:
        for t in sorted(set(hist)):
            x[t] = x[t] * np.float32(repetition_penalty) if x[t] < 0 else x[t] / np.float32(repetition_penalty)
    n = no_repeat_ngram_size
    if n > 0 and len(hist) + 1 >= n:
        tail = hist[len(hist) - (n - 1):] if n > 1 else []
        for i in range(len(hist) - n + 1):
            if hist[i:i + n - 1] == tail:
                x[hist[i + n - 1]] = -np.inf
    return x


def warp_candidates(x, temperature=0.9, top_k=40, top_p=0.9):
    """Temperature, top-k (ties with the k-th value kept), top-p (low tail with cumulative probability <= 1 - top_p
    dropped, at least one kept) -> (indices, weights exp(v - max)) of the kept candidates in (value desc, index asc) order."""
    v = x.astype(np.float32) * np.float32(1.0 / np.float32(temperature))
    order = np.lexsort((np.arange(v.size), -v))  # value descending, index ascending
    order = order[np.isfinite(v[order])]
    k = min(top_k, order.size)
    kth = v[order[k - 1]]
    keep = order[v[order] >= kth][:256]
    w = np.exp((v[keep] - v[keep[0]]).astype(np.float32)).astype(np.float32)
    if top_p < 1.0:
        tot = np.float32(0)
        for e in w:
            tot = np.float32(tot + e)
        tail, cut = np.float32(0), keep.size
        for j in range(keep.size - 1, 0, -1):
            tail = np.float32(tail + np.float32(w[j] / tot))
            if tail <= np.float32(1.0) - np.float32(top_p):
                cut = j
            else:
                break
        keep, w = keep[:cut], w[:cut]
    return keep, w


def philox_uniform(seed, step, row):
    c = philox.philox4x32_10(np.array([step], np.uint32), np.array([row], np.uint32), np.array([0x5A3B], np.uint32),
                             np.array([0], np.uint32), seed & 0xFFFFFFFF, seed >> 32)
    return np.float32((int(c[0][0]) >> 8) * (1.0 / 16777216.0))


def select_token(x, do_sample, temperature=0.9, top_k=40, top_p=0.9, seed=0, step=0, row=0):
    """x: processed scores of one row.  Greedy: first maximum.  Sampling: inverse CDF over the kept candidates with the
    kernel's Philox uniform of (seed, step, row)."""
    if not do_sample:
        return int(np.argmax(x))
    keep, w = warp_candidates(x, temperature, top_k, top_p)
    kt = np.float32(0)
    for e in w:
        kt = np.float32(kt + e)
    u = np.float32(philox_uniform(seed, step, row) * kt)
    acc = np.float32(0)
    for j in range(keep.size):
        acc = np.float32(acc + w[j])
        if u < acc:
            return int(keep[j])
    return int(keep[-1])
