"""Text generation on the HIP path (scripts/train.py:577-654, scripts/check_generation.py:152-222):
prefill with KV cache + one decode step per token + device-side logits processors / selection, hipGraph replay.
Token ids are held BIT-EXACT against (1) the continuation the reference model's own HF LlamaForCausalLM produces
(tests/golden/tiny_generation.npz) and (2) the oracle (oracle/generation.py) -- "bit-exact for token/index selection"."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, load_generation_case, rel_err

pytestmark = pytest.mark.gpu


def _setup(dev):
    from tcavt_amd import model

    fx, cfg, weights, t = load_generation_case()
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    g = {k: v.to(dev) for k, v in t.items()}
    return fx, cfg, weights, t, g, m


@pytest.mark.parametrize("use_graph", [False, True])
def test_greedy_tokens_match_reference_model(gpu, use_graph):
    fx, cfg, weights, t, g, m = _setup(gpu["device"])
    N = fx["greedy_tokens"].shape[1]
    out = m.mllm.generate_batch(g["vision_emb"], None, max_new_tokens=N, input_ids=g["input_ids"],
                                attention_mask=g["attention_mask"], do_sample=False, repetition_penalty=1.0,
                                no_repeat_ngram_size=0, use_graph=use_graph)
    torch.cuda.synchronize()
    m.mllm.check_flags()
    assert out.dtype == torch.int64 and tuple(out.shape) == fx["greedy_tokens"].shape
    assert np.array_equal(out.cpu().numpy(), fx["greedy_tokens"])  # plain arg-max: >= 7 distinct tokens per sample, margins >= 0.1
    # the reference's processors in greedy mode: repetition penalty 1.2 + no-repeat-3-gram (train.py:639-640)
    out2 = m.mllm.generate_batch(g["vision_emb"], None, max_new_tokens=N, input_ids=g["input_ids"],
                                 attention_mask=g["attention_mask"], do_sample=False, repetition_penalty=1.2,
                                 no_repeat_ngram_size=3, use_graph=use_graph)
    assert np.array_equal(out2.cpu().numpy(), fx["greedy_proc_tokens"])


def test_decode_step_logits_match_oracle_on_growing_sequence(gpu):
    """The logits of the LAST decode step (KV cache, per-sample RoPE positions, ragged prompts) against the oracle decoder
    re-run on the whole sequence [image tokens | valid prompt | generated tokens], fp16 contract."""
    from oracle import generation as G

    dev = gpu["device"]
    fx, cfg, weights, t, g, m = _setup(dev)
    N = 7
    out = m.mllm.generate_batch(g["vision_emb"], None, max_new_tokens=N, input_ids=g["input_ids"],
                                attention_mask=g["attention_mask"], do_sample=False, repetition_penalty=1.0,
                                no_repeat_ngram_size=0, use_graph=True).cpu()
    B = out.shape[0]
    logits = m.mllm._ws.get("gen.logits", (B, cfg.llama.vocab), torch.float32, dev).cpu()
    for b in range(B):
        n_text = int(t["attention_mask"][b].sum())
        seq = G.prefix_embeds(weights, cfg, t["vision_emb"], t["input_ids"], "fp16", b, n_text)
        for tok in out[b, : N - 1]:
            seq = torch.cat([seq, G.token_embed(weights, tok, "fp16")[None, None]], dim=1)
        with torch.no_grad():
            ref = G.next_logits(weights, cfg, seq, "fp16")
        e = rel_err(logits[b], ref)
        assert e < 2e-3, (b, e)
        assert int(ref.argmax()) == int(out[b, N - 1])


def test_sampler_kernel_matches_oracle_and_reference_processors(gpu):
    """tcavt_sample_logits on fixed scores / histories: the token it draws equals the oracle's draw (same Philox uniform,
    same candidate order) for every (seed, step), and always lies in the set transformers' processors keep."""
    from oracle import generation as G
    from tcavt_amd import capi, ops

    dev = gpu["device"]
    fx = dict(np.load(os.path.join(GOLDEN, "tiny_generation.npz"), allow_pickle=False))
    R, V = fx["scores"].shape
    cap = fx["hist"].shape[1] + 4
    n_bad = 0
    for seed in (1, 2, 3):
        for step0 in (0, 5, 11, 40):
            logits = torch.from_numpy(fx["scores"]).to(dev).clone()
            hist = torch.zeros(R, cap, dtype=torch.int64, device=dev)
            hist[:, : fx["hist"].shape[1]] = torch.from_numpy(np.maximum(fx["hist"], 0)).to(dev)
            hist_len = torch.from_numpy(fx["hist_len"]).to(dev)
            step = torch.full((1,), step0, dtype=torch.int32, device=dev)
            cur = torch.zeros(R, dtype=torch.int64, device=dev)
            pos = torch.zeros(R, dtype=torch.int32, device=dev)
            fin = torch.zeros(R, dtype=torch.int32, device=dev)
            out = torch.full((R, 64), -1, dtype=torch.int64, device=dev)
            sp = capi.SampleParams(0.9, 0.9, 1.2, 40, 3, 1, -1, 0, seed)
            ops.sample_logits(logits, hist, hist_len, sp, step, cur, pos, fin, out, advance_pos=True)
            torch.cuda.synchronize()
            assert step.item() == step0 + 1 and (pos == 1).all() and (hist_len.cpu().numpy() == fx["hist_len"] + 1).all()
            for r in range(R):
                h = fx["hist"][r, : int(fx["hist_len"][r])]
                x = G.process_logits(fx["scores"][r], h, 1.2, 3)
                want = G.select_token(x, True, 0.9, 40, 0.9, seed=seed, step=step0, row=r)
                got = int(out[r, step0].item())
                assert np.isfinite(fx["warped"][r][got]), (r, got)  # inside the set the reference's warpers keep
                assert got == int(cur[r].item()) == int(hist[r, int(fx["hist_len"][r])].item())
                n_bad += got != want
            # the penalty / bans were applied to the logits in place exactly as transformers applies them
            proc = logits.cpu().numpy()
            assert np.array_equal(np.isinf(proc), np.isinf(fx["processed"]))
            assert np.allclose(proc[np.isfinite(proc)], fx["processed"][np.isfinite(proc)], rtol=1e-6)
    assert n_bad == 0


def test_eos_pads_the_rest_and_sampling_is_reproducible(gpu):
    fx, cfg, weights, t, g, m = _setup(gpu["device"])
    kw = dict(max_new_tokens=10, input_ids=g["input_ids"], attention_mask=g["attention_mask"])
    ref = m.mllm.generate_batch(g["vision_emb"], None, do_sample=False, repetition_penalty=1.2, no_repeat_ngram_size=3, **kw).cpu()
    eos = int(ref[0, 2])  # the third token of sample 0 becomes EOS: everything after it is the pad token
    out = m.mllm.generate_batch(g["vision_emb"], None, do_sample=False, repetition_penalty=1.2, no_repeat_ngram_size=3,
                                eos_token_id=eos, pad_token_id=7, **kw).cpu()
    first = int((ref[0] == eos).nonzero()[0])
    assert torch.equal(out[0, : first + 1], ref[0, : first + 1]) and (out[0, first + 1:] == 7).all()
    a = m.mllm.generate_batch(g["vision_emb"], None, do_sample=True, seed=5, **kw).cpu()
    b = m.mllm.generate_batch(g["vision_emb"], None, do_sample=True, seed=5, use_graph=False, **kw).cpu()
    c = m.mllm.generate_batch(g["vision_emb"], None, do_sample=True, seed=6, **kw).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c)  # a pure function of the seed; graph replay == eager
    assert ((a >= 0) & (a < cfg.llama.vocab)).all()


@pytest.mark.parametrize("V", [512, 128256, 50000])
def test_two_stage_token_selection_equals_the_one_stage_form(gpu, V):
    """tcavt_sample_logits with a workspace (B x 16 slice workgroups + one merging workgroup per sample, round 4) selects the
    SAME token and leaves the same processed logits and device state as the one-workgroup form, on rows built to be awkward:
    heavy ties at the top, a constant row (every slice's threshold bin overflows -> the per-sample fallback), a row with three
    finite scores, a row that is all -inf but one, penalties and bans that hit tokens in different slices; sampling and greedy."""
    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(V)
    R = 6
    base = torch.randn(R, V, generator=g) * 3.0
    base[1] = torch.round(base[1] * 2) / 2                      # many exact ties, also at the k-th value
    base[2] = 1.25                                              # constant: overflow of every slice list
    base[3] = float("-inf"); base[3, [5, V // 2, V - 1]] = torch.tensor([0.5, 2.0, 2.0])
    base[4] = float("-inf"); base[4, V - 3] = -7.0
    base[5, V // 3: V // 3 + 300] = base[5].max() + 1.0         # a plateau at the top wider than top_k
    cap = 40
    hist0 = torch.randint(0, V, (R, cap), generator=g)
    hist0[:, 30:] = 0
    hist0[0, :30] = torch.arange(30) % 7 + (V - 8)              # repeats + 3-gram matches in the last slice
    hl0 = torch.full((R,), 30, dtype=torch.int32)
    for do_sample, top_k, rep, ngram in ((1, 40, 1.2, 3), (1, 1, 1.0, 0), (1, 256, 1.3, 2), (0, 40, 1.2, 3), (0, 40, 1.0, 0)):
        res = []
        for two_stage in (False, True):
            logits = base.clone().to(dev)
            hist = hist0.clone().to(dev)
            hist_len = hl0.clone().to(dev)
            step = torch.full((1,), 3, dtype=torch.int32, device=dev)
            cur = torch.zeros(R, dtype=torch.int64, device=dev)
            pos = torch.zeros(R, dtype=torch.int32, device=dev)
            fin = torch.zeros(R, dtype=torch.int32, device=dev)
            out = torch.full((R, 8), -1, dtype=torch.int64, device=dev)
            sp = capi.SampleParams(0.9, 0.9, rep, top_k, ngram, do_sample, -1, 0, 1234)
            wsp = ops.sample_workspace(R, dev) if two_stage else None
            for _ in range(2):  # two calls in a row: the workspace's control words are left re-armed, *step advances once per call
                ops.sample_logits(logits, hist, hist_len, sp, step, cur, pos, fin, out, advance_pos=True, workspace=wsp)
            torch.cuda.synchronize()
            res.append((logits.cpu(), hist.cpu(), hist_len.cpu(), step.item(), cur.cpu(), pos.cpu(), out.cpu()))
            if two_stage:
                assert int(wsp[:64 + 4 * R].sum().item()) == 0  # ticket and overflow words back to zero
        a, b = res
        assert a[3] == b[3] == 5
        for x, y, nm in zip(a, b, ("logits", "history", "hist_len", "step", "cur", "pos", "out")):
            if nm == "logits":
                assert torch.equal(torch.isinf(x), torch.isinf(y)) and torch.equal(torch.nan_to_num(x, neginf=0.0), torch.nan_to_num(y, neginf=0.0)), (do_sample, top_k)
            elif nm != "step":
                assert torch.equal(x, y), (nm, do_sample, top_k, rep, ngram)


def test_decode_step_on_fragment_major_weights_is_bit_equal(gpu):
    """generate_batch streams fragment-major copies of the decoder weights in its decode steps (LlamaWithCrossAttnPEFT.
    decode_weights, tcavt_decode_args.w_layout): sampled token ids are identical to the row-major path (TCAVT_DECODE_ROWMAJOR=1),
    greedy and sampling, eager and graph replay."""
    import os

    from tcavt_amd import model

    fx, cfg, w, t = load_generation_case()
    dev = gpu["device"]
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(w, device=dev).eval()
    kw = dict(max_new_tokens=12, input_ids=t["input_ids"].to(dev), attention_mask=t["attention_mask"].to(dev))
    outs = {}
    for mode in ("frag", "frag16", "frag_w", "row"):  # "frag": the default (one block of 8 tokens at this batch size)
        if mode == "frag16":  # blocks of 16 tokens, as above 8 samples
            os.environ["TCAVT_DECODE_ACT_FRAG"] = "16"
        if mode == "row":
            os.environ["TCAVT_DECODE_ROWMAJOR"] = "1"
        if mode == "frag_w":  # fragment-major weights, row-major activations
            os.environ["TCAVT_DECODE_ACT_ROWMAJOR"] = "1"
        try:
            outs[mode] = [m.mllm.generate_batch(t["vision_emb"].to(dev), None, do_sample=False, repetition_penalty=1.0,
                                                no_repeat_ngram_size=0, **kw).clone(),
                          m.mllm.generate_batch(t["vision_emb"].to(dev), None, do_sample=True, seed=7, **kw).clone(),
                          m.mllm.generate_batch(t["vision_emb"].to(dev), None, do_sample=True, seed=7, use_graph=False, **kw).clone()]
        finally:
            os.environ.pop("TCAVT_DECODE_ROWMAJOR", None)
            os.environ.pop("TCAVT_DECODE_ACT_ROWMAJOR", None)
            os.environ.pop("TCAVT_DECODE_ACT_FRAG", None)
    torch.cuda.synchronize()
    m.mllm.check_flags()
    assert m.mllm.llama_wrapper._prep_dec is not None
    for a_, b_, c_, d_ in zip(outs["frag"], outs["row"], outs["frag_w"], outs["frag16"]):
        assert torch.equal(a_, b_) and torch.equal(c_, b_) and torch.equal(d_, b_)
    assert torch.equal(outs["frag"][1], outs["frag"][2])


@pytest.mark.parametrize("storage", ["bf16", "fp16_wide"])
def test_decode_step_on_fragment_major_weights_other_stream_contracts(gpu, storage):
    """The contracts whose residual stream is fp32 (bf16 storage; fp16 operands with wide_stream): the decode step still streams
    fragment-major WEIGHTS (activations stay row-major: the fp32 stream's norm kernels are row-major) and samples the same tokens
    as on row-major weights."""
    import os

    from tcavt_amd import model

    fx, cfg, w, t = load_generation_case()
    dev = gpu["device"]
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(w, device=dev).eval()
    if storage == "bf16":
        m.set_storage(torch.bfloat16)
    else:
        m.set_storage(torch.float16, wide_stream=True)
    kw = dict(max_new_tokens=10, input_ids=t["input_ids"].to(dev), attention_mask=t["attention_mask"].to(dev))
    outs = {}
    for mode in ("frag", "row"):
        if mode == "row":
            os.environ["TCAVT_DECODE_ROWMAJOR"] = "1"
        try:
            outs[mode] = [m.mllm.generate_batch(t["vision_emb"].to(dev), None, do_sample=False, repetition_penalty=1.0,
                                                no_repeat_ngram_size=0, **kw).clone(),
                          m.mllm.generate_batch(t["vision_emb"].to(dev), None, do_sample=True, seed=11, **kw).clone()]
        finally:
            os.environ.pop("TCAVT_DECODE_ROWMAJOR", None)
    torch.cuda.synchronize()
    m.mllm.check_flags()
    assert m.mllm.llama_wrapper._prep_dec is not None and not m.mllm.llama_wrapper.stream16
    for a_, b_ in zip(outs["frag"], outs["row"]):
        assert torch.equal(a_, b_)
