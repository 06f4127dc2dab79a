"""BASELINE.json configurations exercised on one MI355X at their own sizes.

  configs[1] "train.py multimodal-LLM, 1xMI355X bf16, batch 32, LoRA rank 8, seq_len 256"  -> test_config2_full_size_*
  configs[4] "test_10.py inference-only, ..., hipGraph-captured decode"                      -> test_hipgraph_replay_*
  (configs[2], 8 GPUs, cannot run on a one-GPU box: tests/test_dp_gloo.py + tests/test_bench_spawn_cpu.py cover the
   rank logic on CPU.)

At the full size the oracle runs on two samples only (seconds); the whole batch is held to size-independent properties:
finiteness, bit-reproducibility, independence of the samples of a batch from one another, and the causal / key-valid
mask property that the ids of padded positions cannot reach any valid row (reference: scripts/train.py:531-532 mask,
HF modeling_llama.py:191-213 causal AND key-valid attention).
"""
import pytest
import torch

from tests.util import batch_tensors, load_case, rel_err

pytestmark = pytest.mark.gpu


def _fwd(m, g, with_loss=False):
    kw = dict(input_ids=g["input_ids"], attention_mask=g["attention_mask"], labels=g.get("labels"))
    if with_loss:
        kw.update(y=g["target_traj"], norm_stat=g["norm_stat"])
    return m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], **kw)


# ------------------------------------------------------------------------------------------------
# configs[4]: the inference pass replayed as a hipGraph (scripts/test_10.py:1301-1342 loop body)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny_6_12_lora_ragged", "tiny_18_30_nolora_ragged"])
def test_hipgraph_replay_bit_equal_eager(gpu, name):
    """Capture the eval forward (loss + decoded) once, replay it on NEW input contents written into the captured
    buffers: bit-equal to the eager pass on the same inputs, twice in a row."""
    from tcavt_amd import model

    dev = gpu["device"]
    cfg, weights, fx = load_case(name)
    t = batch_tensors(fx)
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    static = {k: v.to(dev).clone() for k, v in t.items()}
    # a second batch: permuted samples, perturbed trajectories / vision embeddings, other (valid) token ids
    gen = torch.Generator().manual_seed(5)
    perm = torch.randperm(t["traj_emb"].shape[0], generator=gen)
    other = {k: v[perm].clone() for k, v in t.items()}
    other["vision_emb"] = other["vision_emb"] + 0.25 * torch.randn(other["vision_emb"].shape, generator=gen)
    other["traj_emb"] = (other["traj_emb"] * 0.9 + 0.05).contiguous()
    other["input_ids"] = (other["input_ids"] * 7 + 3) % cfg.llama.vocab
    with torch.no_grad():
        for _ in range(2):  # warm-up: packed weights, workspaces, side streams
            _fwd(m, static, with_loss=True)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _fwd(m, static, with_loss=True)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss, g_dec = _fwd(m, static, with_loss=True)
        for batch in (other, t, other):
            for k in static:
                static[k].copy_(batch[k].to(dev))
            graph.replay()
            torch.cuda.synchronize()
            got_loss, got_dec = g_loss.clone(), g_dec.clone()
            eager = {k: v.to(dev) for k, v in batch.items()}
            e_loss, e_dec = _fwd(m, eager, with_loss=True)
            torch.cuda.synchronize()
            assert torch.equal(got_dec, e_dec), "hipGraph replay differs from the eager pass"
            assert torch.equal(got_loss, e_loss)
    m.mllm.check_flags()
    assert not torch.equal(got_dec.cpu(), torch.from_numpy(fx["exp_decoded"]).to(got_dec.dtype))  # (the batches do differ)


# ------------------------------------------------------------------------------------------------
# configs[1]: B = 32, L = 256, Llama-3.2-1B shape, LoRA r = 8
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_model(gpu):
    from tcavt_amd import config, model, synth
    from tcavt_amd.weights import make_weights

    dev = gpu["device"]
    cfg = config.llama32_1b()
    torch.set_num_threads(16)
    W = make_weights(cfg, seed=1, backend="torch", device="cpu")  # host-generated: shared with the oracle below
    with torch.device(dev):
        m = model.MultiModalTrajectoryModel.from_config(cfg)
    m.load_weights(W).eval()
    b = synth.make_batch(cfg, 32, text_len=240, seed=100, ragged=True, min_text=128)
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    yield cfg, W, m, t
    del m
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_config2_full_size_properties(gpu, full_model):
    cfg, W, m, t = full_model
    dev = gpu["device"]
    g = {k: v.to(dev) for k, v in t.items()}
    B, Nq = 32, cfg.q_num_query_tokens
    with torch.no_grad():
        loss1, dec1 = _fwd(m, g, with_loss=True)
        fh1 = m.last.final_hidden.clone()
        loss2, dec2 = _fwd(m, g, with_loss=True)
        torch.cuda.synchronize()
        m.mllm.check_flags()
        assert dec1.shape == (B, 2, cfg.out_len) and fh1.shape == (B, Nq + 240, cfg.llama.hidden)
        assert torch.isfinite(dec1).all() and torch.isfinite(fh1).all() and torch.isfinite(loss1)
        # bit-reproducible: no order-dependent reduction anywhere in the forward
        assert torch.equal(dec1, dec2) and torch.equal(loss1, loss2) and torch.equal(fh1, m.last.final_hidden)
        # key-valid AND causal mask: ids at padded positions cannot reach a valid row of final_hidden
        mask = g["attention_mask"].bool()
        assert (~mask).any(), "the ragged batch must contain padding"
        g2 = dict(g)
        g2["input_ids"] = torch.where(mask, g["input_ids"], (g["input_ids"] + 12345) % cfg.llama.vocab)
        _fwd(m, g2)
        valid = torch.cat([torch.ones(B, Nq, dtype=torch.bool, device=dev), mask], dim=1)
        fh2 = m.last.final_hidden
        assert torch.equal(fh1[valid], fh2[valid])
        assert not torch.equal(fh1[~valid], fh2[~valid])  # (padded rows do see their own embeddings)
        # kv_len: a key at or beyond kv_len[b] is never attended -- truncating the mask of one sample further changes
        # only that sample
        g3 = dict(g)
        am = g["attention_mask"].clone()
        am[3, 100:] = 0
        g3["attention_mask"] = am
        dec3 = _fwd(m, g3)
        keep = torch.arange(B, device=dev) != 3
        assert torch.equal(dec3[keep], dec1[keep]) and not torch.equal(dec3[3], dec1[3])
        # samples are independent: the first two samples alone (other GEMM tile forms at M = 512) give the same result
        g4 = {k: v[:2].contiguous() for k, v in g.items()}
        dec4 = _fwd(m, g4)
        e = rel_err(dec4.cpu(), dec1[:2].cpu())
        print(f"[config 2] B=2 vs rows 0-1 of B=32: rel {e:.2e} ({'bit-equal' if torch.equal(dec4, dec1[:2]) else 'not bit-equal'})")
        assert e < 1e-3


@pytest.mark.timeout(1200)
def test_config2_full_size_two_samples_against_oracle(gpu, full_model):
    """Two samples of the full-size batch against the CPU oracle with the same host-generated weights (fp32 graph and the
    fp16-contract graph): decoded, ADE and FDE within 1e-3 of the fp32 result."""
    from oracle import forward as O

    cfg, W, m, t = full_model
    dev = gpu["device"]
    t2 = {k: v[:2].contiguous() for k, v in t.items()}
    g2 = {k: v.to(dev) for k, v in t2.items()}
    with torch.no_grad():
        dec = _fwd(m, g2).float().cpu()
        torch.cuda.synchronize()
        args = (W, cfg, t2["traj_emb"], t2["vision_emb"], t2["lane_polygon"], t2["lane_polygon_len"], t2["input_ids"],
                t2["attention_mask"])
        _, d32 = O.model_forward(*args, y=t2["target_traj"], norm_stat=t2["norm_stat"], contract="fp32")
        _, d16 = O.model_forward(*args, y=t2["target_traj"], norm_stat=t2["norm_stat"], contract="fp16")
    f_dec, o_dec, e_dec = rel_err(dec, d32), rel_err(d16, d32), rel_err(dec, d16)
    mg, m32, m16 = (O.traj_metrics(d, t2["target_traj"], t2["norm_stat"]) for d in (dec, d32, d16))
    ade16 = abs(mg["ade_sum"] - m16["ade_sum"]) / m16["ade_sum"]
    fde16 = abs(mg["fde_sum"] - m16["fde_sum"]) / m16["fde_sum"]
    ade32 = abs(mg["ade_sum"] - m32["ade_sum"]) / m32["ade_sum"]
    fde32 = abs(mg["fde_sum"] - m32["fde_sum"]) / m32["fde_sum"]
    print(f"[config 2 vs oracle] decoded: vs fp32 {f_dec:.2e} (contract's own {o_dec:.2e}), vs contract {e_dec:.2e}; "
          f"ADE/FDE vs contract {ade16:.2e}/{fde16:.2e}, vs fp32 {ade32:.2e}/{fde32:.2e}")
    # BASELINE.json's bar at the full model size, against the reference's fp32 arithmetic (fp16 operand storage)
    assert f_dec < 1e-3 and e_dec < 1e-3
    assert ade32 < 1e-3 and fde32 < 1e-3 and ade16 < 1e-3 and fde16 < 1e-3


@pytest.mark.timeout(900)
@pytest.mark.parametrize("B", [4, 20])  # activations in one block of 8 tokens / in blocks of 16 with the token blocks on two workgroups
def test_config2_full_size_decode_step_equals_prefill_of_the_extended_sequence(gpu, full_model, B):
    """Text generation at the full model size through a size-independent round trip: after N greedy tokens from the KV cache
    (tcavt_llama_decode_step: skinny GEMMs, decode attention, per-sample RoPE positions on ragged prompts, hipGraph replay),
    the logits that selected token N must equal the logits a fresh PREFILL (tile GEMMs, the causal GQA kernel) computes for
    the prompt extended by the first N - 1 generated tokens -- two disjoint kernel paths to the same quantity
    (scripts/train.py:577-654 semantics as stated in DESIGN.md section 7)."""
    cfg, W, m, t = full_model
    dev = gpu["device"]
    N = 6
    assert t["input_ids"].shape[0] >= B
    g = {k: t[k][:B].contiguous().to(dev) for k in ("vision_emb", "input_ids", "attention_mask")}
    kw = dict(do_sample=False, repetition_penalty=1.0, no_repeat_ngram_size=0, use_graph=True)
    out = m.mllm.generate_batch(g["vision_emb"], None, max_new_tokens=N, input_ids=g["input_ids"],
                                attention_mask=g["attention_mask"], **kw)
    torch.cuda.synchronize()
    V = cfg.llama.vocab
    last = m.mllm._ws.get("gen.logits", (B, V), torch.float32, dev).clone()
    assert out.shape == (B, N) and torch.isfinite(last).all()
    # the prompt of every sample extended by its own first N - 1 generated tokens (right padding keeps the samples ragged)
    Lt = g["input_ids"].shape[1]
    ids = torch.zeros(B, Lt + N - 1, dtype=g["input_ids"].dtype, device=dev)
    am = torch.zeros(B, Lt + N - 1, dtype=g["attention_mask"].dtype, device=dev)
    for b in range(B):
        n = int(g["attention_mask"][b].sum())
        ids[b, :n] = g["input_ids"][b, :n]
        ids[b, n:n + N - 1] = out[b, :N - 1]
        am[b, :n + N - 1] = 1
    out2 = m.mllm.generate_batch(g["vision_emb"], None, max_new_tokens=1, input_ids=ids, attention_mask=am, **kw)
    torch.cuda.synchronize()
    first = m.mllm._ws.get("gen.logits", (B, V), torch.float32, dev)
    errs = [rel_err(last[b].cpu(), first[b].cpu()) for b in range(B)]
    print("[config 2 generation] decode-step logits vs prefill of the extended sequence:", [f"{e:.1e}" for e in errs])
    assert max(errs) < 2e-3
    for b in range(B):  # the same token, unless the two leading logits are closer than the paths agree
        top2 = torch.topk(first[b], 2).values
        if (top2[0] - top2[1]).item() > 4e-3 * first[b].abs().max().item():
            assert int(out2[b, 0]) == int(out[b, N - 1]), b


def _gradient_linearity(tr, t, dev, tag):
    """gradient(two-sample batch) vs mean(gradient(sample 0), gradient(sample 1)) over the trainer's flat gradient."""
    keys = ("traj_emb", "vision_emb", "lane_polygon", "lane_polygon_len", "target_traj", "norm_stat", "input_ids",
            "attention_mask", "labels")

    def grads(rows):
        g = {k: t[k][rows].contiguous().to(dev) for k in keys}
        loss, _ = tr.forward_backward(*[g[k] for k in keys])
        torch.cuda.synchronize()
        return loss.item(), tr.book.grads.detach().clone()

    l01, g01 = grads(slice(0, 2))
    l0, g0 = grads(slice(0, 1))
    l1, g1 = grads(slice(1, 2))
    mean = 0.5 * (g0 + g1)
    assert abs(0.5 * (l0 + l1) - l01) / l01 < 1e-4
    flat = ((g01 - mean).double().norm() / mean.double().norm()).item()
    worst, worst_name = 0.0, None
    for name in tr.book.names:
        o, n, _ = tr.book.offsets[name]
        a, b = g01[o:o + n], mean[o:o + n]
        if b.abs().max() == 0:
            assert a.abs().max() == 0, name
            continue
        e = rel_err(a.cpu(), b.cpu())
        if e > worst:
            worst, worst_name = e, name
    print(f"[config 2 backward, {tag}] two-sample gradient vs mean of one-sample gradients: flat {flat:.2e}, "
          f"worst tensor {worst:.2e} ({worst_name})")
    assert torch.isfinite(g01).all()
    return flat, worst


@pytest.mark.timeout(900)
def test_config2_full_size_gradient_is_the_mean_of_per_sample_gradients(gpu, full_model):
    """The training step's backward (scripts/train.py:1168-1183) at the full model size through a size-independent property:
    the loss is a mean over the batch and the samples are independent, so the gradient of a two-sample batch is the mean of
    the two one-sample gradients -- every trainable tensor, computed through different GEMM shapes (M = 512 vs 256 rows in
    the decoder, 60 vs 30 rows in the head's contractions).  This is also what makes data-parallel averaging exact."""
    from tcavt_amd import training

    cfg, W, m, t = full_model
    flags = (m.pipeline_decoder, m.mllm.skip_f32_hidden)
    tr = training.Trainer(m, lr=1e-4)
    try:
        flat, worst = _gradient_linearity(tr, t, gpu["device"], "train.py set")
        assert flat < 6e-4 and worst < 5e-3  # 3 x the measured 1.9e-4 / 1.7e-3
    finally:
        m.pipeline_decoder, m.mllm.skip_f32_hidden = flags


@pytest.mark.timeout(900)
def test_config2_full_size_lora_gradient_is_the_mean_of_per_sample_gradients(gpu, full_model):
    """The same property for the whole trainable set of modify_scripts/modify_train.py (:512-528: adapters, Q-Former, q_proj,
    modality embeddings next to the train.py set): the backward walks all 16 decoder layers (bf16 tapes, recomputed
    attention probabilities, skinny adapter contractions over M = 512 / 256 token rows) and the Q-Former."""
    from tcavt_amd import model, training

    cfg, W, _, t = full_model
    dev = gpu["device"]
    with torch.device(dev):
        m2 = model.MultiModalTrajectoryModel.from_config(cfg)
    m2.load_weights(W).eval()
    tr = training.Trainer(m2, lr=1e-4, lora_trainable=True, train_mllm_front=True)
    flat, worst = _gradient_linearity(tr, t, dev, "modify_train.py set")
    assert flat < 5e-3 and worst < 1.8e-2  # 3 x the measured 1.8e-3 / 6.0e-3 (16-bit tapes: the two batch shapes round differently)
    del tr, m2
    torch.cuda.empty_cache()
