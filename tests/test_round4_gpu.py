"""Round-4 additions on the GPU: the tokenizer branch of LlamaMultiModal.forward (scripts/train.py:556-575), and feeding the
training step from the host (scripts/train.py:1153-1166) through data.DeviceFeeder without two batches in flight aliasing."""
import numpy as np
import pytest
import torch

from tests.util import batch_tensors, load_case, rel_err

pytestmark = pytest.mark.gpu


def test_tokenizer_branch_of_the_mllm_forward(gpu):
    """context_str alone (no input_ids): tokenised with padding as the reference does, then the ids branch -- so the result is
    BIT-equal to calling the ids branch with the tokenizer's output; without a tokenizer the call is refused."""
    from tcavt_amd import model
    from tcavt_amd.synth import SyntheticTokenizer

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    vision = torch.from_numpy(np.asarray(fx["vision_emb"])).to(dev)
    B = vision.shape[0]
    ctx = [f"A1: vehicle {i} drives in lane A{1 + i % 3} of Site C moving right to left.\n" + "A2: speed 41.5 km/h. " * (1 + i)
           for i in range(B)]
    with pytest.raises(NotImplementedError):
        m.mllm(vision, ctx)
    tok = SyntheticTokenizer(vocab=cfg.llama.vocab)
    m.mllm.tokenizer = tok
    with torch.no_grad():
        fh, nq = m.mllm(vision, ctx)
        enc = tok(ctx, return_tensors="pt", padding=True, truncation=True)
        assert int(enc["attention_mask"].sum(1).min()) < enc["input_ids"].shape[1]   # (ragged rows: the padding is exercised)
        fh2, nq2 = m.mllm(vision, None, input_ids=enc["input_ids"].to(dev), attention_mask=enc["attention_mask"].to(dev))
    torch.cuda.synchronize()
    m.mllm.check_flags()
    assert nq == nq2 == cfg.q_num_query_tokens and tuple(fh.shape) == (B, nq + enc["input_ids"].shape[1], cfg.llama.hidden)
    assert torch.equal(fh, fh2) and torch.isfinite(fh).all()
    # ... and through the model's forward (train.py:914-924 hands context_str down)
    t = batch_tensors(fx)
    g = {k: v.to(dev) for k, v in t.items()}
    with torch.no_grad():
        d1 = m(g["traj_emb"], g["vision_emb"], ctx, g["lane_polygon"], g["lane_polygon_len"])
        d2 = m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], input_ids=enc["input_ids"].to(dev),
               attention_mask=enc["attention_mask"].to(dev))
    assert torch.equal(d1, d2)


def test_steps_fed_from_the_host_do_not_alias(gpu):
    """Four different batches go through data.DeviceFeeder's three-slot ring into the pipelined training step (the next batch's
    copy and Q-Former prefetch are enqueued before the current step, no host synchronisation anywhere): the frozen MLLM's
    output of every step is BIT-equal to the same steps on resident copies of the batches, losses and parameters agree."""
    from tcavt_amd import data, model, synth, training

    dev = gpu["device"]
    cfg, weights, fx = load_case("tiny_6_12_lora_ragged")
    B, Lt = 4, 20
    host = [synth.batch_to_samples(synth.make_batch(cfg, B, text_len=Lt, seed=50 + j, ragged=True)) for j in range(4)]
    n_steps = 9

    def run(feed):
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).train()
        tr = training.Trainer(m, lr=1e-4)
        hid, losses = [], []
        if feed:
            fd = data.DeviceFeeder(dev)
            cur = fd.put(data.custom_collate_fn(host[0]))
            for i in range(n_steps):
                nxt = fd.put(data.custom_collate_fn(host[(i + 1) % 4]))
                loss, _ = tr.step(cur["traj_emb"], cur["vision_emb"], cur["lane_polygon"], cur["lane_polygon_len"], cur["target_traj"],
                                  cur["norm_stat"], cur["input_ids"], cur["attention_mask"], cur["labels"],
                                  next_vision_embs=nxt["vision_emb"], next_ready=nxt.ready, inputs_ready=cur.ready)
                fd.release(cur)
                hid.append(m.last.final_hidden_bf16.clone())
                losses.append(loss)
                cur = nxt
            assert fd.bytes_per_batch > 0
        else:
            res = []
            for j in range(4):
                c = data.custom_collate_fn(host[j])
                res.append({k: (c[k].to(dev) if torch.is_tensor(c[k]) else c[k]) for k in c})
            for i in range(n_steps):
                g = res[i % 4]
                loss, _ = tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                                  g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"], inputs_ready=True)
                hid.append(m.last.final_hidden_bf16.clone())
                losses.append(loss)
        torch.cuda.synchronize()
        m.mllm.check_flags()
        return hid, [l.item() for l in losses], tr.book.params.clone()

    ha, la, pa = run(False)
    hb, lb, pb = run(True)
    for i in range(n_steps):
        assert torch.equal(ha[i], hb[i]), f"step {i}: the MLLM saw another batch's inputs"
    assert not torch.equal(ha[0], ha[1])
    assert np.allclose(la, lb, rtol=5e-3) and rel_err(pa.cpu(), pb.cpu()) < 1e-3


def test_copy_batch_from_pinned_host_memory(gpu):
    """tcavt_copy_batch: up to 16 (device <- pinned host) copies in one kernel launch; sizes that are not multiples of 16 bytes."""
    from tcavt_amd import capi, ops

    dev = gpu["device"]
    g = torch.Generator().manual_seed(4)
    shapes = [((32, 2, 18), torch.float32), ((32, 18, 512), torch.float32), ((32,), torch.int32), ((32, 4), torch.float32),
              ((32, 240), torch.int64), ((7,), torch.int32), ((3, 5), torch.float16), ((1,), torch.int64)]
    srcs = []
    for shp, dt in shapes:
        t = torch.randint(-1000, 1000, shp, generator=g).to(dt) if dt in (torch.int32, torch.int64) else torch.randn(shp, generator=g).to(dt)
        srcs.append(t.pin_memory())
    dsts = [torch.zeros(s.shape, dtype=s.dtype, device=dev) for s in srcs]
    ops.copy_batch(dsts, srcs)
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert torch.equal(d.cpu(), s)
    with pytest.raises(capi.TcavtError):
        ops.copy_batch(dsts[:1], [torch.zeros(5)])  # size mismatch / not pinned
