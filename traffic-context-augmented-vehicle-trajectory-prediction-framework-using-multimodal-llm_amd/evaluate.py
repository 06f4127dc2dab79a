"""Evaluation protocols of the reference scripts.

* ``ConstantVelocityPredictor`` / ``evaluate_cv``  -- scripts/baseline_cv.py:186-238, 280-360 (BASELINE.json
  configs[0]: CPU-only plumbing; no GPU, no hot path involved -- it validates the data builder, the collate
  layout and the metric definitions).
* ``evaluate_model`` -- the test loops of scripts/train.py:1274-1326 (single pass ADE/FDE) and
  scripts/test.py:1298-1382 (K candidates, min over K of ADE/FDE/RMSE), on the HIP path: predictions go
  through ``tcavt_traj_metrics`` (de-normalise, errors, min/argmin over K on the GPU), per-batch sums stay
  on the device, ranks exchange 4 scalars at the end (the reference evaluates on rank 0 only and so sees
  1/W of the test set, train.py:1255,1271 -- SURVEY.md 2c).
"""
import torch
import torch.distributed as dist


class ConstantVelocityPredictor(torch.nn.Module):
    """pred[:, i, t] = last + (last - prev + noise_i) * (t + 1); noise_i ~ N(0, noise_scale) per candidate, drawn in
    candidate order with torch.randn(B, 2) (baseline_cv.py:193-238) so that a seeded run reproduces the reference."""

    def __init__(self, seq_len, out_len, feature_size=2):
        super().__init__()
        self.seq_len, self.out_len, self.feature_size = seq_len, out_len, feature_size

    def forward(self, x, y=None, norm_stat=None, num_candidates=1, noise_scale=0.1):
        B = x.size(0)
        last, prev = x[:, :, -1], x[:, :, -2]
        velocity = last - prev
        steps = torch.arange(1, self.out_len + 1, device=x.device, dtype=x.dtype)
        cands = []
        for _ in range(num_candidates):
            v = velocity + torch.randn(B, self.feature_size, device=x.device) * noise_scale
            cands.append(last[:, None, :] + v[:, None, :] * steps[None, :, None])  # (B, T_out, 2)
        pred = torch.stack(cands, dim=1)  # (B, K, T_out, 2)
        if y is not None and norm_stat is not None:
            ns = torch.as_tensor(norm_stat, dtype=x.dtype, device=x.device)
            scale = torch.stack([ns[:, 1] - ns[:, 0], ns[:, 3] - ns[:, 2]], dim=1)
            shift = torch.stack([ns[:, 0], ns[:, 2]], dim=1)
            pd = pred * scale[:, None, None, :] + shift[:, None, None, :]
            gd = y.permute(0, 2, 1) * scale[:, None, :] + shift[:, None, :]
            loss = ((pd - gd[:, None]) ** 2).mean(dim=[2, 3])
            return loss, pred
        return pred


def candidate_metrics_cpu(pred_bkt2, y_b2t, norm_stat):
    """min-over-K ADE / FDE / RMSE sums on the host (baseline_cv.py:327-352).  pred (B,K,T,2), y (B,2,T)."""
    ns = torch.as_tensor(norm_stat, dtype=pred_bkt2.dtype)
    scale = torch.stack([ns[:, 1] - ns[:, 0], ns[:, 3] - ns[:, 2]], dim=1)
    shift = torch.stack([ns[:, 0], ns[:, 2]], dim=1)
    pd = pred_bkt2 * scale[:, None, None, :] + shift[:, None, None, :]
    gd = y_b2t.permute(0, 2, 1) * scale[:, None, :] + shift[:, None, :]
    diff = pd - gd[:, None]
    err = torch.sqrt((diff ** 2).sum(dim=-1))
    ade, fde = err.mean(dim=-1), err[..., -1]
    rmse = torch.sqrt((diff ** 2).mean(dim=[-2, -1]))
    return ade.min(dim=1).values.sum().item(), fde.min(dim=1).values.sum().item(), rmse.min(dim=1).values.sum().item()


def evaluate_cv(tracks, seq_len=6, out_len=30, batch_size=16, stride=6, downsample=5, max_step=50.0,
                max_speed_diff=30.0, num_candidates=10, noise_scale=0.1, split=True):
    """baseline_cv.py:280-360 on an in-memory track list -> (minADE, minFDE, minRMSE, n_samples)."""
    from . import data

    if split:
        _, _, tracks = data.split_all_data(tracks, 0.7, 0.2, 0.1)
    ins, outs = data.build_dataset_from_tracks_sliding(tracks, seq_len=seq_len, out_len=out_len, stride=stride,
                                                       max_step=max_step, max_speed_diff=max_speed_diff,
                                                       downsample=downsample)
    model = ConstantVelocityPredictor(seq_len, out_len).eval()
    tot = [0.0, 0.0, 0.0]
    n = 0
    # The reference iterates a torch DataLoader (baseline_cv.py:299,317); creating its iterator draws one
    # int64 "base seed" from the global generator before the first batch.  Drawn here too, so that a run
    # seeded like the reference consumes the identical noise stream.
    torch.empty((), dtype=torch.int64).random_()
    with torch.no_grad():
        for i in range(0, len(ins), batch_size):
            x = torch.stack([s["trajectory_embeddings"].transpose(0, 1) for s in ins[i:i + batch_size]])
            y = torch.stack([t.transpose(0, 1) for t in outs[i:i + batch_size]])
            ns = [s["norm_stat"] for s in ins[i:i + batch_size]]
            pred = model(x, num_candidates=num_candidates, noise_scale=noise_scale)
            a, f, r = candidate_metrics_cpu(pred, y, ns)
            tot[0] += a
            tot[1] += f
            tot[2] += r
            n += x.size(0)
    if n == 0:
        return 0.0, 0.0, 0.0, 0
    return tot[0] / n, tot[1] / n, tot[2] / n, n


def evaluate_model(model, batches, num_candidates=1, process_group=None, mc_dropout=False, reuse_prefix=False):
    """Test loop on the HIP path.  `batches` yields dicts in custom_collate_fn layout already on the GPU.
    K = 1: train.py:1274-1326; K > 1: test.py:1301-1382 (K forward passes per batch, min over K).
    mc_dropout=True runs the passes in train mode under no_grad, as test.py:1308-1309 does: the K candidates then
    differ through dropout (in-kernel Philox masks, one seed per pass).
    reuse_prefix=True: when the MLLM has no active dropout site (``model.mllm_is_deterministic()``) its
    pass -- 99 % of the work -- is identical for all K candidates of a batch, so it runs once and only the lane-polygon
    encoder and the LTSF run K times; results are identical to K full passes (dropout sites are numbered per module).
    Raises if the MLLM does have dropout: its candidates genuinely differ and must each be computed.
    Returns dict(ADE, FDE, RMSE, n) averaged over ALL ranks' samples."""
    from . import ops

    was_training = model.training
    lw = model.mllm.llama_wrapper
    was_saving, lw.save_for_backward = lw.save_for_backward, False  # (LoRA-trainable training keeps a decoder tape; evaluation needs none)
    model.train(bool(mc_dropout))
    dev = next(model.parameters()).device
    sums = torch.zeros(5, dtype=torch.float32, device=dev)
    n = 0
    try:
        with torch.no_grad():
            for b in batches:
                B = b["traj_emb"].shape[0]
                preds = []
                if reuse_prefix and not model.mllm_is_deterministic():
                    raise ValueError("reuse_prefix: the MLLM has active dropout sites (Q-Former / LoRA), its K passes differ")
                for kc in range(num_candidates):
                    out = model(b["traj_emb"], b["vision_emb"], None, b["lane_polygon"], b["lane_polygon_len"],
                                input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=None)
                    if reuse_prefix and kc == 0 and num_candidates > 1:
                        model._llm_cache = (model.last.final_hidden, model.last.final_hidden_bf16)
                    preds.append(out)
                model._llm_cache = None
                pred = torch.stack(preds, dim=1).contiguous()  # (B, K, 2, T_out)
                ns = b["norm_stat"]
                ns = ns if torch.is_tensor(ns) else torch.tensor([list(t) for t in ns], dtype=torch.float32)
                ns = ns.to(device=dev, dtype=torch.float32).contiguous()
                ops.traj_metrics(pred, b["target_traj"].contiguous(), ns, sums, None, None, B, num_candidates,
                                 pred.shape[-1])
                n += B
    finally:  # also on errors: leave the caller's train/eval mode and no stale MLLM cache behind
        model._llm_cache = None
        lw.save_for_backward = was_saving
        model.train(was_training)
    stats = torch.cat([sums[2:5].double(), torch.tensor([float(n)], dtype=torch.float64, device=dev)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(stats, group=process_group)
    ade, fde, rmse, tot = stats.tolist()
    model.mllm.check_flags()  # ids outside the vocabulary / masks that are not right-padded in ANY batch of the loop
    tot = max(tot, 1.0)
    return {"ADE": ade / tot, "FDE": fde / tot, "RMSE": rmse / tot, "n": int(tot)}
