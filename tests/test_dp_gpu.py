"""Data-parallel training step with REAL kernels: two ranks share the one card and exchange gradients over gloo (RCCL refuses
two ranks on one device, so the collective library differs from production; everything else -- Trainer's bucket hand-off from
the backward's streams, SUM semantics, 1 / world in the optimizer, the reduced stream layout of a data-parallel rank -- is the
N > 1 path bench.py runs under torch.distributed.run).  What DistributedDataParallel guarantees the reference
(scripts/train.py:1127-1132) is checked end to end: both ranks hold the same summed gradient and the same parameters after
the step, and they equal one process stepping on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests.util import batch_tensors, load_case, rel_err

pytestmark = pytest.mark.gpu

ARGS = ("traj_emb", "vision_emb", "lane_polygon", "lane_polygon_len", "target_traj", "norm_stat", "input_ids", "attention_mask",
        "labels")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step(case, rows, dev):
    """One Trainer step on the given batch rows: (summed gradient, parameters after the step, loss)."""
    from tcavt_amd import model, training

    cfg, weights, fx = load_case(case)
    t = batch_tensors(fx)
    g = {k: v[rows].contiguous().to(dev) for k, v in t.items()}
    m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
    tr = training.Trainer(m, lr=1e-3)
    loss, _ = tr.forward_backward(*[g[k] for k in ARGS])
    torch.cuda.synchronize()
    grads = tr.book.grads.detach().clone()
    tr.optimizer_step()
    torch.cuda.synchronize()
    return tr, grads.cpu(), tr.book.params.detach().clone().cpu(), float(loss.item())


def _worker(rank, world, port, case, outdir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        tr, grads, params, loss = _step(case, slice(rank, rank + 1), torch.device("cuda", 0))
        assert tr.world == world
        # self-diagnosis of a data-parallel rank (bench.py's dp_diagnostics): per-bucket all-reduce timings + stream budget
        cfg, weights, fx = load_case(case)
        g = {k: v[rank:rank + 1].contiguous().to("cuda:0") for k, v in batch_tensors(fx).items()}
        tr.enable_diagnostics(True)
        tr.forward_backward(*[g[k] for k in ARGS])
        torch.cuda.synchronize()
        d = tr.diagnostics()
        tr.enable_diagnostics(False)
        assert sum(b["bytes"] for b in d["buckets"]) == 4 * tr.book.total and all(b["launches"] == 1 and b["mean_ms"] >= 0 for b in d["buckets"])
        assert d["hip_streams_in_use"] <= 5, d  # the five-stream budget of the pipelined step
        before = tr.book.params.detach().clone()
        with tr.local_steps():  # the same-build single-rank comparison leg: gradients stay local ...
            tr.forward_backward(*[g[k] for k in ARGS])
            torch.cuda.synchronize()
            local = tr.book.grads.detach().clone().cpu()
            tr.optimizer_step()  # ... the replicas drift apart ...
            torch.cuda.synchronize()
            drift = tr.book.params.detach().clone()
        torch.cuda.synchronize()
        after = tr.book.params.detach().cpu()  # ... and are one state again on exit (rank 0's); (gloo gathers CPU tensors)
        both = [torch.zeros_like(after) for _ in range(world)]
        dist.all_gather(both, after)
        assert all(torch.equal(b, both[0]) for b in both) and not torch.equal(drift, before)
        torch.save({"grads": grads, "params": params, "loss": loss, "diag": d, "local": local}, os.path.join(outdir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case", ["tiny_18_30_nolora_ragged", "tiny_6_12_lora_ragged"])
def test_two_ranks_equal_one_process_on_the_concatenated_batch(gpu, case, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    # the exchange: every rank holds the same SUM, and after the step the same parameters
    assert torch.equal(r0["grads"], r1["grads"])
    assert torch.equal(r0["params"], r1["params"])
    assert r0["loss"] != r1["loss"]  # (they did see different samples)
    assert not torch.equal(r0["local"], r1["local"])  # exchange switched off: every rank keeps its own gradient
    assert [b["bytes"] for b in r0["diag"]["buckets"]] == [b["bytes"] for b in r1["diag"]["buckets"]]
    # ... and that is what one process computes on both samples (the loss is a mean over the batch: MEAN of the ranks' gradients)
    tr, g_full, p_full, loss_full = _step(case, slice(0, world), gpu["device"])
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - loss_full) / loss_full < 1e-4
    g_dp = r0["grads"] / world
    worst = 0.0
    for name in tr.book.names:
        o, n, _ = tr.book.offsets[name]
        a, b = g_dp[o:o + n], g_full[o:o + n]
        if b.abs().max() == 0:
            assert a.abs().max() == 0, name
            continue
        worst = max(worst, rel_err(a, b))
    flat = ((g_dp - g_full).double().norm() / g_full.double().norm()).item()
    print(f"[dp] gradients: flat rel {flat:.2e}, worst tensor {worst:.2e}")
    assert flat < 1e-4 and worst < 1e-3  # (measured 3e-6 / 5e-5: only the summation order over the batch differs)
    moved = (p_full - _initial_params(case, tr)).double().norm()
    apart = (r0["params"] - p_full).double().norm()
    print(f"[dp] parameters moved {moved:.3e}, two ranks vs one process {apart:.3e}")
    assert apart < 0.05 * moved


def _initial_params(case, tr):
    """The flat trainable vector before any step, in the trainer's layout."""
    _, weights, _ = load_case(case)
    out = torch.zeros(tr.book.total)
    for name in tr.book.names:
        o, n, shape = tr.book.offsets[name]
        out[o:o + n] = torch.from_numpy(weights[name]).reshape(-1)
    return out


def _rccl_worker(port, case, outdir):
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # (1) the plain one-process step
    _, grads0, params0, loss0 = _step(case, slice(0, 2), dev)
    # (2) the same step through a ONE-rank RCCL process group taking the data-parallel path (TCAVT_FORCE_DP)
    os.environ["TCAVT_FORCE_DP"] = "1"
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        tr, grads1, params1, loss1 = _step(case, slice(0, 2), dev)
        assert tr._force_dp and tr.world == 1
        cfg, weights, fx = load_case(case)
        g = {k: v[0:2].contiguous().to(dev) for k, v in batch_tensors(fx).items()}
        tr.enable_diagnostics(True)
        tr.forward_backward(*[g[k] for k in ARGS])
        torch.cuda.synchronize()
        d = tr.diagnostics()
        tr.enable_diagnostics(False)
        torch.save({"g0": grads0, "g1": grads1, "p0": params0, "p1": params1, "l0": loss0, "l1": loss1, "diag": d},
                   os.path.join(outdir, "rccl.pt"))
    finally:
        dist.destroy_process_group()
        os.environ.pop("TCAVT_FORCE_DP", None)


@pytest.mark.timeout(600)
def test_one_rank_rccl_group_takes_the_data_parallel_path(gpu, tmp_path):
    """The collective library of production on the one card of the box: a one-rank RCCL (backend "nccl") process group with
    TCAVT_FORCE_DP=1 makes Trainer launch the real per-bucket all-reduce on the process group's stream, with the reduced stream
    layout of a data-parallel rank.  SUM over one rank is the identity: gradients, parameters and loss must equal the plain step,
    every bucket must have been launched, and the rank stays within the five-stream budget."""
    case = "tiny_6_12_lora_ragged"
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), case, str(tmp_path)))
    p.start()
    p.join(500)
    assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    r = torch.load(os.path.join(str(tmp_path), "rccl.pt"))
    # (equal up to the order of the backward's atomically accumulated column sums, which differs from run to run)
    # (parameters: AdamW's first step is lr * g / |g| per element -- where a gradient is a rounding away from zero its sign, and so
    #  2 lr of that parameter, follows the summation order)
    assert rel_err(r["g1"], r["g0"]) < 1e-5 and rel_err(r["p1"], r["p0"]) < 1e-3 and abs(r["l0"] - r["l1"]) <= 1e-6 * abs(r["l0"])
    d = r["diag"]
    assert len(d["buckets"]) >= 1 and all(b["launches"] == 1 for b in d["buckets"]) and d["hip_streams_in_use"] <= 5, d
