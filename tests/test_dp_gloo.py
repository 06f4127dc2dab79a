"""Data-parallel path on CPU: 2 ranks over gloo (the N > 1 path of training.Trainer).

Kernels cannot run here, so the C library is stubbed (as in test_host_dryrun.py); what is checked is
the distributed logic itself: the flat gradient is exchanged in two buckets (LTSF first, lane-polygon
encoder second) with SUM semantics, every element exactly once, and the optimizer receives
grad_scale = 1/world (DDP's gradient averaging, reference train.py:1127-1132)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _StubLib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("tcavt_"):
            raise AttributeError(name)

        def fn(*args):
            self.calls.append((name, args))
            return 0

        return fn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, lora=False):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tcavt_amd import config, model, ops, training
        from tcavt_amd.weights import make_weights

        stub = _StubLib()
        ops.lib = lambda: stub
        ops.stream_ptr = lambda: None
        ops._ALLOW_CPU = True
        cfg = config.tiny()
        m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(make_weights(cfg, 0)).eval()
        # the LoRA-trainable loop clips (modify_train.py:1192); sync_initial_state exercises the construction broadcast
        if rank == 1:  # rank 1 starts from different weights: the broadcast must bring rank 0's over
            with torch.no_grad():
                m.ltsf.decoder.out_proj.weight.add_(1.0)
                m.mllm.q_proj.weight.add_(1.0)
        ref_w = (m.ltsf.decoder.out_proj.weight.detach().clone(), m.mllm.q_proj.weight.detach().clone())
        tr = training.Trainer(m, lora_trainable=lora, max_grad_norm=1.0 if lora else None)
        same = [torch.zeros_like(w) for w in ref_w]
        for w, s in zip((m.ltsf.decoder.out_proj.weight, m.mllm.q_proj.weight), same):
            s.copy_(w.detach())
            dist.broadcast(s, src=0)
        ok_bcast = all(torch.equal(s, w.detach()) for s, w in zip(same, (m.ltsf.decoder.out_proj.weight, m.mllm.q_proj.weight)))
        assert tr.world == world
        n = tr.book.total
        assert (tr.n_base < n) if lora else (tr.n_base == n)
        base = torch.arange(n, dtype=torch.float32) % 97
        tr.book.grads.copy_(base * (rank + 1))
        tr._allreduce_bucket(0, tr.n_ltsf)
        tr._allreduce_bucket(tr.n_ltsf, tr.n_base)
        if lora:  # third bucket: the adapter gradients, ready last (modify_scripts/modify_train.py:512-528)
            tr._allreduce_bucket(tr.n_base, n)
        expect = base * sum(r + 1 for r in range(world))
        ok_sum = torch.equal(tr.book.grads, expect)
        tr._last_loss = torch.zeros(())
        tr.optimizer_step()
        if lora:
            # clipping applies to the MEAN gradient: the SUM buckets are scaled by 1/world inside the clip call, the
            # gated optimizer (finite-loss test of modify_train.py:1190-1196) then takes grad_scale = 1
            _, cargs = [c for c in stub.calls if c[0] == "tcavt_clip_grad_norm"][-1]
            ok_clip = abs(cargs[2] - 1.0) < 1e-9 and abs(cargs[3] - 1.0 / world) < 1e-9
            name, args = [c for c in stub.calls if c[0] == "tcavt_adamw_gated"][-1]
            ok_clip = ok_clip and abs(args[10] - 1.0) < 1e-9 and not any(c[0] == "tcavt_adamw" for c in stub.calls)
            grad_scale = 1.0 / world
        else:
            ok_clip = not any(c[0] == "tcavt_clip_grad_norm" for c in stub.calls)
            name, args = [c for c in stub.calls if c[0] == "tcavt_adamw"][-1]
            grad_scale = args[-2]
        # trainable set == everything outside mllm (train.py:1140-1145), flat order: ltsf then polygon encoder
        names = tr.book.names
        ok_names = all(k.startswith("ltsf.") for k in names[: sum(k.startswith("ltsf.") for k in names)]) and \
            set(names) == {k for k, _ in m.named_parameters() if not k.startswith("mllm.") or (lora and ".lora_" in k)}
        if lora:  # adapters sit behind the train.py set, and only they have requires_grad inside the MLLM
            first_lora = min(i for i, k in enumerate(names) if ".lora_" in k)
            ok_names = ok_names and all(".lora_" in k for k in names[first_lora:]) and \
                all(p.requires_grad == (".lora_" in k) for k, p in m.mllm.named_parameters())
        q.put((rank, ok_sum, abs(grad_scale - 1.0 / world) < 1e-9, ok_names, 0 < tr.n_ltsf < n, ok_clip, ok_bcast))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise


@pytest.mark.timeout(300)
@pytest.mark.parametrize("lora", [False, True])
def test_two_rank_gradient_exchange_gloo(lora):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, lora)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    for r in results:
        assert len(r) == 7, r
        assert all(r[1:]), r
