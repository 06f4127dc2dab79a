#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec of the multimodal-LLM trajectory-prediction hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`,
     one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment)

Workload (BASELINE.json configs[1], "train.py multimodal-LLM, 1xMI355X bf16, batch 32, LoRA rank
8, seq_len 256"): one step = one pass of MultiModalTrajectoryModel.forward (reference
scripts/train.py:914-964, loss included) over one batch of 32 synthetic samples per GPU with
fused LLM length L = 16 query tokens + 240 text tokens = 256, T_in/T_out = 18/30, Llama-3.2-1B
shape (16 layers, hidden 2048, 32/8 heads x 64, MLP 8192, vocab 128256), LoRA r=8 alpha=32 on
q_proj/v_proj, random-init weights, inputs resident in HBM before the timed region.
Data parallel: every rank runs the same step on its own shard of the global batch (weak scaling);
the forward has no data-path collective.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     -- dominant kernel (gate|up projection GEMM with SiLU*up epilogue), timed in situ with
                  HIP events on the launching stream during the timed steps
  cpu_baseline -- the CPU oracle (oracle/forward.py, "port") on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (the pool's host driver supports dmabuf IPC only: without this RCCL's peer mappings fail in hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0
RATED_SCLK_MHZ = 2400.0  # MI355X_MICROARCH.md "Max clock"


class CardTelemetry:
    """Shader clock and socket power of THIS process's card, polled from its hwmon files in sysfs by a background thread
    (the host has several cards: matched by PCI address).  Used outside the timed region only: the dominant GEMM runs at the
    power cap on real data and the chip lowers its clock (MI355X_MICROARCH.md "DVFS give-back"), so the roofline fraction is
    also quoted against the MFMA peak at the clock the chip actually sustained.  Unreadable files -> no figures (None)."""

    def __init__(self, dev_index):
        import glob
        import threading

        self.rows, self._stop, self._thr, self.files = [], False, None, None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
                if os.path.basename(os.path.realpath(card)) != want:
                    continue
                hw = sorted(glob.glob(card + "/hwmon/hwmon*"))
                if hw and os.path.exists(hw[0] + "/freq1_input"):
                    pw = hw[0] + ("/power1_input" if os.path.exists(hw[0] + "/power1_input") else "/power1_average")
                    self.files = (hw[0] + "/freq1_input", pw, hw[0] + "/power1_cap")
        except Exception:
            self.files = None
        self._threading = threading

    @staticmethod
    def _num(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop:
            self.rows.append((self._num(self.files[0]), self._num(self.files[1])))
            time.sleep(0.01)

    def start(self):
        if self.files:
            self._stop, self.rows = False, []
            self._thr = self._threading.Thread(target=self._run, daemon=True)
            self._thr.start()

    def stop(self):
        if not self._thr:
            return None
        self._stop = True
        self._thr.join()
        rows = self.rows[len(self.rows) // 4:]  # (skip the ramp)
        ck = [r[0] / 1e6 for r in rows if r[0]]
        pw = [r[1] / 1e6 for r in rows if r[1]]
        cap = self._num(self.files[2])
        if not ck:
            return None
        return {"sclk_mhz": round(sum(ck) / len(ck), 0), "rated_sclk_mhz": RATED_SCLK_MHZ,
                "power_w": round(sum(pw) / len(pw), 0) if pw else None, "power_cap_w": round(cap / 1e6, 0) if cap else None,
                "samples": len(ck)}
GFLOP_PER_SAMPLE = 509.8  # forward, L=256, 18->30


def gflop_per_sample(cfg, L):
    """Forward GFLOP per sample at fused length L (which is for L = 256, scaled by shape): dense
    decoder GEMMs and the cross-attention K/V projections are linear in L, causal attention goes with L (L + 1) / 2."""
    ll = cfg.llama
    per_tok = 2.0 * ll.layers * (ll.hidden * (ll.n_q_heads + 2 * ll.n_kv_heads) * ll.head_dim + ll.n_q_heads * ll.head_dim * ll.hidden
                                 + 3 * ll.hidden * ll.inter) / 1e9
    lora = (2.0 * ll.layers * cfg.lora_r * (2 * ll.hidden + (ll.n_q_heads + ll.n_kv_heads) * ll.head_dim) / 1e9) if cfg.use_lora else 0.0
    attn = 4.0 * ll.layers * ll.n_q_heads * ll.head_dim * (L * (L + 1) / 2) / 1e9
    kv_proj = 2.0 * 2 * ll.hidden * ll.hidden / 1e9  # cross-attention K and V in-projections, per token
    return (per_tok + lora + kv_proj) * L + attn + 1.84 + 0.05 + 0.58 + 0.07  # + Q-Former, q_proj, LTSF rest, polygon
ATTN_MB_PER_SAMPLE = 41.9  # q,k,v in + o out, 16 layers, bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--text-len", type=int, default=240)
    ap.add_argument("--seq-len", type=int, default=18)
    ap.add_argument("--out-len", type=int, default=30)
    ap.add_argument("--preset", default="llama32_1b")
    ap.add_argument("--no-lora", action="store_true")
    ap.add_argument("--storage", choices=["fp16", "bf16"], default="fp16",
                    help="16-bit storage type of GEMM operands (weights copies and activations); fp32 accumulation either "
                         "way.  fp16 (default) keeps the whole model within 1e-3 of the fp32 reference; bf16 is round 1's "
                         "contract (the LoRA-trainable variant follows the same switch: fp16 tapes and device-scaled fp16 "
                         "gradients by default)")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8,
                    help="samples in the CPU-baseline pass (SURVEY 8d asks for 32 or the largest that fits; 8 keeps the default "
                         "run within a few minutes -- the B=32 figure is recorded in DESIGN.md)")
    ap.add_argument("--cpu-b32", choices=["auto", "off"], default="auto",
                    help="auto (default): after the --cpu-batch protocol, ONE warm and ONE timed same-work pass on the 32 samples "
                         "of the survey's protocol (SURVEY 8d) if the box's budget allows (~45 s); the line's cpu_baseline.value is "
                         "then the B = 32 figure and says so, the B = 8 protocol figure stays next to it")
    ap.add_argument("--cpu-warm", type=int, default=3)
    ap.add_argument("--cpu-timed", type=int, default=5)
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: exercises the rank plumbing only (process group, barrier, max-over-ranks, JSON line); "
                         "use with --backend gloo on a CPU box")
    ap.add_argument("--launch", choices=["eager", "graph"], default="eager",
                    help="eager (default): the host enqueues every step; it runs a whole decoder (15 ms) ahead of the "
                         "GPU, and the side streams of the forward / backward overlap as written.  graph: replay of a "
                         "hipGraph of the step; measured slower (18.85 vs 18.4 ms) because the graph executor maps the "
                         "captured side-stream branches onto its own queues and serialises independent chains")
    ap.add_argument("--lora-trainable", action="store_true",
                    help="the LoRA-trainable variant (modify_scripts/modify_train.py:512-528,1192; SURVEY 8f.1): adapters of "
                         "q_proj / v_proj train too, the backward walks through the frozen decoder layers, grad clip 1.0; "
                         "not the headline configuration")
    ap.add_argument("--train-mllm-front", action="store_true",
                    help="with --lora-trainable: the WHOLE trainable set of modify_scripts/modify_train.py -- the Q-Former, "
                         "mllm.q_proj and the modality embeddings train too (backward continues below decoder layer 0)")
    ap.add_argument("--no-dropout", action="store_true",
                    help="train mode only: run the step with model.eval() arithmetic (dropout = identity) instead of "
                         "ddp_model.train() (train.py:1152; dropout 0.1 in the lane-polygon encoder, Q-Former, LoRA branch, LTSF)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not start the (frozen) Q-Former of the next batch underneath the step in flight")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="train.py variant: do not run the frozen MLLM pass on a stream of its own (the decoder of step i+1 "
                         "then waits for step i's backward and optimizer instead of running over them)")
    ap.add_argument("--no-graph", action="store_true", help="(kept for old command lines; same as --launch eager)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--rccl-self-test", action="store_true",
                    help="--gpus 1 only: a one-rank RCCL process group takes Trainer's data-parallel path (real all-reduce per "
                         "bucket, five-stream layout); adds dp_diagnostics to the line")
    ap.add_argument("--feed", choices=["resident", "host"], default="resident",
                    help="resident (default, the driver's contract: inputs in HBM before the timed region): every step runs on one "
                         "resident batch.  host: every step collates a fresh batch on the host (data.custom_collate_fn over "
                         "per-sample dicts, four synthetic batches in rotation), stages it in pinned memory and uploads it with "
                         "non-blocking copies on a side stream (data.DeviceFeeder; scripts/train.py:1153-1166 does seven blocking "
                         ".to(device) calls); the step starts on the copy's event")
    ap.add_argument("--mode", choices=["train", "forward"], default="train",
                    help="train: zero_grad + forward + backward + grad all-reduce + AdamW (train.py:1168-1183); "
                         "forward: MultiModalTrajectoryModel.forward incl. loss only (test.py / validation)")
    return ap.parse_args()


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually use (affinity / cgroup share), capped at the box's 16-core share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("TCAVT_CPU_THREADS", "16"))))


def parity_batch(cfg, args):
    from tcavt_amd import synth

    b = synth.make_batch(cfg, args.cpu_batch, text_len=args.text_len, seed=1, ragged=True,
                         min_text=128 if args.text_len > 128 else max(1, args.text_len // 2))
    return {k: torch.from_numpy(v) for k, v in b.items()}


def cpu_baseline(cfg, args, gpu_decoded=None, W=None):
    """Oracle ("port") on the host cores: B=cpu_batch samples of the same workload, fp32.  With `gpu_model` (same seeded
    weights) the HIP path's eval-mode results on those samples are also CHECKED against the oracle's at the full model
    size: relative error of the decoded trajectories and of ADE / FDE -> "parity_full_size"."""
    from oracle import forward as O
    from tcavt_amd import synth
    from tcavt_amd.weights import make_weights

    cores = host_cores()
    torch.set_num_threads(cores)
    if W is None:
        log(f"cpu_baseline: generating fp32 weights on the host ({cores} threads)")
        W = make_weights(cfg, seed=1, backend="torch", device="cpu")
    log("cpu_baseline: timing the oracle")
    t = parity_batch(cfg, args)

    train = args.mode == "train"
    if train:  # autograd over exactly the parameters train.py trains (everything outside mllm)
        for k, v in W.items():
            if not k.startswith("mllm."):
                v.requires_grad_(True)

    def run(labels):
        with torch.set_grad_enabled(train):
            loss, dec = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"],
                                        t["lane_polygon_len"], t["input_ids"], t["attention_mask"], y=t["target_traj"],
                                        norm_stat=t["norm_stat"], contract="fp32", labels=labels)
            if train:
                loss.backward()
                for k, v in W.items():
                    v.grad = None
            return loss

    def timed(labels, n):
        for _ in range(max(1, args.cpu_warm)):
            run(labels)  # warm
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            run(labels)
            ts.append(time.perf_counter() - t0)
            log(f"cpu_baseline: {'faithful' if labels is not None else 'same-work'} pass {ts[-1]:.2f} s")
        ts.sort()
        return ts[len(ts) // 2]

    same_work = timed(None, max(1, args.cpu_timed))
    faithful = timed(t["labels"], max(1, args.cpu_timed))
    # SURVEY 8d's batch (32) when the budget allows: one warm + one timed same-work pass, only if the B = cpu_batch passes say
    # that the two together stay under ~60 s
    b32 = None
    if args.cpu_b32 == "auto" and args.cpu_batch < 32 and same_work * (32.0 / args.cpu_batch) * 2 < 60.0:
        t_small = t
        a32 = argparse.Namespace(**dict(vars(args), cpu_batch=32))
        t = parity_batch(cfg, a32)
        run(None)
        t0 = time.perf_counter()
        run(None)
        b32 = time.perf_counter() - t0
        log(f"cpu_baseline: B = 32 same-work pass {b32:.2f} s")
        t = t_small
    parity = None
    if gpu_decoded is not None:
        with torch.no_grad():
            _, dec_o = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                       t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                       contract="fp32")
            _, dec_c = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                       t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                       contract=args.storage)
        dec_g = gpu_decoded
        mc = O.traj_metrics(dec_c, t["target_traj"], t["norm_stat"])
        mo = O.traj_metrics(dec_o, t["target_traj"], t["norm_stat"])
        mg = O.traj_metrics(dec_g, t["target_traj"], t["norm_stat"])
        parity = {
            "decoded_rel_err": round(((dec_g - dec_o).norm() / dec_o.norm()).item(), 6),
            "ade_rel_diff": round(abs(mg["ade_sum"] - mo["ade_sum"]) / mo["ade_sum"], 6),
            "fde_rel_diff": round(abs(mg["fde_sum"] - mo["fde_sum"]) / mo["fde_sum"], 6),
            "contract_own_decoded_rel_err": round(((dec_c - dec_o).norm() / dec_o.norm()).item(), 6),
            "decoded_rel_err_vs_contract": round(((dec_g - dec_c).norm() / dec_c.norm()).item(), 6),
            "ade_rel_diff_vs_contract": round(abs(mg["ade_sum"] - mc["ade_sum"]) / mc["ade_sum"], 6),
            "fde_rel_diff_vs_contract": round(abs(mg["fde_sum"] - mc["fde_sum"]) / mc["fde_sum"], 6),
            "contract": args.storage,
            "note": f"HIP path ({args.storage} operands, eval arithmetic) vs the oracle on the cpu_baseline samples at the "
                    f"full model size; first three fields against the fp32 oracle, then the {args.storage}-contract oracle's "
                    "own distance from fp32 and the HIP path's distance from that contract",
        }
        log(f"full-size parity: {parity}")
    return {
        "parity_full_size": parity,
        "value": round((32 / b32) if b32 else (args.cpu_batch / same_work), 4), "unit": "trajectories/sec", "cores": cores, "kind": "port",
        "sample": (f"32 samples of the same workload (SURVEY 8d's batch; L={16 + args.text_len}, fp32, torch CPU ops"
                   f"{', forward + autograd backward of the trainable part' if train else ''}), one timed pass after one warm-up pass; "
                   "same work as the GPU path (no lm_head/CE, no optimizer step)" if b32 else
                   f"{args.cpu_batch} samples of the same workload (L={16 + args.text_len}, fp32, torch CPU ops"
                   f"{', forward + autograd backward of the trainable part' if train else ''}), "
                   f"median of {max(1, args.cpu_timed)} timed passes after {max(1, args.cpu_warm)} warm-up passes (SURVEY 8d protocol); "
                   f"same work as the GPU path (no lm_head/CE, no optimizer step)"),
        "batch": 32 if b32 else args.cpu_batch,
        "value_small_batch_protocol": {"batch": args.cpu_batch, "value": round(args.cpu_batch / same_work, 4),
                                       "protocol": f"median of {max(1, args.cpu_timed)} timed passes after {max(1, args.cpu_warm)} warm-up passes"},
        "reference_faithful_value": round(args.cpu_batch / faithful, 4),
        "reference_faithful_note": f"B = {args.cpu_batch}; adds the lm_head + cross-entropy the reference computes and discards (train.py:547-554)",
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a torchrun environment: start N fresh ranks (one process per GPU, reference
    scripts/train.py:1044-1049 `mp.spawn(train_ddp, nprocs=world_size)`) as children of this process, which has not
    touched the GPU, and pass rank 0's JSON line through.  Never re-exec: the children are new interpreters."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"spawning {args.gpus} ranks: {' '.join(cmd)}")
    raise SystemExit(subprocess.run(cmd).returncode)


def dry_run(args, rank, world):
    """Rank plumbing without a GPU: barrier, a timed region, MAX over ranks, one JSON line from rank 0."""
    dist.barrier() if world > 1 else None
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "trajectories/sec", "value": None, "unit": "trajectories/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": "weak",
                          "max_rank_elapsed_s": round(el.item(), 4),
                          "config": {"per_gpu_batch": args.batch, "global_batch": world * args.batch,
                                     "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    elif args.rccl_self_test:
        # one rank, real RCCL: the data-parallel path of Trainer (bucketed all-reduce on the process group's own stream, the
        # reduced stream layout) on the one GPU of a box -- RCCL refuses two ranks on one device, so this is as close as a
        # single card gets to the N > 1 run; the line carries dp_diagnostics like an N > 1 line
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ["TCAVT_FORCE_DP"] = "1"
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    dp = world > 1 or args.rccl_self_test
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPUs visible")
    local_dev = local_rank % ndev  # (gloo rehearsal: several ranks may share a card)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    from tcavt_amd import capi, config, model, synth, training
    from tcavt_amd.weights import make_weights

    capi.init(local_dev)
    cfg = config.PRESETS[args.preset](seq_len=args.seq_len, out_len=args.out_len, use_lora=not args.no_lora)
    B, L = args.batch, cfg.q_num_query_tokens + args.text_len

    t0 = time.time()
    log(f"building {args.preset} model on {dev}")
    with torch.device(dev):
        m = model.MultiModalTrajectoryModel.from_config(cfg)
    # With the CPU baseline / full-size parity leg the weights are drawn ONCE on the host and copied to the GPU (torch's
    # CPU and GPU generators give different streams for one seed); otherwise they are drawn on the GPU directly.
    W_cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        log("generating fp32 weights on the host (shared by the GPU model and the CPU oracle)")
        torch.set_num_threads(host_cores())
        W_cpu = make_weights(cfg, seed=1, backend="torch", device="cpu")
        m.load_weights(W_cpu)
    else:
        W = make_weights(cfg, seed=1, backend="torch", device=dev)
        m.load_weights(W)
        del W
    # (a captured graph would replay ONE set of dropout masks: seeds are kernel arguments -> graph mode runs eval arithmetic)
    m.set_storage(torch.float16 if args.storage == "fp16" else torch.bfloat16)
    dropout_on = args.mode == "train" and not args.no_dropout and args.launch != "graph"
    m.train(dropout_on)
    m.mllm.llama_wrapper.gemm_tile = args.tile
    torch.cuda.empty_cache()

    # rank r works on its own shard of the global batch (seeded by rank): weak scaling
    b = synth.make_batch(cfg, B, text_len=args.text_len, seed=100 + rank, ragged=True, min_text=128 if args.text_len > 128 else max(1, args.text_len // 2))
    g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}

    # full-size parity sample (checked against the oracle in the cpu_baseline leg): eval-mode output of the freshly seeded
    # model on the cpu_baseline samples, before any optimizer step changes the weights
    parity_dec = None
    if not args.no_cpu_baseline and rank == 0:
        pt = {k: v.to(dev) for k, v in parity_batch(cfg, args).items()}
        was = m.training
        m.eval()
        with torch.no_grad():
            parity_dec = m(pt["traj_emb"], pt["vision_emb"], None, pt["lane_polygon"], pt["lane_polygon_len"],
                           input_ids=pt["input_ids"], attention_mask=pt["attention_mask"]).float().cpu()
        m.train(was)
        del pt

    trainer = None
    if args.mode == "train":
        trainer = training.Trainer(m, lr=5e-4, weight_decay=1e-4, lora_trainable=args.lora_trainable,
                                   max_grad_norm=1.0 if args.lora_trainable else None,
                                   train_mllm_front=args.lora_trainable and args.train_mllm_front)
        if args.no_pipeline:
            m.pipeline_decoder = False

    # --feed host: a fresh batch per step, collated on the host and uploaded under the step in flight
    feeder, fed = None, {"i": 0, "cur": None}
    if args.feed == "host":
        from tcavt_amd import data as tdata

        # The host side of a fed step is a few small torch CPU ops (stack, pad_sequence, staging copies).  Left at its default,
        # torch's intra-op pool starts one thread per CPU the box SHOWS (not the share this process may use), they spin, the
        # cgroup's CPU quota runs out and the whole process is frozen until the next scheduler period: 90 ms stalls every few
        # steps, 31-42 instead of 15 ms per step (tools/feed_probe.py; profiles/r04_feed_host.txt).  Two threads are plenty.
        torch.set_num_threads(2)

        n_sets = 4
        host_sets = [synth.batch_to_samples(synth.make_batch(cfg, B, text_len=args.text_len, seed=100 + rank + 1000 * (j + 1), ragged=True,
                                                             min_text=128 if args.text_len > 128 else max(1, args.text_len // 2)))
                     for j in range(n_sets)]
        feeder = tdata.DeviceFeeder(dev)

        def fetch():
            j = fed["i"] % n_sets
            fed["i"] += 1
            return feeder.put(tdata.custom_collate_fn(host_sets[j]))  # (host work of the step: collate + pinned staging)

    def step(next_vision=None):
        if feeder is not None:
            cur = fed["cur"] or fetch()
            nxt = fetch()  # batch i + 1 is on its way before step i is enqueued
            if trainer is not None:
                out = trainer.step(cur["traj_emb"], cur["vision_emb"], cur["lane_polygon"], cur["lane_polygon_len"],
                                   cur["target_traj"], cur["norm_stat"], cur["input_ids"], cur["attention_mask"], cur["labels"],
                                   next_vision_embs=nxt["vision_emb"] if next_vision is not None else None, next_ready=nxt.ready,
                                   inputs_ready=cur.ready)
            else:
                torch.cuda.current_stream().wait_event(cur.ready)
                out = m(cur["traj_emb"], cur["vision_emb"], None, cur["lane_polygon"], cur["lane_polygon_len"], y=cur["target_traj"],
                        norm_stat=cur["norm_stat"], input_ids=cur["input_ids"], attention_mask=cur["attention_mask"],
                        labels=cur["labels"])
                if next_vision is not None:
                    m.prefetch(nxt["vision_emb"], ready=nxt.ready)
            feeder.release(cur)
            fed["cur"] = nxt
            return out
        if trainer is not None:
            return trainer.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"],
                                g["target_traj"], g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"],
                                next_vision_embs=next_vision, inputs_ready=True if m.pipeline_decoder else None)  # (the
            # synthetic batch is resident: nothing writes the MLLM inputs between steps)
        return m(g["traj_emb"], g["vision_emb"], None, g["lane_polygon"], g["lane_polygon_len"], y=g["target_traj"],
                 norm_stat=g["norm_stat"], input_ids=g["input_ids"], attention_mask=g["attention_mask"],
                 labels=g["labels"])

    with torch.no_grad():
        loss, decoded = step()  # builds packed weights + workspaces
        torch.cuda.synchronize()
        m.mllm.check_flags()
        if not torch.isfinite(loss).item():
            raise SystemExit("non-finite loss in warm-up")
        setup_s = time.time() - t0
        first_loss = float(loss.item())
        log(f"setup + first step done in {setup_s:.1f} s; loss {first_loss:.3f}")

        graph = None
        if args.launch == "graph" and feeder is not None:
            raise SystemExit("--feed host goes with --launch eager (a captured step replays on fixed input tensors)")
        if args.launch == "graph" and not args.no_graph and world == 1:  # multi-rank: RCCL all-reduces are launched eagerly
            # hipGraph of the whole step (all launches are stream-ordered and allocation-free after the first call).  In train
            # mode Trainer.capture moves the per-step host state to the device (optimizer step count, dropout epoch): every
            # replay is a full train.py step with fresh dropout masks.
            if trainer is not None:
                graph, (g_loss, g_decoded) = trainer.capture(
                    g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
                    g["input_ids"], g["attention_mask"], g["labels"])
            else:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    step()
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    g_loss, g_decoded = step()

        prefetch = graph is None and not args.no_prefetch

        last = {"loss": loss}

        def run_step():
            if graph is not None:
                graph.replay()
                last["loss"] = g_loss
            else:
                if trainer is not None or feeder is not None:  # batch i+1's Q-Former goes to a side stream and runs under step i's decoder
                    last["loss"] = step(g["vision_emb"] if prefetch else None)[0]
                else:
                    last["loss"] = step()[0]
                    if prefetch:
                        m.prefetch(g["vision_emb"])

        log("graph captured" if graph is not None else "eager mode")
        for _ in range(args.warmup):
            run_step()
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            run_step()
        enqueue_s = time.perf_counter() - t_start  # host time to enqueue the K steps (eager: must stay below the GPU time)
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        log(f"host enqueue time {enqueue_s / args.steps * 1e3:.3f} ms/step")
        log(f"timed {args.steps} steps: {elapsed / args.steps * 1e3:.3f} ms/step")
        loss_timed = float(last["loss"].item())  # (read here: the un-timed passes below keep training)

        # in-situ kernel timing: tcavt_llama_stack_forward records HIP events on the launching stream around the five big
        # kernels of every layer (ops.StackEvents); one pass = 16 launches of each, averaged over a few passes
        from tcavt_amd import ops
        timer = ops.StackEvents(cfg.llama.layers)
        m.mllm.llama_wrapper.timer = timer
        acc = {}
        n_pass = max(2, min(5, args.steps))
        for _ in range(n_pass):
            step()
            torch.cuda.synchronize()
            for k, (cnt, ms) in timer.summary().items():
                c0, t0_ = acc.get(k, (0, 0.0))
                acc[k] = (c0 + cnt, t0_ + cnt * ms)
        m.mllm.llama_wrapper.timer = None
        timer.close()
        ksum = {k: (c, t / c) for k, (c, t) in acc.items()}
        # sustained clock / power of the same step loop (un-timed repeat, sampled from sysfs)
        tele = CardTelemetry(local_dev)
        tele.start()
        for _ in range(max(20, min(args.steps, 60))):
            run_step()
        torch.cuda.synchronize()
        clock = tele.stop()

        # ---- data-parallel self-diagnosis (world > 1; un-timed repeats of the same steps): what the first multi-GPU run needs
        # to check the overlap claims of DESIGN section 6 instead of arguing them -- (i) the same build's step time with the
        # gradient exchange switched off (the N = 1 figure on this very card, all ranks running it at once), (ii) how long
        # every bucket's all-reduce held its launching stream, (iii) busy / idle time of the MLLM stream per step, (iv) the
        # number of HIP streams a rank uses (the five-stream budget)
        dp_diag = None
        if dp and trainer is not None and graph is None:
            n_diag = max(5, min(args.steps, 20))

            def timed_loop(n):
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for _ in range(n):
                    run_step()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                return (time.perf_counter() - t_) / n * 1e3

            with trainer.local_steps():  # (re-synchronises the replicas from rank 0 on exit: they drift apart inside)
                run_step()
                ms_no_exchange = timed_loop(n_diag)
            run_step()
            trainer.enable_diagnostics(True)
            m.pipe_trace = [] if m.pipeline_decoder else None
            ms_diag = timed_loop(n_diag)
            d = trainer.diagnostics()
            trainer.enable_diagnostics(False)
            ev, m.pipe_trace = m.pipe_trace, None
            pipe = None
            if ev:
                busy = [a.elapsed_time(b_) for a, b_ in ev]
                idle = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)] or [0.0]
                pipe = {"mllm_pass_busy_ms": round(sum(busy) / len(busy), 3), "mllm_idle_between_passes_ms": round(sum(idle) / len(idle), 3),
                        "mllm_idle_max_ms": round(max(idle), 3), "passes": len(busy)}
            loc = torch.tensor([ms_no_exchange, ms_diag], dtype=torch.float64, device=dev)
            dist.all_reduce(loc, op=dist.ReduceOp.MAX)
            dp_diag = dict(d, ms_per_step_exchange_off=round(loc[0].item(), 3), ms_per_step_with_event_timing=round(loc[1].item(), 3),
                           steps=n_diag, mllm_stream=pipe, backend=args.backend,
                           note="rank 0's buckets and stream trace; step times are MAX over ranks between barriers; "
                                "exchange_off = the same build and card without the gradient all-reduce (the N = 1 step)")
            log(f"dp diagnostics: {dp_diag}")

    if trainer is not None:
        trainer.release_graph()
    if dp:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    ll = cfg.llama
    M = B * L
    gu_ms = ksum["gateup"][1]
    gu_flops = 2.0 * M * (2 * ll.inter) * ll.hidden
    gu_tflops = gu_flops / (gu_ms * 1e-3) / 1e12
    attn_ms = ksum["attn"][1]
    attn_bytes = B * ATTN_MB_PER_SAMPLE * 1e6 / ll.layers
    kernels = {}
    for name, n_, k_ in (("qkv", (ll.n_q_heads + 2 * ll.n_kv_heads) * ll.head_dim, ll.hidden),
                         ("o", ll.hidden, ll.n_q_heads * ll.head_dim), ("gateup", 2 * ll.inter, ll.hidden),
                         ("down", ll.hidden, ll.inter)):
        ms = ksum[name][1]
        kernels[name] = {"avg_us": round(ms * 1e3, 1), "tflops": round(2.0 * M * n_ * k_ / (ms * 1e-3) / 1e12, 1)}
    kernels["attn"] = {"avg_us": round(attn_ms * 1e3, 1), "algorithmic_GBps": round(attn_bytes / (attn_ms * 1e-3) / 1e9, 1),
                       "hbm_frac": round(attn_bytes / (attn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # HBM-side traffic of the dominant kernel: PMC counters cannot be read inside this process, they come
    # from the committed rocprofv3 --pmc passes of the same kernel / shape (profiles/)
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r04_gateup_gemm_pmc.json")
    if os.path.exists(pmc_path) and (M, 2 * ll.inter, ll.hidden) == (8192, 16384, 2048):
        with open(pmc_path) as f:
            traffic = json.load(f)["traffic_bytes_per_launch"]
        traffic_src = "profiles/r04_gateup_gemm_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950-corrected)"

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        out = {
            "metric": "trajectories/sec", "value": round(value, 2), "unit": "trajectories/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic the path computes in: 16-bit operands of that type on the MFMA units, fp32 accumulation
            # (fp16 = the default storage: same MFMA rate as bf16, 3 more significant bits; --storage bf16 = round 1's)
            "dtype": "f16" if m.storage == torch.float16 else "bf16", "data": "synthetic",
            "feed": ("host: every step collates a fresh batch (custom_collate_fn, 4 synthetic batches in rotation), stages it in pinned "
                     "memory and uploads it with non-blocking copies under the step in flight (data.DeviceFeeder)"
                     if feeder is not None else "resident: one batch in HBM before the timed region (the driver's contract)"),
            "h2d_bytes_per_step": feeder.bytes_per_batch if feeder is not None else 0,
            "config": {
                "workload": ("train.py step (:1168-1183): zero_grad + MultiModalTrajectoryModel.forward incl. loss "
                             "(:914-964) + backward through the trainable part (LTSF + lane-polygon encoder; MLLM "
                             "frozen, :1140-1145) + gradient all-reduce + AdamW(lr 5e-4, wd 1e-4)"
                             if args.mode == "train" else
                             "MultiModalTrajectoryModel.forward incl. loss (train.py:914-964)")
                            + "; Llama-3.2-1B shape + LoRA r=8 + Q-Former + LTSF cross-attention head; " + ("dropout on as under ddp_model.train() (in-kernel Philox masks, regenerated in the backward)" if dropout_on else "dropout off (eval arithmetic)"),
                "mode": args.mode,
                "variant": (("modify_train.py, whole trainable set (adapters + Q-Former + q_proj + modality embeddings; backward "
                             "through all decoder and Q-Former layers, clip_grad_norm 1.0)") if args.train_mllm_front else
                            ("lora_trainable (modify_scripts/modify_train.py:512-528: adapters train, backward through all "
                             "decoder layers, clip_grad_norm 1.0)")) if (args.lora_trainable and args.mode == "train")
                           else "train.py (MLLM frozen)",
                "per_gpu_batch": B, "global_batch": world * B, "fused_seq_len": L, "t_in": cfg.seq_len,
                "t_out": cfg.out_len, "lora_r": cfg.lora_r if cfg.use_lora else 0,
                "parallelism": f"dp{world}" + (" (one-rank RCCL group: the data-parallel path on one card)" if args.rccl_self_test else ""),
                "launch": "eager" if graph is None else "hipGraph replay",
                "pipelining": "; ".join(
                    ([("Q-Former of batch i+1 prefetched on a side stream during step i (every timed step runs one Q-Former "
                       "pass; results identical)")] if prefetch else []) +
                    ([("frozen MLLM pass on a stream of its own: the decoder of step i+1 runs over step i's head, backward "
                       "and AdamW, which it does not depend on (every timed step runs one pass of each; results identical)")]
                     if (graph is None and m.pipeline_decoder) else [])) or "none",
            },
            "achieved_model_tflops": round(value * gflop_per_sample(cfg, L) / 1e3, 1),
            "gflop_per_sample_forward": round(gflop_per_sample(cfg, L), 1),
            "roofline": {
                "kernel": "gemm_bf16_w4_kernel<SILU> (256x256 tile, 4 waves x 128x128; gate|up projection with the post-attention RMSNorm "
                          "fused in as a row scale, SiLU*up epilogue; M=%d N=%d K=%d)" % (M, 2 * ll.inter, ll.hidden),
                "bound": "mfma", "achieved": round(gu_tflops, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(gu_tflops / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src, "algorithmic_bytes": int(2 * (M * ll.hidden + 2 * ll.inter * ll.hidden + M * ll.inter)),
                "launches_timed": ksum["gateup"][0], "avg_launch_us": round(gu_ms * 1e3, 1),
                # the chip holds its clock down at the power cap on real data: clock and power of the step loop (sysfs,
                # un-timed repeat of the same steps) and the fraction against the MFMA peak at THAT clock
                "sustained_clock": (dict(clock, frac_at_sustained_clock=round(
                    gu_tflops / (MFMA_BF16_PEAK_TFLOPS * clock["sclk_mhz"] / RATED_SCLK_MHZ), 4)) if clock else None),
            },
            "kernels": kernels,
            "setup_s": round(setup_s, 1),
            # the synthetic batch is the same every step: in train mode the loss must fall as the optimizer fits it
            "loss_first_step": round(first_loss, 3), "loss_last_timed_step": round(loss_timed, 3),
            "peak_device_memory_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2),
        }
        if trainer is not None and (trainer.skip_nonfinite or trainer.device_step):
            applied, skipped = trainer.optimizer_counters()  # (the gated optimizer skips a step whose loss / gradient norm is not finite)
            out["optimizer_updates"] = {"applied": applied, "skipped": skipped}
            if skipped:
                log(f"WARNING: {skipped} of {applied + skipped} optimizer updates were SKIPPED (non-finite loss or gradient norm)")
        if dp_diag is not None:
            out["dp_diagnostics"] = dp_diag
        if not args.no_cpu_baseline and world >= 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args, gpu_decoded=parity_dec, W=W_cpu) if args.gpus == 1 or world == 1 else None
            if out["cpu_baseline"]:
                out["speedup_vs_cpu_port"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
