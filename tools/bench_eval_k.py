"""BASELINE.json configs[4] (SURVEY 8d "Config 5", test_10.py shape: T_in 6, T_out 12, L = 256): throughput of the
evaluation protocol on one GPU -- K = 1 (eval arithmetic, train.py:1274-1326) and K = 10 MC-dropout candidates
(test.py:1301-1382), plus K = 10 with the shared MLLM pass when the MLLM has no dropout site (reuse_prefix)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, config, evaluate, model, synth
from tcavt_amd.weights import make_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
capi.init(0)
dev = torch.device("cuda:0")
cfg = config.PRESETS["llama32_1b"](seq_len=6, out_len=12, use_lora=True)
with torch.device(dev):
    m = model.MultiModalTrajectoryModel.from_config(cfg)
m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev))
m.eval()
b = synth.make_batch(cfg, B, text_len=240, seed=3, ragged=True, min_text=128)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}


def run(label, n_batches, **kw):
    evaluate.evaluate_model(m, [g], **kw)  # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = evaluate.evaluate_model(m, [g] * n_batches, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {n_batches * B / dt:8.1f} trajectories/s  ({dt / n_batches * 1e3:7.1f} ms per batch of {B})  "
          f"ADE {r['ADE']:.3f} FDE {r['FDE']:.3f}", flush=True)


run("K=1  eval", 3, num_candidates=1)
run("K=10 MC-dropout (10 full passes)", 1, num_candidates=10, mc_dropout=True)
m.mllm.qformer.dropout_p = 0.0
m.mllm.llama_wrapper.lora_dropout = 0.0
run("K=10 MC-dropout, MLLM without dropout sites, shared MLLM pass", 2, num_candidates=10, mc_dropout=True, reuse_prefix=True)
