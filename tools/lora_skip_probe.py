"""Which steps does the gated optimizer skip in the whole-set variant (Trainer(lora_trainable, train_mllm_front)), and what is not
finite there: loss, gradient norm, and the per-group norms of the flat gradient.  Measurement only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tcavt_amd import capi, config, model, synth, training
from tcavt_amd.weights import make_weights

capi.init(0)
dev = torch.device("cuda", 0)
cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
with torch.device(dev):
    m = model.MultiModalTrajectoryModel.from_config(cfg)
m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev))
m.train(True)
full = "--lora-only" not in sys.argv
tr = training.Trainer(m, lr=5e-4, weight_decay=1e-4, lora_trainable=True, max_grad_norm=1.0, train_mllm_front=full)
b = synth.make_batch(cfg, 32, text_len=240, seed=100, ragged=True, min_text=128)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
args = (g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"], g["input_ids"],
        g["attention_mask"], g["labels"])
names = list(tr.book.g)
for i in range(14):
    loss, _ = tr.forward_backward(*args)
    torch.cuda.synchronize()
    gr = tr.book.grads
    bad = [n for n in names if not torch.isfinite(tr.book.g[n]).all().item()]
    sc = tr.lbw._buf("scale", (2,), torch.float32).tolist() if hasattr(tr.lbw, "_buf") else None
    tr.optimizer_step()
    torch.cuda.synchronize()
    a, s = tr.optimizer_counters()
    print(f"step {i}: loss {loss.item():.4g}  |grad| {gr.float().norm().item():.4g}  finite {torch.isfinite(gr).all().item()}  scale {sc}  "
          f"applied {a} skipped {s}  non-finite tensors: {bad[:6]}{' ...' if len(bad) > 6 else ''} ({len(bad)})", flush=True)
