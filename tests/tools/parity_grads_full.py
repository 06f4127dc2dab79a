"""Full-size gradient parity (train.py parameter set and the LoRA adapters of the LoRA-trainable variant) (Llama-3.2-1B shape, L = 256): the HIP path's adapter gradients for a few
synthetic samples against torch autograd through the oracle (bf16 contract and fp32), same host-generated weights.
Prints one JSON object; the numbers are quoted in DESIGN.md.   usage: tests/tools/parity_grads_full.py [samples=2]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from oracle import forward as O
from tcavt_amd import capi, config, model, synth, training
from tcavt_amd.weights import make_weights

capi.init(0)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_num_threads(min(16, os.cpu_count() or 1))
cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
t0 = time.time()
W = make_weights(cfg, seed=1, backend="torch", device="cpu")
print(f"[parity] weights generated in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
b = synth.make_batch(cfg, B, text_len=240, seed=1, ragged=True, min_text=128)
t = {k: torch.from_numpy(v) for k, v in b.items()}

with torch.device(dev):
    m = model.MultiModalTrajectoryModel.from_config(cfg)
m.load_weights(W)
m.to(dev).eval()
m.invalidate_prepared()
tr = training.Trainer(m, lora_trainable=True)
g = {k: v.to(dev) for k, v in t.items()}
loss, _ = tr.forward_backward(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"],
                              g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"])
torch.cuda.synchronize()
keys = [k for k in W if ".lora_A." in k or ".lora_B." in k]
base = [k for k in W if not k.startswith("mllm.")]  # the train.py trainable set (LTSF + lane-polygon encoder)
got = {k: tr.book.g[k].detach().float().cpu() for k in keys + base}
res = {"samples": B, "L": cfg.q_num_query_tokens + 240, "hip_loss": float(loss.item()), "adapter_tensors": len(keys),
       "train_py_tensors": len(base)}
for contract in ("bf16", "fp32"):
    Wc = {k: v.detach().clone() for k, v in W.items()}
    for k in keys + base:
        Wc[k].requires_grad_(True)
    t1 = time.time()
    ref_loss, _ = O.model_forward(Wc, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                  t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                  contract=contract)
    ref_loss.backward()
    print(f"[parity] oracle ({contract}) forward + backward in {time.time() - t1:.1f} s", file=sys.stderr, flush=True)
    res[contract] = {"oracle_loss": float(ref_loss.item())}
    for name, ks in (("adapters", keys), ("train_py_set", base)):
        ks = [k for k in ks if Wc[k].grad is not None and Wc[k].grad.abs().max() > 0]
        fr = torch.cat([Wc[k].grad.reshape(-1).double() for k in ks])
        fg = torch.cat([got[k].reshape(-1).double() for k in ks])
        per = sorted(((got[k].double() - Wc[k].grad.double()).norm() / Wc[k].grad.double().norm()).item() for k in ks)
        res[contract][name] = {"flat_rel_err": ((fg - fr).norm() / fr.norm()).item(),
                               "cosine": (fg @ fr / (fg.norm() * fr.norm())).item(),
                               "per_tensor_rel_err_median": per[len(per) // 2], "per_tensor_rel_err_max": per[-1],
                               "grad_norm": fr.norm().item(), "tensors": len(ks)}
    del Wc
print(json.dumps(res, indent=1))
