import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.util import batch_tensors, load_case, rel_err
from tests.test_training_gpu import _oracle_grads
from tcavt_amd import capi, model, training
capi.init(0)
dev = torch.device("cuda:0")
cfg, weights, fx = load_case(sys.argv[1] if len(sys.argv) > 1 else "tiny_6_12_lora_ragged")
t = batch_tensors(fx)
_, ref16 = _oracle_grads(cfg, weights, t, contract="bf16")
m = model.MultiModalTrajectoryModel.from_config(cfg).load_weights(weights, device=dev).eval()
tr = training.Trainer(m)
g = {k: v.to(dev) for k, v in t.items()}
for rep in range(2):
    loss, dec = tr.forward_backward(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"], g["input_ids"], g["attention_mask"], g["labels"])
    torch.cuda.synchronize()
    errs = sorted(((rel_err(tr.book.g[k].cpu(), ref16[k]), k) for k in ref16 if ref16[k].abs().max() > 0), reverse=True)
    print("rep", rep, "loss", loss.item())
    keys = [k for e, k in errs if "linears" not in k]
    for e, k in errs:
        if "linears" in k and not k.endswith(".0.weight") and not k.endswith(".0.bias"): continue
        print(f"  {e:.3e} {k}  |g|={tr.book.g[k].norm().item():.3e} |ref|={ref16[k].norm().item():.3e}")
