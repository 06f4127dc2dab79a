"""Host time of a LoRA-trainable step (the Python call returning) against its GPU time: is the launch queue ahead of the GPU,
or does something in the step wait for the device?  Measurement tool; never on the product path.
    python tools/lora_host_time.py [--front]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, config, model, synth, training
from tcavt_amd.weights import make_weights

capi.init(0)
dev = torch.device("cuda:0")
cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
with torch.device(dev):
    m = model.MultiModalTrajectoryModel.from_config(cfg)
m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev))
m.train(True)
b = synth.make_batch(cfg, 32, text_len=240, seed=100, ragged=True, min_text=128)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
tr = training.Trainer(m, lr=5e-4, weight_decay=1e-4, lora_trainable=True, max_grad_norm=1.0, train_mllm_front="--front" in sys.argv)


def step():
    return tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
                   g["input_ids"], g["attention_mask"], g["labels"], next_vision_embs=g["vision_emb"])


for _ in range(4):
    step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    step()
    host.append((time.perf_counter() - a) * 1e3)
t_host = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) * 1e3
print(f"10 steps: host enqueue {t_host:.1f} ms (per step: {' '.join(f'{h:.1f}' for h in host)}), with the final sync {t_all:.1f} ms "
      f"= {t_all / 10:.2f} ms per step")
