"""Where the pipelined train step spends its time on the MLLM stream (no profiler: HIP events only): busy time of every
MLLM pass (embed + decoder + final norm), the idle time between two passes, and the step period."""
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
if os.environ.get("TCAVT_EXP_NO_LORA_DROPOUT"):  # (timing experiment: what the adapters' mask generation costs the step)
    m.mllm.llama_wrapper.lora_dropout = 0.0
b = synth.make_batch(cfg, 32, text_len=240, seed=100, ragged=True, min_text=128)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
tr = training.Trainer(m, lr=5e-4, weight_decay=1e-4)


def step():
    return tr.step(g["traj_emb"], g["vision_emb"], g["lane_polygon"], g["lane_polygon_len"], g["target_traj"], g["norm_stat"],
                   g["input_ids"], g["attention_mask"], g["labels"], next_vision_embs=g["vision_emb"], inputs_ready=True)


with torch.no_grad():
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    m.pipe_trace = []
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    period = (time.perf_counter() - t) / n * 1e3
    ev = m.pipe_trace
    busy = [a.elapsed_time(b_) for a, b_ in ev]
    idle = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)]
    print(f"step period {period:.3f} ms; MLLM pass busy {sum(busy) / len(busy):.3f} ms (min {min(busy):.3f}, max {max(busy):.3f}); "
          f"idle between passes {sum(idle) / len(idle):.3f} ms (min {min(idle):.3f}, max {max(idle):.3f})")
