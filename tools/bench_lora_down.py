"""tcavt_lora_down alone at the bench shape (M = 8192, H = 2048), eval and train mode, rotating (cache-cold) inputs.  Measurement only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tcavt_amd import capi, ops  # noqa: E402

capi.init(0)
dev = torch.device("cuda:0")
M, H = 8192, 2048
xs = [torch.randn(M, H, device=dev).half() for _ in range(12)]
a_cat = (torch.randn(64, H, device=dev) * 0.02).half()
t = torch.zeros(M, 64, dtype=torch.float16, device=dev)
for name, drop in (("eval", None), ("train p=0.1", (0.1, 1234, 7))):
    for i in range(4):
        ops.lora_down(xs[i], a_cat, t, 4.0, dropout=drop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(48):
        ops.lora_down(xs[i % 12], a_cat, t, 4.0, dropout=drop)
    e1.record()
    torch.cuda.synchronize()
    print(f"lora_down {name}: {e0.elapsed_time(e1) / 48 * 1e3:.1f} us", flush=True)

# ---- tcavt_lora_dgrad (the adapters' input gradients, LoRA-trainable backward): 33.5 MB written per call
g_t = (torch.randn(M, 64, device=dev) * 0.1).half()
aqT = (torch.randn(H, 64, device=dev) * 0.02).half()
avT = (torch.randn(H, 64, device=dev) * 0.02).half()
outs = [torch.empty(M, H, dtype=torch.float16, device=dev) for _ in range(12)]
for name, drop in (("eval", None), ("train p=0.1", (0.1, 1234, 7))):
    for i in range(4):
        ops.lora_dgrad(g_t, aqT, avT, outs[i], dropout=drop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(48):
        ops.lora_dgrad(g_t, aqT, avT, outs[i % 12], dropout=drop)
    e1.record()
    torch.cuda.synchronize()
    print(f"lora_dgrad {name}: {e0.elapsed_time(e1) / 48 * 1e3:.1f} us", flush=True)
