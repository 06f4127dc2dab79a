"""Where the decoder attention kernel's waves spend their cycles: s_memtime stamps at the phase boundaries (experiments build:
python -m tcavt_amd.build --experiments; TCAVT_LIB=exp).  Prints, per wave position, the median cycles of: entry -> staging
stores issued -> K / V visible -> each query block done, over all workgroups, and the kernel's wall time."""
import ctypes
import os
import sys

os.environ["TCAVT_LIB"] = "exp"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi

capi.init(0)
dev = torch.device("cuda:0")
B, L, nq, nkv = 32, 256, 32, 8
qkv = (torch.randn(B * L, (nq + 2 * nkv) * 64, device=dev) * 0.5).to(torch.float16)
out = torch.empty(B * L, nq * 64, dtype=torch.float16, device=dev)
g = torch.Generator().manual_seed(0)
kv_len = torch.randint(144, 257, (B,), generator=g, dtype=torch.int32).to(dev)
stamps = torch.zeros(B * nkv, 8, 16, dtype=torch.int64, device=dev)
fn = capi.lib().tcavt_attn_causal_gqa_stamped
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
for _ in range(5):
    assert fn(qkv.data_ptr(), out.data_ptr(), kv_len.data_ptr(), B, L, nq, nkv, 0.125, stamps.data_ptr(), capi.stream_ptr()) == 0
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    fn(qkv.data_ptr(), out.data_ptr(), kv_len.data_ptr(), B, L, nq, nkv, 0.125, stamps.data_ptr(), capi.stream_ptr())
b.record()
torch.cuda.synchronize()
print(f"stamped kernel: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch")
s = stamps.cpu().double()
t0 = s[:, :, 0:1]
d = (s - t0)  # cycles since the wave's entry
names = ["entry", "staging issued", "K/V visible", "block 1", "block 2", "block 3", "block 4"]
for w in range(8):
    row = [f"{d[:, w, i].median().item():8.0f}" for i in range(1, 7)]
    print(f"wave {w} (head {w % 4}, parity {w // 4}): " + " ".join(f"{n}={v}" for n, v in zip(names[1:], row)))
first = s[:, :, 0].min()
last = s[:, :, 6].max()
print(f"all workgroups: first entry -> last block done = {(last - first).item():.0f} cycles; "
      f"median wave lifetime {d[:, :, 6].median().item():.0f} cycles; entry spread (max - min of wave entries) {(s[:, :, 0].max() - first).item():.0f}")
