"""Socket power and shader clock WHILE the gate|up GEMM runs (M=8192, N=16384, K=2048, fp16, SiLU epilogue + fused row scale),
on all-zero and on random operands: the direct reading behind DESIGN.md section 4's statement that the kernel is power
limited on real data.  A background thread polls the card's hwmon / DPM files in sysfs (no privileges needed; if they are
not readable it falls back to `rocm-smi --json`) while the main thread launches the kernel back to back for a few seconds.
Measurement tool; never on the product path.

    python tools/power_probe.py [seconds per phase]
"""
import ctypes
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcavt_amd import capi, ops

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
capi.init(0)
dev = torch.device("cuda:0")
M, N, K = 8192, 16384, 2048
dt = torch.float16


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def our_pci_id():
    pr = torch.cuda.get_device_properties(0)
    try:
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except AttributeError:
        return None


def sysfs_sources():
    """(power file in microwatts, sclk file) of THIS process's card (matched by PCI address; the host has eight)."""
    want = our_pci_id()
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    mine = [c for c in cards if want and os.path.basename(os.path.realpath(c)) == want]
    print(f"device 0 = PCI {want}; sysfs card: {mine[0] if mine else 'not found (reading the first card: NOT necessarily ours)'}", flush=True)
    for card in (mine or cards):
        pw = sorted(glob.glob(card + "/hwmon/hwmon*/power1_average")) + sorted(glob.glob(card + "/hwmon/hwmon*/power1_input"))
        fr = sorted(glob.glob(card + "/hwmon/hwmon*/freq1_input"))
        if pw and _read(pw[0]) is not None:
            return pw[0], (fr[0] if fr else None), card + "/pp_dpm_sclk"
    return None, None, None


PW, FR, DPM = sysfs_sources()


def sample():
    if PW:
        p = _read(PW)
        f = _read(FR) if FR else None
        mhz = float(f) / 1e6 if f and f.isdigit() else None
        if mhz is None and DPM:
            cur = [l for l in (_read(DPM) or "").splitlines() if l.endswith("*")]
            if cur:
                mhz = float(cur[0].split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        return (float(p) / 1e6 if p and p.isdigit() else None), mhz
    try:
        out = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                        timeout=5).stdout)
        c = next(iter(out.values()))
        pw = next((float(v) for k, v in c.items() if "Power" in k and "W" in k), None)
        ck = next((float(str(v).strip("()").lower().replace("mhz", "")) for k, v in c.items() if "sclk" in k.lower() and "mhz" in str(v).lower()), None)
        return pw, ck
    except Exception:
        return None, None


class Poller(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.rows = False, []

    def run(self):
        while not self.stop:
            self.rows.append(sample())
            time.sleep(0.02 if PW else 0.2)


part = torch.rand(M, K // 64, device=dev) + 0.5
out = torch.empty(M, N // 2, dtype=dt, device=dev)
print(f"sources: power={PW} clock={FR or DPM}" if PW else "sources: rocm-smi --json", flush=True)
for data in ("idle", "zeros", "randn", "zeros", "randn"):
    if data == "randn":
        x = torch.randn(M, K, device=dev).to(dt)
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(6)]
    else:
        x = torch.zeros(M, K, device=dev, dtype=dt)
        ws = [torch.zeros(N, K, device=dev, dtype=dt) for _ in range(6)]

    def run(i):
        g = capi.GemmArgs()
        w = ws[i % 6]
        g.A, g.lda, g.W, g.ldw, g.C, g.ldc = x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N // 2
        g.M, g.N, g.K, g.tile = M, N, K, 0
        g.in_dtype = g.out_dtype = ops._DT[dt]
        g.epilogue = capi.EPI_SILU_MUL | capi.EPI_ROWSCALE
        g.rowscale_part, g.rowscale_npart, g.rowscale_h, g.rowscale_eps = part.data_ptr(), K // 64, K, 1e-5
        capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")

    torch.cuda.synchronize()
    po = Poller()
    po.start()
    t0 = time.time()
    n = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if data == "idle":
        time.sleep(min(SECS, 2.0))
    else:
        while time.time() - t0 < SECS:
            for i in range(200):
                run(n + i)
            n += 200
            torch.cuda.synchronize()
    b.record()
    torch.cuda.synchronize()
    po.stop = True
    po.join()
    rows = po.rows[len(po.rows) // 4:]  # (skip the ramp)
    pw = [r[0] for r in rows if r[0] is not None]
    ck = [r[1] for r in rows if r[1] is not None]
    us = a.elapsed_time(b) / max(n, 1) * 1e3
    msg = f"{data:6s}: " + (f"{us:7.1f} us per launch ({2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s)  " if n else " " * 42)
    msg += (f"power {sum(pw) / len(pw):6.0f} W (max {max(pw):.0f})  " if pw else "power n/a  ")
    msg += (f"sclk {sum(ck) / len(ck):6.0f} MHz (min {min(ck):.0f}, max {max(ck):.0f})" if ck else "sclk n/a")
    print(msg + f"  [{len(rows)} samples]", flush=True)
    del x, ws
