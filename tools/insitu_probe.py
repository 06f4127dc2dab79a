"""Why the decoder's GEMMs take longer inside the model than alone.  Measurement only.

Part 1: the MLLM pass ALONE (nothing on any other stream), per-kernel times from the stack's own HIP events.
Part 2: the in-model forms of o / down / gate|up launched alone, each launch timed by its own event pair, in three cache states:
   hot    -- same activation buffer, 16 rotating weight matrices, back to back (what tools/ab_w4.py measures)
   fresh  -- the activation operand is re-written by a copy kernel right before every launch (as the producer kernel does)
   cold   -- 1 GiB streamed through the chip before every launch (nothing of the operands left in L2 / Infinity Cache)
   cold+w -- cold, then the launch's WEIGHT matrix read once by a streaming kernel (what a weight prefetch would leave behind)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tcavt_amd import capi, config, model, ops, synth  # noqa: E402
from tcavt_amd.weights import make_weights  # noqa: E402

capi.init(0)
dev = torch.device("cuda:0")


def part1():
    cfg = config.PRESETS["llama32_1b"](seq_len=18, out_len=30, use_lora=True)
    with torch.device(dev):
        m = model.MultiModalTrajectoryModel.from_config(cfg)
    m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev))
    m.train(True)
    b = synth.make_batch(cfg, 32, text_len=240, seed=100, ragged=True, min_text=128)
    g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    timer = ops.StackEvents(cfg.llama.layers)
    acc = {}
    with torch.no_grad():
        for it in range(8):
            if it == 3:
                m.mllm.llama_wrapper.timer = timer
            m.mllm(g["vision_emb"], None, input_ids=g["input_ids"], attention_mask=g["attention_mask"], return_bf16=True)
            torch.cuda.synchronize()
            if it >= 3:
                for k, (cnt, ms) in timer.summary().items():
                    c0, t0 = acc.get(k, (0, 0.0))
                    acc[k] = (c0 + cnt, t0 + cnt * ms)
    m.mllm.llama_wrapper.timer = None
    timer.close()
    print("MLLM pass alone (no other stream), per kernel:", {k: round(t / c * 1e3, 1) for k, (c, t) in acc.items()}, flush=True)
    del m
    torch.cuda.empty_cache()


def part2():
    dt = torch.float16
    flush_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    flush_b = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for shape, (M, N, K) in (("o", (8192, 2048, 2048)), ("down", (8192, 2048, 8192)), ("gateup", (8192, 16384, 2048))):
        a = (torch.randn(M, K, device=dev) * (0.2 if shape != "gateup" else 1.0)).to(dt)
        a_src = a.clone()
        ws = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(16)]
        out = torch.empty(M, N // 2, dtype=dt, device=dev) if shape == "gateup" else None
        part = torch.rand(M, K // 64, device=dev) + 0.5
        h16 = torch.randn(M, N, device=dev).to(dt) if shape != "gateup" else None
        pout = torch.empty(M, N // 64, device=dev)
        sink = torch.empty(N * K // 2, dtype=torch.float32, device=dev)

        def launch(w):
            g = capi.GemmArgs()
            g.A, g.lda, g.W, g.ldw = a.data_ptr(), K, w.data_ptr(), K
            g.M, g.N, g.K, g.tile = M, N, K, 0
            g.in_dtype = capi.F16
            if shape == "gateup":
                g.C, g.ldc, g.out_dtype = out.data_ptr(), N // 2, capi.F16
                g.epilogue = capi.EPI_SILU_MUL | capi.EPI_ROWSCALE
                g.rowscale_part, g.rowscale_npart, g.rowscale_h, g.rowscale_eps = part.data_ptr(), K // 64, K, 1e-5
            else:
                g.C, g.ldc, g.out_dtype = None, N, capi.F32
                g.epilogue = capi.EPI_RESIDUAL | capi.EPI_NORM_OUT
                g.norm_h16, g.norm_part = h16.data_ptr(), pout.data_ptr()
            capi.check(capi.lib().tcavt_gemm_bf16(ctypes.byref(g), capi.stream_ptr()), "gemm")

        def run(mode, n=24):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for i in range(n + 4):
                w = ws[i % 16]
                if mode in ("cold", "cold+w"):
                    flush_a.copy_(flush_b)
                if mode == "cold+w":
                    sink.copy_(w.view(torch.float32).view(-1))  # (reads W once; the small write goes elsewhere)
                if mode == "fresh":
                    a.copy_(a_src)
                if i >= 4:
                    ev[i - 4][0].record()
                launch(w)
                if i >= 4:
                    ev[i - 4][1].record()
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
            return ts[len(ts) // 2], ts[0], ts[-1]

        line = f"{shape:7s}"
        for mode in ("hot", "fresh", "cold", "cold+w", "hot"):
            med, lo, hi = run(mode)
            line += f" | {mode}: {med:6.1f} us ({lo:.1f}-{hi:.1f})"
        print(line, flush=True)


if __name__ == "__main__":
    if "--no-model" not in sys.argv:
        part1()
    if "--part1-only" not in sys.argv:
        part2()
