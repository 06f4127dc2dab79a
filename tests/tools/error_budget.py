"""Error budget of the 16-bit storage contract (VERDICT r1 task 2a): which rounding point costs what, end to end.

Runs the CPU oracle (oracle/forward.py) on the cpu_baseline samples of bench.py with per-point storage modes and
tabulates the relative error of final_hidden / decoded and the relative difference of ADE / FDE against the fp32 graph:
  * the whole contract in bf16 (the round-1 HIP path), in fp16, and mixed forms;
  * every rounding point toggled alone (everything else fp32), in bf16 and in fp16.
CPU only (no GPU, nothing from the product package but config / synth / weights).  Usage:
    python tests/tools/error_budget.py [--preset llama32_1b|midi|tiny] [--batch 2] [--text-len 240] [--out profiles/x.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from oracle import forward as O  # noqa: E402
from tcavt_amd import config, synth  # noqa: E402
from tcavt_amd.weights import make_weights  # noqa: E402

POINTS = ["w", "gamma", "xn", "t", "qkv", "p", "att", "act", "res", "emb", "qf", "fh", "xa"]
NOTE = {"w": "decoder weights", "gamma": "RMSNorm gains", "xn": "normalised rows (A operand of q|k|v, gate|up)",
        "t": "LoRA down-projection", "qkv": "rotated q, k, v", "p": "attention probabilities", "att": "attention output",
        "act": "silu(gate)*up", "res": "decoder residual stream (after every residual add)", "emb": "embedding table", "qf": "Q-Former + q_proj (all points)",
        "fh": "final hidden states handed to the head", "xa": "LTSF cross-attention head (all other points)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="llama32_1b")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--text-len", type=int, default=240)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = config.PRESETS[args.preset]()
    t0 = time.time()
    W = make_weights(cfg, seed=1, backend="torch", device="cpu")
    b = synth.make_batch(cfg, args.batch, text_len=args.text_len, seed=1, ragged=True,
                         min_text=128 if args.text_len > 128 else max(1, args.text_len // 2))
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    print(f"weights + batch in {time.time() - t0:.0f} s", flush=True)

    def run(contract):
        ex = {}
        with torch.no_grad():
            _, dec = O.model_forward(W, cfg, t["traj_emb"], t["vision_emb"], t["lane_polygon"], t["lane_polygon_len"],
                                     t["input_ids"], t["attention_mask"], y=t["target_traj"], norm_stat=t["norm_stat"],
                                     contract=contract, extras=ex)
        return ex["final_hidden"], dec, O.traj_metrics(dec, t["target_traj"], t["norm_stat"])

    fh0, dec0, m0 = run("fp32")

    def row(name, contract):
        t1 = time.time()
        fh, dec, m = run(contract)
        rel = lambda a, b_: ((a.double() - b_.double()).norm() / b_.double().norm()).item()
        r = {"contract": name, "final_hidden": rel(fh, fh0), "decoded": rel(dec, dec0),
             "ade": abs(m["ade_sum"] - m0["ade_sum"]) / m0["ade_sum"], "fde": abs(m["fde_sum"] - m0["fde_sum"]) / m0["fde_sum"]}
        print(f"{name:44s} final_hidden {r['final_hidden']:.2e}  decoded {r['decoded']:.2e}  ADE {r['ade']:.2e}  "
              f"FDE {r['fde']:.2e}   ({time.time() - t1:.0f} s)", flush=True)
        return r

    rows = [row("all bf16 (round-1 contract, P fp16)", "bf16"),
            row("all fp16 (the HIP path's contract: residual stream fp16)", "fp16"),
            row("fp16, residual stream fp32 (the contract before the 16-bit stream)", {"default": "fp16", "gamma": "fp32", "res": "fp32"}),
            row("fp16, weights bf16", {"default": "fp16", "w": "bf16"}),
            row("bf16, weights fp16", {"default": "bf16", "p": "fp16", "w": "fp16"}),
            row("bf16, activations xn/att/act fp16", {"default": "bf16", "p": "fp16", "xn": "fp16", "att": "fp16", "act": "fp16"}),
            row("bf16, xn/att/act/qkv + weights fp16", {"default": "bf16", "p": "fp16", "xn": "fp16", "att": "fp16", "act": "fp16",
                                                          "qkv": "fp16", "w": "fp16"}),
            row("decoder fp16, Q-Former + head bf16", {"default": "fp16", "qf": "bf16", "xa": "bf16", "fh": "bf16"})]
    for mode in ("bf16", "fp16"):
        for pt in POINTS:
            rows.append(row(f"only {pt} in {mode} ({NOTE[pt]})", {"default": "fp32", pt: mode}))
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"preset": args.preset, "batch": args.batch, "fused_len": cfg.q_num_query_tokens + args.text_len,
                       "against": "fp32 oracle graph, same weights and samples", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
