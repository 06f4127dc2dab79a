"""Text-generation throughput at the full model size (Llama-3.2-1B shape): prefill + hipGraph-replayed decode steps
(BASELINE.json configs[4] "hipGraph-captured decode"; reference: scripts/train.py:577-654).  One MI355X.
    python tools/bench_generate.py [--batch 32] [--new 64] [--text-len 240] [--no-graph] [--greedy]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tcavt_amd import capi, config, model, synth
from tcavt_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--new", type=int, default=64)
ap.add_argument("--text-len", type=int, default=240)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--greedy", action="store_true")
ap.add_argument("--preset", default="llama32_1b")
args = ap.parse_args()
capi.init(0)
dev = torch.device("cuda:0")
cfg = config.PRESETS[args.preset]()
with torch.device(dev):
    m = model.MultiModalTrajectoryModel.from_config(cfg)
m.load_weights(make_weights(cfg, seed=1, backend="torch", device=dev)).eval()
b = synth.make_batch(cfg, args.batch, text_len=args.text_len, seed=3, ragged=True, min_text=min(128, args.text_len // 2))
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
kw = dict(input_ids=g["input_ids"], attention_mask=g["attention_mask"], do_sample=not args.greedy, use_graph=not args.no_graph)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.mllm.generate_batch(g["vision_emb"], None, max_new_tokens=n, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


run(4)  # packed weights, workspaces
run(args.new)  # clocks, graph capture path
reps = []
for _ in range(5):  # (a decode step is latency-bound and sensitive to the clock state: median of five)
    t1, _ = run(1)          # prefill + first token
    tn, out = run(args.new)  # prefill + args.new tokens
    reps.append(((tn - t1) / max(args.new - 1, 1), t1))
reps.sort()
per_tok, t1 = reps[len(reps) // 2]
ll = cfg.llama
wbytes = 2 * (ll.layers * (ll.hidden * (ll.n_q_heads + 2 * ll.n_kv_heads) * ll.head_dim + ll.n_q_heads * ll.head_dim * ll.hidden
                           + 3 * ll.hidden * ll.inter) + ll.vocab * ll.hidden)
print(json.dumps({
    "workload": f"generate_batch: B={args.batch}, prompt {cfg.q_num_query_tokens}+{args.text_len} tokens (ragged), {args.new} new tokens, "
                f"{'greedy' if args.greedy else 'sampling T=0.9 top-k 40 top-p 0.9 rep 1.2 no-repeat-3'}, "
                f"{'eager launches' if args.no_graph else 'hipGraph replay of the decode step'}",
    "prefill_plus_first_token_ms": round(t1 * 1e3, 2), "decode_ms_per_step": round(per_tok * 1e3, 3),
    "decode_ms_per_step_min_max_of_5": [round(reps[0][0] * 1e3, 3), round(reps[-1][0] * 1e3, 3)],
    "decode_tokens_per_s": round(args.batch / per_tok, 1),
    "weight_bytes_per_step": wbytes, "weight_stream_GBps": round(wbytes / per_tok / 1e9, 1),
    "hbm_frac_of_8TBps": round(wbytes / per_tok / 8e12, 4), "finite": bool(((out >= 0) & (out < ll.vocab)).all().item())}))
