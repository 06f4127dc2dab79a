"""The decoder's HIP queue in one steady-state train.py step (frozen-MLLM variant, pipelined): busy time by kernel, idle time by
(kernel before, kernel after) pair -- where does the stream that bounds the step stand still?  Input: a rocprofv3 kernel trace
(csv) of bench.py.  Measurement tool.

    python tools/decoder_queue_timeline.py <kernel_trace.csv>
"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tcavt::", "")[:56]
    dq = collections.Counter(r["Queue_Id"] for r in rows if "gemm_bf16_w4_kernel" in r["Kernel_Name"]).most_common(1)[0][0]
    q = [r for r in rows if r["Queue_Id"] == dq]
    # steady-state window: from the start of an embed_fuse to the start of the embed_fuse three passes later
    emb = [i for i, r in enumerate(q) if "embed_fuse" in r["Kernel_Name"]]
    if len(emb) < 8:
        raise SystemExit("fewer than 8 decoder passes in the trace")
    i0, i1 = emb[-6], emb[-3]
    win = q[i0:i1]
    span = (q[i1]["s"] - q[i0]["s"]) / 1e3 / 3
    busy = sum(r["e"] - r["s"] for r in win) / 1e3 / 3
    print(f"decoder queue {dq}: {len(win) // 3} kernels per pass, pass period {span:.1f} us, kernels {busy:.1f} us, idle {span - busy:.1f} us")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in win:
        a = acc[short(r)]
        a[0] += 1
        a[1] += (r["e"] - r["s"]) / 1e3
    print("by kernel (per pass):")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t / 3:9.1f} us  n={n // 3:3d}  avg {t / n:7.1f} us  {k}")
    gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for a, b in zip(q[i0:i1], q[i0 + 1:i1 + 1]):
        g = (b["s"] - a["e"]) / 1e3
        e = gaps[(short(a), short(b))]
        e[0] += 1
        e[1] += g
        e[2] = max(e[2], g)
    print("idle by neighbours (per pass):")
    for (a, b), (n, t, mx) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {t / 3:8.1f} us  n={n // 3:3d}  avg {t / n:6.2f} us  max {mx:7.1f} us  after {a}  before {b}")


if __name__ == "__main__":
    main(sys.argv[1])
