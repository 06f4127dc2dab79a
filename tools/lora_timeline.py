"""Per-stream timeline of one steady-state step from a rocprofv3 kernel trace (csv) of the LoRA-trainable bench: which HIP
queue is busy for how long, what the main queue spends its time on, and its idle gaps.  Kernel durations on the side
queues are inflated by contention (their workgroups wait for CUs held by the main queue's GEMMs), so the statement that
matters is the main queue's busy time against the step.  Measurement tool; never on the product path.

    python tools/lora_timeline.py gpurun_out/r02/prof_lora/lora_kernel_trace.csv
"""
import collections
import csv
import sys


def union_ms(v):
    ev = sorted((r["s"], r["e"]) for r in v)
    busy, (cs, ce) = 0, ev[0]
    for s, e in ev[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return (busy + ce - cs) / 1e6


def main(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    ends = [r for r in rows if "adamw_gated_kernel" in r["Kernel_Name"] or r["Kernel_Name"].startswith("tcavt::adamw_kernel")]
    if len(ends) < 4:
        raise SystemExit("fewer than 4 optimizer launches in the trace")
    t0, t1 = ends[-3]["e"], ends[-2]["e"]
    step = [r for r in rows if r["s"] >= t0 and r["e"] <= t1 + 1]
    print(f"step (optimizer launch to optimizer launch, under the profiler): {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels")
    by = collections.defaultdict(list)
    for r in step:
        by[r["Queue_Id"]].append(r)
    for q, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        print(f"queue {q}: {len(v):4d} kernels, busy {union_ms(v):6.2f} ms")
    print(f"any queue busy: {union_ms(step):.2f} ms")
    main_q = max(by.values(), key=len)
    acc = collections.defaultdict(lambda: [0, 0])
    for r in main_q:
        k = r["Kernel_Name"].split("(")[0][-70:]
        acc[k][0] += 1
        acc[k][1] += r["e"] - r["s"]
    print("main queue by kernel:")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {t / 1000:8.1f} us  n={n:3d}  avg {t / n / 1000:7.1f} us  {k}")
    gaps = [(b["s"] - a["e"]) / 1000 for a, b in zip(main_q, main_q[1:])]
    print(f"main queue idle: {sum(g for g in gaps if g > 0) / 1000:.2f} ms in {len(gaps)} gaps ({sum(1 for g in gaps if g > 30)} above 30 us)")
    # the largest gaps with their neighbours, and what the other queues ran meanwhile
    order = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:24]
    short = lambda r: r["Kernel_Name"].split("(")[0][-48:]
    for i in sorted(order):
        a, b = main_q[i], main_q[i + 1]
        other = [r for r in step if r["Queue_Id"] != a["Queue_Id"] and r["e"] > a["e"] and r["s"] < b["s"]]
        busy = union_ms(other) * 1000 if other else 0.0
        print(f"  gap {gaps[i]:7.1f} us at +{(a['e'] - t0) / 1e6:6.2f} ms  after {short(a)}  before {short(b)}  "
              f"(other queues: {len(other)} kernels, {busy:.0f} us busy)")
        if gaps[i] > 400:  # what the other queues run inside the big gaps
            for r in sorted(other, key=lambda r: r["s"]):
                print(f"        q{r['Queue_Id']} +{(r['s'] - a['e']) / 1e3:7.1f} us  {(r['e'] - r['s']) / 1e3:6.1f} us  {short(r)}")


if __name__ == "__main__":
    main(sys.argv[1])
