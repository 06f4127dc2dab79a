"""Kernel breakdown of ONE decode step (between two consecutive token selections) from a rocprofv3 kernel trace (csv) of
tools/bench_generate.py.  Measurement tool; never on the product path.

    python tools/decode_breakdown.py gpurun_out/r02/prof_gen/gen_kernel_trace.csv
"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    smp = [i for i, r in enumerate(rows) if "sample_kernel" in r["Kernel_Name"]]
    i0, i1 = smp[-10], smp[-9]
    step = rows[i0 + 1:i1 + 1]
    print(f"decode step: {len(step)} kernels, span {(step[-1]['e'] - step[0]['s']) / 1000:.1f} us, "
          f"sum of kernel durations {sum(r['e'] - r['s'] for r in step) / 1000:.1f} us")
    acc = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][0] += 1
        acc[k][1] += r["e"] - r["s"]
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {t / 1000:8.1f} us  n={n:3d}  avg {t / n / 1000:6.1f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
