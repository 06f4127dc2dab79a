"""Which HIP API calls block the host: from a rocprofv3 --hip-trace csv, the calls longer than a threshold, grouped by name,
plus the longest ones in time order (measurement only: where does the host wait on the card?)."""
import csv
import sys
from collections import defaultdict

path, thr_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
rows = list(csv.DictReader(open(path)))
name_k = "Function" if "Function" in rows[0] else "Name"
t0 = min(int(r["Start_Timestamp"]) for r in rows)
agg = defaultdict(lambda: [0, 0.0, 0.0])
long_calls = []
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[r[name_k]]
    a[0] += 1
    a[1] += d
    a[2] = max(a[2], d)
    if d >= thr_us:
        long_calls.append(((int(r["Start_Timestamp"]) - t0) / 1e6, d / 1e3, r[name_k]))
print(f"{len(rows)} HIP calls; by total host time:")
for n, (c, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {tot / 1e3:10.1f} ms  n={c:6d}  max {mx / 1e3:8.2f} ms  {n}")
print(f"calls >= {thr_us:.0f} us, in time order (start ms, duration ms):")
for s, d, n in sorted(long_calls)[:80]:
    print(f"  +{s:9.2f}  {d:8.2f}  {n}")
