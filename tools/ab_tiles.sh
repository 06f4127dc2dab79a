#!/bin/bash
# A/B of the decoder GEMM main-loop variants: micro-benchmark and inside the real model (forward mode), same box
python tools/bench_gemm_cold.py 2>/dev/null | grep -E "rotate"
for t in 255 256 254; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode forward --tile $t 2>/dev/null | tail -1 > /tmp/ab_$t.json
  python - <<PY
import json
d = json.load(open("/tmp/ab_$t.json"))
print("model tile $t", d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
