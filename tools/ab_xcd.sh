#!/bin/bash
# A/B of the XCD partition of the GEMM tile grid (TCAVT_GEMM_XCD_GX forces gx; 8 = row bands)
for gx in 8 4 2 1; do
  echo "== gx=$gx"
  TCAVT_GEMM_XCD_GX=$gx python tools/bench_gemm_cold.py 2>/dev/null | grep -E "tile=256 rotate"
  TCAVT_GEMM_XCD_GX=$gx timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode forward 2>/dev/null | tail -1 > /tmp/ab_x.json
  python - <<PY
import json
d = json.load(open("/tmp/ab_x.json"))
print("model gx=$gx", d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
PY
done
