# HBM-side bytes per launch of the decode step's skinny GEMMs (rocprofv3 --pmc FETCH_SIZE, its own pass; kernel-trace only),
# row-major weights against the fragment-major copy: the access-pattern story of DESIGN.md section 7 in counters.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_decode; mkdir -p $O
for cfg in frag row; do
  D=/tmp/pmc_dec_$cfg; rm -rf $D
  if [ $cfg = row ]; then export TCAVT_DECODE_ROWMAJOR=1; else unset TCAVT_DECODE_ROWMAJOR; fi
  i=0
  for c in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $D/p$i -o p -- python3 $R/tools/bench_generate.py --batch 8 --no-graph --new 24 > $D.log 2>&1 || echo "pass failed: $cfg $c"
  done
  python3 $R/tools/pmc_parse.py $D > $O/pmc_decode_$cfg.json; echo "$cfg done"
done
python3 - <<'PY'
import json,os
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","pmc_decode")
for cfg in ("frag","row"):
    d=json.load(open(os.path.join(O,f"pmc_decode_{cfg}.json")))
    for k,v in d.items():
        if "skinny" in k:
            print(cfg, k, {a:(round(b,1) if not isinstance(b,list) else b[:2]) for a,b in v.items()})
PY
