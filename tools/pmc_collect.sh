#!/bin/bash
# Collects the PMC passes for one GEMM shape/kernel on the GPU box and writes a JSON summary.
#   tools/pmc_collect.sh <tile code | lib> <gateup|down|o> <out.json>
# Separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE each on their own), parsed by tools/pmc_parse.py.
set -u
W=$1; SHAPE=$2; OUT=$(readlink -f "$3")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_${W}_${SHAPE}; rm -rf $D
i=0
for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $D/p$i -o p -- python3 $R/tools/pmc_gemm.py $W $SHAPE > $D.log 2>&1 || echo "pass failed: $c"
done
python3 $R/tools/pmc_parse.py $D > $OUT
