#!/bin/bash
# PMC passes for the attention-backward kernels (attn_bwd_scores_kernel, attn_bwd_dkv_kernel): tools/pmc_attn_bwd.sh <out.json>
set -u
OUT=$(readlink -f "$1")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
D=/tmp/pmc_attn_bwd; rm -rf $D
i=0
for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $D/p$i -o p -- python3 $R/tools/pmc_attn_bwd.py > $D.log 2>&1 || echo "pass failed: $c"
done
python3 $R/tools/pmc_parse.py $D > $OUT
