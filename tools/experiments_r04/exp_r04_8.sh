set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp8; mkdir -p $O
cd $R
TCAVT_AB_NOCHECK=1 TCAVT_AB_DTYPE=fp16 TCAVT_AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_w4.py 257,266,272,274 > $O/ab_wtiled.txt 2>&1
cat $O/ab_wtiled.txt | tail -14
