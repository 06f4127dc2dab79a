set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp23; mkdir -p $O
cd $R
for cfg in "X=1" "TCAVT_DECODE_SPLITK=1" "TCAVT_DECODE_SPLITK=1 TCAVT_DECODE_LORA_LAUNCH=1" "TCAVT_DECODE_LORA_LAUNCH=1"; do
  for bs in 8; do
    echo -n "[$cfg] B=$bs " | tee -a $O/out.txt
    env $cfg timeout -k 10 200 python3 tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms_per_step'], d['decode_ms_per_step_min_max_of_5'])" | tee -a $O/out.txt
  done
done
