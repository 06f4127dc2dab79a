set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp1; mkdir -p $O
cd $R
timeout -k 10 240 python3 tools/exp_pipeline.py > $O/pipeline.txt 2>&1; tail -5 $O/pipeline.txt
for pr in 0 -1; do TCAVT_DEC_PRIO=$pr timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('DEC_PRIO $pr', d['ms_per_step'])" | tee -a $O/prio.txt; done
for cfg in "" "TCAVT_SK_NT=1" "TCAVT_SK_U=9" "TCAVT_SK_U=8" "TCAVT_SK_NT=1 TCAVT_SK_U=9" "TCAVT_SK_NT=1 TCAVT_SK_U=8"; do
  for bs in 8 32; do
    echo -n "[$cfg] B=$bs " | tee -a $O/skinny.txt
    env $cfg timeout -k 10 200 python3 tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms_per_step'], d['decode_ms_per_step_min_max_of_5'])" | tee -a $O/skinny.txt
  done
done
cd /tmp
for cfg in "TCAVT_SK_NT=1" "TCAVT_SK_NT=1 TCAVT_SK_U=9"; do
  export $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gtx -o gen -- python3 $R/tools/bench_generate.py --batch 8 > /tmp/genx.log 2>&1
  echo "--- $cfg" >> $O/breakdown.txt; python3 $R/tools/decode_breakdown.py /tmp/gtx/gen_kernel_trace.csv >> $O/breakdown.txt 2>&1
  rm -rf /tmp/gtx
done
cat $O/breakdown.txt
