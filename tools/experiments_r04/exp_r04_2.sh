set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp2; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_generation_gpu.py -x -q -m gpu -k "skinny or generation or decode or greedy or sampl or token" > $O/tests.log 2>&1; tail -3 $O/tests.log
for cfg in "TCAVT_SK_U=4" "TCAVT_SK_U=9" "TCAVT_SK_U=8"; do
  for bs in 8 32; do
    echo -n "[$cfg] B=$bs " | tee -a $O/skinny.txt
    env $cfg timeout -k 10 200 python3 tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms_per_step'], d['decode_ms_per_step_min_max_of_5'])" | tee -a $O/skinny.txt
  done
done
cd /tmp
for cfg in "TCAVT_SK_U=4" "TCAVT_SK_U=8"; do
  export $cfg
  for bs in 8 32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gtx -o gen -- python3 $R/tools/bench_generate.py --batch $bs > /tmp/genx.log 2>&1
  echo "--- $cfg B=$bs" >> $O/breakdown.txt; python3 $R/tools/decode_breakdown.py /tmp/gtx/gen_kernel_trace.csv >> $O/breakdown.txt 2>&1
  rm -rf /tmp/gtx
  done
done
cat $O/breakdown.txt
