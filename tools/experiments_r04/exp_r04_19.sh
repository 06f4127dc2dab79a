set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp19; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_generation_gpu.py tests/test_range_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "skinny or generation or decode or greedy or fragment or token or eos or sampl" > $O/tests.log 2>&1; tail -3 $O/tests.log
for cfg in "X=1" "TCAVT_DECODE_ACT_ROWMAJOR=1" "TCAVT_DECODE_ACT_FRAG=16"; do
  for bs in 4 8; do
    echo -n "[$cfg] B=$bs " | tee -a $O/out.txt
    env $cfg timeout -k 10 200 python3 tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms_per_step'], d['decode_ms_per_step_min_max_of_5'])" | tee -a $O/out.txt
  done
done
