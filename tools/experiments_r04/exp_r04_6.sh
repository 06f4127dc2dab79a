set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp6; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_generation_gpu.py -x -q -m gpu -k "skinny or generation or decode or greedy or fragment" > $O/tests.log 2>&1; tail -2 $O/tests.log
for cfg in "X=1" "TCAVT_SK_NO_NT=1"; do
  for bs in 8 16 32; do
    echo -n "[$cfg] B=$bs " | tee -a $O/skinny.txt
    env $cfg timeout -k 10 200 python3 tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms_per_step'], d['decode_ms_per_step_min_max_of_5'])" | tee -a $O/skinny.txt
  done
done
export TCAVT_LIB=exp
for cfg in "X=1" "TCAVT_SK_XPACK_TIMING=1"; do
  for bs in 16 32; do
    echo -n "[exp $cfg] B=$bs " | tee -a $O/skinny.txt
    env $cfg timeout -k 10 200 python3 tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms_per_step'], d['decode_ms_per_step_min_max_of_5'])" | tee -a $O/skinny.txt
  done
done
