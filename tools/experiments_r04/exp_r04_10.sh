set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp10; mkdir -p $O
cd $R
for cfg in "X=1" "TCAVT_GEMM_NO_PERSIST=1"; do
  echo "== $cfg" | tee -a $O/out.txt
  env $cfg timeout -k 10 200 python3 tools/insitu_probe.py --part1-only 2>&1 | grep MLLM | tee -a $O/out.txt
  env $cfg timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d['roofline']['sustained_clock'])" | tee -a $O/out.txt
done
