set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp11; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_model_parity_gpu.py tests/test_kernels_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "attn or lora or forward or config2 or stage" > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
for cfg in "X=1" "TCAVT_NO_WEIGHT_PREFETCH=1"; do
  echo -n "[$cfg] " | tee -a $O/out.txt
  env $cfg timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d['roofline']['frac'])" | tee -a $O/out.txt
done
done
