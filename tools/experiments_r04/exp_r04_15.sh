set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp15; mkdir -p $O
cd $R
timeout -k 10 100 python3 tools/bench_lora_down.py 2>&1 | grep lora_down | tee -a $O/out.txt
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_llm_backward_gpu.py -x -q -m gpu -k "lora" > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $O/out.txt
