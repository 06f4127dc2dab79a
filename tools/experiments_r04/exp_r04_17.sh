set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp17; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_llm_backward_gpu.py tests/test_training_gpu.py tests/test_dp_gpu.py tests/test_reference_grads_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
for cfg in "X=1" "TCAVT_LORA_NO_DEFER=1"; do
  echo -n "[$cfg] lora " | tee -a $O/out.txt
  env $cfg timeout -k 10 300 python3 bench.py --lora-trainable --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d.get('optimizer_updates'))" | tee -a $O/out.txt
done
done
for cfg in "X=1" "TCAVT_LORA_NO_DEFER=1"; do
  echo -n "[$cfg] lora full " | tee -a $O/out.txt
  env $cfg timeout -k 10 300 python3 bench.py --lora-trainable --train-mllm-front --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d.get('optimizer_updates'))" | tee -a $O/out.txt
done
