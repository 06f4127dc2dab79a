#!/bin/bash
# BASELINE.json configs[3] (SURVEY 8d "Config 4"): the no-LoRA variant, B x L sweep of the train step and the forward,
# one bench.py run per point; prints  B L mode ms/step trajectories/s model-TFLOP/s gate|up-TFLOP/s
cd "$(dirname "$0")/.."
for L in 128 256 512; do
  for B in 8 16 32 64; do
    for mode in train forward; do
      timeout -k 10 300 python bench.py --no-lora --mode $mode --batch $B --text-len $((L - 16)) --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(c['per_gpu_batch'], c['fused_seq_len'], c['mode'], d['ms_per_step'], d['value'], d['achieved_model_tflops'], d['roofline']['achieved'])"
    done
  done
done
