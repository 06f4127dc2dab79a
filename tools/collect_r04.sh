#!/bin/bash
# Round-4 evidence, two GPU calls (the whole of it exceeds one call's time limit):
#   collect_r04.sh a : the driver-style bench line, the rocprofv3 --kernel-trace --stats summary of the SAME command, the PMC passes
#                      (separate runs, --pmc with --kernel-trace only) of the decoder GEMMs in their in-model forms and of the attention
#   collect_r04.sh b : the variants (forward, --no-pipeline, bf16 storage, --feed host, LoRA-trainable, whole modify_train.py set),
#                      the LoRA-trainable step's per-queue timeline, generation + decode-step breakdown, config-4 sweep, config-5
#                      evaluation, the MLLM stream's busy / idle time, a two-rank gloo rehearsal of the N > 1 code
# Everything lands under gpurun_out/r04/; the summaries that are cited get copied to profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
part=${1:-a}
if [ "$part" = "a" ]; then
  python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -2 $O/bench_default.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_step.log 2>&1; echo "rocprof rc=$?"
  cp $O/prof_step/step_kernel_stats.csv $O/train_step_kernel_stats.csv; rm -rf $O/prof_step
  for sh in gateup down o attn; do
    D=/tmp/pmc_r04_$sh; rm -rf $D; i=0
    for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
             "SQ_INSTS_VALU SQ_WAIT_ANY"; do
      i=$((i+1))
      timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $D/p$i -o p -- python3 $R/tools/pmc_gemm.py 0 $sh > $D.log 2>&1 || echo "pass failed: $sh $c"
    done
    python3 $R/tools/pmc_parse.py $D > $O/pmc_$sh.json; echo "pmc $sh done"
  done
  exit 0
fi
for v in "forward:--mode forward" "nopipeline:--no-pipeline" "bf16:--storage bf16" "feed_host:--feed host" "rccl1:--rccl-self-test"; do
  n=${v%%:*}; f=${v#*:}
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $f > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n: $(tail -1 $O/bench_$n.err)"
done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable > $O/bench_lora.json 2> $O/bench_lora.err; tail -1 $O/bench_lora.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable --train-mllm-front > $O/bench_lora_full.json 2> $O/bench_lora_full.err; tail -1 $O/bench_lora_full.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable --storage bf16 > $O/bench_lora_bf16.json 2> $O/bench_lora_bf16.err; tail -1 $O/bench_lora_bf16.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lora -o lora -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --lora-trainable > $O/prof_lora.log 2>&1; echo "rocprof lora rc=$?"
python3 $R/tools/lora_timeline.py $O/prof_lora/lora_kernel_trace.csv > $O/lora_timeline.txt 2>&1; head -3 $O/lora_timeline.txt
cp $O/prof_lora/lora_kernel_stats.csv $O/lora_kernel_stats.csv; rm -rf $O/prof_lora
for bs in 8 16 32; do python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b$bs.json; cut -c1-330 $O/generate_b$bs.json; done
python3 $R/tools/bench_generate.py --batch 8 --greedy 2>/dev/null | tail -1 > $O/generate_b8_greedy.json
TCAVT_SAMPLE_ONE_STAGE=1 python3 $R/tools/bench_generate.py --batch 8 2>/dev/null | tail -1 > $O/generate_b8_one_stage_sampler.json
for bs in 8 32; do TCAVT_DECODE_ROWMAJOR=1 python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b${bs}_row_major_weights.json; done
TCAVT_DECODE_ACT_ROWMAJOR=1 python3 $R/tools/bench_generate.py --batch 32 2>/dev/null | tail -1 > $O/generate_b32_row_major_activations.json
for bs in 8 32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt$bs -o gen -- python3 $R/tools/bench_generate.py --batch $bs > /tmp/gen$bs.log 2>&1
  echo "--- B = $bs" >> $O/decode_step_breakdown.txt; python3 $R/tools/decode_breakdown.py /tmp/gt$bs/gen_kernel_trace.csv >> $O/decode_step_breakdown.txt 2>&1
done
python3 $R/tools/pipe_trace.py 2>/dev/null | tail -1 > $O/pipe_trace.txt; cat $O/pipe_trace.txt
python3 $R/tools/bench_sampler.py > $O/bench_sampler.txt 2>&1
python3 $R/tools/bench_eval_k.py 512 > $O/config5_eval_k.txt 2>&1; tail -3 $O/config5_eval_k.txt
bash $R/tools/sweep_config4.sh > $O/config4_sweep.txt 2>&1; tail -4 $O/config4_sweep.txt
cd $R && timeout -k 10 300 python3 bench.py --gpus 2 --steps 8 --warmup 2 --backend gloo --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -2 $O/bench_gloo2.err
