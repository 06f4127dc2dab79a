#!/bin/bash
# Round-3 evidence in one GPU call: the driver-style bench line, the rocprofv3 --kernel-trace --stats summary of the SAME
# command, the PMC passes (separate runs, --pmc with --kernel-trace only) of the four decoder GEMMs in their in-model forms
# and of the attention kernel, the LoRA-trainable step, generation, and a two-rank gloo rehearsal of the N > 1 code.
# Everything lands under gpurun_out/r03/; the summaries that are cited get copied to profiles/ by hand afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -2 $O/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_step.log 2>&1; echo "rocprof rc=$?"
cp $O/prof_step/step_kernel_stats.csv $O/train_step_kernel_stats.csv; rm -rf $O/prof_step
for sh in gateup down o attn; do
  D=/tmp/pmc_r03_$sh; rm -rf $D; i=0
  for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $D/p$i -o p -- python3 $R/tools/pmc_gemm.py 0 $sh > $D.log 2>&1 || echo "pass failed: $sh $c"
  done
  python3 $R/tools/pmc_parse.py $D > $O/pmc_$sh.json; echo "pmc $sh done"
done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable > $O/bench_lora.json 2> $O/bench_lora.err; tail -1 $O/bench_lora.err
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable --train-mllm-front > $O/bench_lora_full.json 2> $O/bench_lora_full.err; tail -1 $O/bench_lora_full.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode forward > $O/bench_forward.json 2> $O/bench_forward.err; tail -1 $O/bench_forward.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > $O/bench_nopipeline.json 2> $O/bench_nopipeline.err; tail -1 $O/bench_nopipeline.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lora -o lora -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --lora-trainable > $O/prof_lora.log 2>&1; echo "rocprof lora rc=$?"
python3 $R/tools/lora_timeline.py $O/prof_lora/lora_kernel_trace.csv > $O/lora_timeline.txt 2>&1; tail -3 $O/lora_timeline.txt
cp $O/prof_lora/lora_kernel_stats.csv $O/lora_kernel_stats.csv; rm -rf $O/prof_lora
for bs in 8 32; do python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b$bs.json; cut -c1-300 $O/generate_b$bs.json; done
python3 $R/tools/ab_epilogue.py > $O/ab_epilogue.txt 2>&1; tail -9 $O/ab_epilogue.txt
cd $R && timeout -k 10 300 python3 bench.py --gpus 2 --steps 8 --warmup 2 --backend gloo --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -2 $O/bench_gloo2.err
