#!/bin/bash
# Round-2 evidence in one GPU call: the driver-style bench line, the rocprofv3 --kernel-trace --stats summary of the SAME
# command, and the PMC passes (separate runs, --pmc with --kernel-trace only) of the dominant kernel.  Everything lands
# under gpurun_out/r02/; the summaries that are cited get copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -2 $O/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o step -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_step.log 2>&1; echo "rocprof rc=$?"
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable > $O/bench_lora.json 2> $O/bench_lora.err; tail -1 $O/bench_lora.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode forward > $O/bench_forward.json 2> $O/bench_forward.err; tail -1 $O/bench_forward.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch graph > $O/bench_graph.json 2> $O/bench_graph.err; tail -1 $O/bench_graph.err
for bs in 8 32; do python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b$bs.json; cut -c180-400 $O/generate_b$bs.json; done
for sh in gateup down o; do
  D=/tmp/pmc_r02_$sh; rm -rf $D; i=0
  for c in "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d $D/p$i -o p -- python3 $R/tools/pmc_gemm.py 0 $sh > $D.log 2>&1 || echo "pass failed: $sh $c"
  done
  python3 $R/tools/pmc_parse.py $D > $O/pmc_$sh.json; echo "pmc $sh done"
done
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lora-trainable --train-mllm-front > $O/bench_lora_full.json 2> $O/bench_lora_full.err; tail -1 $O/bench_lora_full.err
# LoRA-trainable step under the profiler (kernel stats + trace: tools/lora_timeline.py reads the trace)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lora -o lora -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --lora-trainable > $O/prof_lora.log 2>&1; echo "rocprof lora rc=$?"
python3 $R/tools/lora_timeline.py $O/prof_lora/lora_kernel_trace.csv > $O/lora_timeline.txt 2>&1; tail -3 $O/lora_timeline.txt
# two data-parallel ranks sharing the one card over gloo: rehearses the N > 1 code of bench.py / Trainer with GPU tensors
# (RCCL refuses two ranks on one device; the number is NOT a scaling figure)
cd $R && timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -2 $O/bench_gloo2.err
