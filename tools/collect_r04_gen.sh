# the generation lines of tools/collect_r04.sh alone (re-run after the last decode-step change)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for bs in 4 8 16 32; do python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b$bs.json; cut -c1-330 $O/generate_b$bs.json; done
python3 $R/tools/bench_generate.py --batch 8 --greedy 2>/dev/null | tail -1 > $O/generate_b8_greedy.json
TCAVT_SAMPLE_ONE_STAGE=1 python3 $R/tools/bench_generate.py --batch 8 2>/dev/null | tail -1 > $O/generate_b8_one_stage_sampler.json
for bs in 8 32; do TCAVT_DECODE_ROWMAJOR=1 python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b${bs}_row_major_weights.json; done
for bs in 8 32; do TCAVT_DECODE_ACT_ROWMAJOR=1 python3 $R/tools/bench_generate.py --batch $bs 2>/dev/null | tail -1 > $O/generate_b${bs}_row_major_activations.json; done
rm -f $O/decode_step_breakdown.txt
cd /tmp
for bs in 8 32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt$bs -o gen -- python3 $R/tools/bench_generate.py --batch $bs > /tmp/gen$bs.log 2>&1
  echo "--- B = $bs" >> $O/decode_step_breakdown.txt; python3 $R/tools/decode_breakdown.py /tmp/gt$bs/gen_kernel_trace.csv >> $O/decode_step_breakdown.txt 2>&1
done
cat $O/decode_step_breakdown.txt
