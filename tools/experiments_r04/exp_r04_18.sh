set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp18; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_dp_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err; tail -1 $O/bench_plain.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain step', d['ms_per_step'])"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rccl-self-test > $O/bench_rccl1.json 2> $O/bench_rccl1.err; tail -1 $O/bench_rccl1.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-rank RCCL step', d['ms_per_step'], json.dumps(d.get('dp_diagnostics'))[:900])"
tail -3 $O/bench_rccl1.err
