set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp12; mkdir -p $O
cd $R
for B in 8 16; do
for tile in 0 64 128; do
  echo -n "B=$B L=128 tile=$tile " | tee -a $O/out.txt
  timeout -k 10 200 python3 bench.py --no-lora --mode forward --batch $B --text-len 112 --steps 10 --warmup 3 --no-cpu-baseline --tile $tile 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a $O/out.txt
done
done
