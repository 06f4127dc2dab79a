set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp9; mkdir -p $O
cd $R
timeout -k 10 500 python3 tools/insitu_probe.py > $O/insitu.txt 2>&1
grep -v amdgpu.ids $O/insitu.txt | tail -12
