# usage: bash scripts/gpu_prof.sh TAG [bench args]  -> rocprofv3 kernel stats of a short bench run
TAG=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_$TAG.log 2>&1
echo rc=$?; grep '^{' gpurun_out/prof_$TAG.log | cut -c1-2000
python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_$TAG/*/*kernel_stats.csv"):
    for r in csv.reader(open(f)): print(" | ".join(x[:44] for x in r[:7]))
PY
