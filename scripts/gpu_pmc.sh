# usage: bash scripts/gpu_pmc.sh TAG "COUNTER LIST" [bench args]
TAG=$1; PMC=$2; shift; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ ! -f gpurun_out/counters_list.txt ]; then rocprofv3 -L > gpurun_out/counters_list.txt 2>&1; fi
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_$TAG.log 2>&1
echo rc=$?; tail -2 gpurun_out/pmc_$TAG.log | cut -c1-300; find gpurun_out/pmc_$TAG -name "*.csv" | head
