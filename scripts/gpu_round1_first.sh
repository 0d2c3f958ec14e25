set -x
mkdir -p gpurun_out
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke1.log 2>&1; rc=$?; tail -5 gpurun_out/smoke1.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest1.log 2>&1; rc=$?; tail -40 gpurun_out/pytest1.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1; rc=$?; tail -5 gpurun_out/bench1.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof1.log 2>&1; rc=$?; tail -3 gpurun_out/prof1.log
find gpurun_out/prof1 -name "*stats*" | head
