# usage: bash scripts/gpu_test_bench.sh TAG [pytest-args]   (runs on the GPU box via gpurun)
TAG=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" > gpurun_out/pytest_$TAG.log 2>&1; rc=$?; tail -15 gpurun_out/pytest_$TAG.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1; rc=$?; tail -3 gpurun_out/bench_$TAG.log
exit $rc
