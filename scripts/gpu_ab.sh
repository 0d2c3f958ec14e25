# A/B of prebuilt libraries (sphfluidsimulation_amd/lib/variants/*.so, built here with SPH_HIPCC_EXTRA) on ONE box:
#   usage (on the GPU box): bash scripts/gpu_ab.sh [rounds] ; every library is benched `rounds` times, interleaved
mkdir -p gpurun_out
R=${1:-2}
for r in $(seq 1 $R); do
  for L in sphfluidsimulation_amd/lib/variants/*.so; do
    SPH_HIP_LIBRARY=$PWD/$L timeout -k 10 300 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/ab.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/ab.log; exit 1; }
    echo "$(basename $L) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab.log) $(grep -o '"density": [0-9.]*' gpurun_out/ab.log | head -1)" | tee -a gpurun_out/ab_results.txt
  done
done
# (BENCH_ARGS: extra bench.py arguments, e.g. "--mode tolerance")
