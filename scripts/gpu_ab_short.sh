# like gpu_ab.sh with a SHORT run (frames 3-5): for diagnostic builds whose physics is wrong (ablations)
mkdir -p gpurun_out
for L in sphfluidsimulation_amd/lib/variants/*.so; do
  SPH_HIP_LIBRARY=$PWD/$L timeout -k 10 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/ab.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/ab.log; exit 1; }
  echo "$(basename $L) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab.log) $(grep -o '"density": [0-9.]*' gpurun_out/ab.log | head -1)"
done
