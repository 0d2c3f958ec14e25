mkdir -p gpurun_out
for A in 1 2 3 0; do
  SPH_HIPCC_EXTRA="-DSPH_ABLATE=$A" python -m sphfluidsimulation_amd.build --force > /dev/null 2>&1
  echo "ABLATE=$A"; timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"passes_ms": {[^}]*}'
done
