# usage: bash scripts/gpu_variants.sh "FLAGS1" "FLAGS2" ...   (each: extra hipcc -D flags; quick bench only)
mkdir -p gpurun_out
for V in "$@"; do
  SPH_HIPCC_EXTRA="$V" python -m sphfluidsimulation_amd.build --force > gpurun_out/variant_build.log 2>&1 || { echo "BUILD FAILED: $V"; tail -5 gpurun_out/variant_build.log; continue; }
  echo "VARIANT: $V"; timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o -E '"value": [0-9.e+]*|"passes_ms": {[^}]*}|"queued_targets_last_sweep": [0-9]*' | tr '\n' ' '; echo
done
