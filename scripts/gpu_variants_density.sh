# usage: bash scripts/gpu_variants_density.sh "flags1" ... (on the GPU box): density pass time of each SPH_HIPCC_EXTRA variant
mkdir -p gpurun_out
i=0
for V in "$@"; do
  i=$((i+1))
  SPH_HIPCC_EXTRA="$V" python -m sphfluidsimulation_amd.build --force > gpurun_out/variant_build_$i.log 2>&1 || { echo "BUILD FAILED: $V"; tail -5 gpurun_out/variant_build_$i.log; continue; }
  echo "VARIANT [$V]"
  timeout -k 10 300 python3 scripts/density_time.py 2>&1 | grep "lists True"
done
