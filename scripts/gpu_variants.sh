# (BENCH_ARGS: extra bench.py arguments)
# usage: bash scripts/gpu_variants.sh "flags1" "flags2" ...   (on the GPU box): bench of each SPH_HIPCC_EXTRA variant
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for V in "$@"; do
  i=$((i+1))
  SPH_HIPCC_EXTRA="$V" python -m sphfluidsimulation_amd.build --force > gpurun_out/variant_build_$i.log 2>&1 || { echo "BUILD FAILED: $V"; tail -5 gpurun_out/variant_build_$i.log; continue; }
  echo "VARIANT [$V]"
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/variant_$i.log 2>&1 || { echo failed; tail -3 gpurun_out/variant_$i.log; continue; }
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/variant_$i.log | head -1
  grep -o '"kernel_ms": [0-9.]*' gpurun_out/variant_$i.log
  grep -o '"passes_ms": {[^}]*}' gpurun_out/variant_$i.log
done
