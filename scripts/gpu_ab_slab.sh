# A/B of prebuilt libraries on the serialised 8-slab emulation (GPU_MAX_HW_QUEUES=1: the slabs' kernels run one after the other)
mkdir -p gpurun_out
for L in sphfluidsimulation_amd/lib/variants/*.so; do
  SPH_HIP_LIBRARY=$PWD/$L GPU_MAX_HW_QUEUES=1 timeout -k 10 300 python3 scripts/gpu_slab_emulation.py 16777216 ${1:-overlap} > gpurun_out/ab_slab.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/ab_slab.log; exit 1; }
  echo "$(basename $L) $(tail -1 gpurun_out/ab_slab.log | cut -c1-60)"
done
