# usage: bash scripts/gpu_variants2.sh "<flags1>" "<flags2>" ...   (on the GPU box): bench per SPH_HIPCC_EXTRA variant
mkdir -p gpurun_out
for V in "$@"; do
  SPH_HIPCC_EXTRA="$V" python -m sphfluidsimulation_amd.build --force > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  echo "VARIANT: $V"
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']/1e6,1),'M p-steps/s', round(j['ms_per_step'],2),'ms', {k:round(v,3) for k,v in j['passes_ms'].items()}, 'queued', j['queued_targets_last_sweep'])
"
done
