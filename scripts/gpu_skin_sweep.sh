# usage: bash scripts/gpu_skin_sweep.sh "0.1 0.05 0.03"   (on the GPU box): bench per skin fraction
mkdir -p gpurun_out
for S in ${1:-0.1 0.05 0.03}; do
  echo "SKIN=$S"
  SPH_NL_SKIN=$S timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']/1e6,1),'M p-steps/s', round(j['ms_per_step'],2),'ms', j['passes_ms'], j['neighbour_lists'], 'queued', j['queued_targets_last_sweep'])
"
done
