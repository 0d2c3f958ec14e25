# usage: bash scripts/gpu_ablate_list.sh "5 6 7 0" ["-DSPH_LIST_MINWAVES=5 ..."]  (on the GPU box)
# -> time of the list kernels per SPH_ABLATE variant (5: no entry loop, 6: coalesced gathers + full arithmetic,
#    7: real gathers without the pair arithmetic, 0: production)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for A in ${1:-5 6 7 0}; do
  SPH_HIPCC_EXTRA="-DSPH_ABLATE=$A $2" python -m sphfluidsimulation_amd.build --force > /dev/null 2>&1
  echo "ABLATE=$A $2"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$A -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/abl_$A.log 2>&1 || exit 1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/abl_$A.log
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/abl_$A/*/*kernel_stats.csv"):
    for r in csv.reader(open(f)):
        if "velpos_list" in r[0] or "Name" in r[0]: print(" | ".join(x[:40] for x in r[:7]))
PY
done
