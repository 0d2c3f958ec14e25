# SQ counters of the LATE frames of a 300-frame dam break (one --pmc pass; kernel-trace only beside it)
#   usage (on the GPU box): bash scripts/late_phase_pmc.sh TAG [exact|tolerance] [first_frame] [frames] ["COUNTERS"]
TAG=${1:-late}; MODE=${2:-exact}; F0=${3:-276}; FRAMES=${4:-300}
CNT=${5:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d gpurun_out/latepmc_$TAG -- python3 scripts/long_run.py 16777216 $FRAMES 25 lists 0 0 $MODE > gpurun_out/latepmc_$TAG.log 2>&1 || { tail -5 gpurun_out/latepmc_$TAG.log; exit 1; }
python3 - <<PY > gpurun_out/latepmc_${TAG}.csv
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/latepmc_$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
frame = 0; last = None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for did, k, c, v in rows:
    if "k_cell_key_hist" in k and did != last:
        frame += 1; last = did
    if frame >= $F0:
        agg[k][c].append(v)
names = sorted({c for k in agg for c in agg[k]})
print("kernel,launches," + ",".join(names) + ",valu_insts_per_launch_x3.3cyc_over_1024simd_ms_at_2.4GHz")
for k in sorted(agg, key=lambda k: -sum(agg[k].get("SQ_INSTS_VALU", [0]))):
    m = {c: sum(agg[k][c]) / max(len(agg[k][c]), 1) for c in names}
    n = max(len(v) for v in agg[k].values())
    print('"%s",%d,' % (k[:80], n) + ",".join("%.0f" % m[c] for c in names) + ",%.3f" % (m.get("SQ_INSTS_VALU", 0) * 3.3 / 1024 / 2.4e6))
PY
cat gpurun_out/latepmc_${TAG}.csv
rm -rf gpurun_out/latepmc_$TAG
