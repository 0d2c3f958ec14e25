# SQ (shader sequencer) counters of every kernel of a bench run, two --pmc passes of 8 SQ counters each
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots"; never combined with tracing other than --kernel-trace).
#   usage (on the GPU box): bash scripts/gpu_pmc_sq.sh TAG ["extra bench.py args"]
#   -> gpurun_out/sq_TAG.csv : per kernel, per-launch MEDIANS of every counter + derived fractions
TAG=${1:-r02}
ARGS=${2:-}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASS_A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS_B="SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD"
i=0
for P in "$PASS_A" "$PASS_B"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/sq_${TAG}_$i -- python3 bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-3} --no-cpu-baseline --no-extras $ARGS > gpurun_out/sq_${TAG}_$i.log 2>&1 || { tail -5 gpurun_out/sq_${TAG}_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_${TAG}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in vals for c in vals[k]})
def med(v):
    v = sorted(v); return v[len(v) // 2]
with open("gpurun_out/sq_${TAG}.csv", "w") as fo:
    fo.write("kernel,launches," + ",".join(names) + ",valu_active_over_busy,valu_insts_per_wave,wait_inst_any_frac,wait_any_frac,trans_share_of_valu\n")
    for k in sorted(vals):
        m = {c: med(vals[k][c]) if c in vals[k] else 0.0 for c in names}
        n = max(len(v) for v in vals[k].values())
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1.0
        d = [m.get("SQ_ACTIVE_INST_VALU", 0) / (m.get("SQ_BUSY_CYCLES", 0) or 1.0), m.get("SQ_INSTS_VALU", 0) / (m.get("SQ_WAVES", 0) or 1.0),
             m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_INSTS_VALU_TRANS_F32", 0) / (m.get("SQ_INSTS_VALU", 0) or 1.0)]
        fo.write('"%s",%d,' % (k[:90], n) + ",".join("%.0f" % m[c] for c in names) + "," + ",".join("%.4f" % x for x in d) + "\n")
print(open("gpurun_out/sq_${TAG}.csv").read())
PY
