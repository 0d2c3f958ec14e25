# L1 (TCP) / texture-addresser counters of every kernel of a bench run: how many cache-line look-ups the gathers cost.
#   usage (on the GPU box): bash scripts/gpu_pmc_tcp.sh TAG   -> gpurun_out/tcp_TAG.csv (per-launch medians)
TAG=${1:-r02}
ARGS=${2:-}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASS_A="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
PASS_B="TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
PASS_C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
i=0
for P in "$PASS_A" "$PASS_B" "$PASS_C"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/tcp_${TAG}_$i -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras $ARGS > gpurun_out/tcp_${TAG}_$i.log 2>&1 || { tail -5 gpurun_out/tcp_${TAG}_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/tcp_${TAG}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in vals for c in vals[k]})
def med(v):
    v = sorted(v); return v[len(v) // 2]
with open("gpurun_out/tcp_${TAG}.csv", "w") as fo:
    fo.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(vals):
        m = {c: med(vals[k][c]) if c in vals[k] else 0.0 for c in names}
        fo.write('"%s",%d,' % (k[:90], max(len(v) for v in vals[k].values())) + ",".join("%.0f" % m[c] for c in names) + "\n")
print(open("gpurun_out/tcp_${TAG}.csv").read())
PY
