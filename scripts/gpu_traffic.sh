# HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section)
#   usage (on the GPU box): bash scripts/gpu_traffic.sh TAG ["extra bench.py args"]
TAG=${1:-r01}
ARGS=${2:-}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras $ARGS > gpurun_out/pmc_${TAG}_$C.log 2>&1 || exit 1
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --no-cpu-baseline --no-extras $ARGS > gpurun_out/prof_${TAG}.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_${TAG}_{c}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            v.sort()
            out[k][c] = v[len(v) // 2]; out[k]["calls_" + c] = len(v)   # median: the first frame's launches differ (lists blocked)
with open("gpurun_out/traffic_${TAG}.csv", "w") as fo:
    fo.write("kernel,avg_FETCH_SIZE_raw,avg_WRITE_SIZE_raw,calls\n")   # (columns keep their names; values are per-launch medians)
    for k, v in sorted(out.items()):
        fo.write('"%s",%.1f,%.1f,%d\n' % (k, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), v.get("calls_FETCH_SIZE", 0)))
print(open("gpurun_out/traffic_${TAG}.csv").read())
for f in glob.glob("gpurun_out/prof_${TAG}/*/*kernel_stats.csv"):
    for r in csv.reader(open(f)): print(" | ".join(x[:50] for x in r[:5]))
PY
