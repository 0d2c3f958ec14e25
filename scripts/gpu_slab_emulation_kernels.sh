# per-kernel time of an N-slab emulation on one GPU (rocprofv3 kernel trace of scripts/gpu_slab_emulation_profile.py N)
#   usage (on the GPU box): bash scripts/gpu_slab_emulation_kernels.sh N TAG
N=${1:-8}; TAG=${2:-slab8}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=${HWQ:-4} timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 scripts/gpu_slab_emulation_profile.py $N $OVERLAP > gpurun_out/prof_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_$TAG.log; exit 1; }
python3 - <<PY > gpurun_out/${TAG}_kernel_stats.txt
import csv, glob
for f in glob.glob("gpurun_out/prof_$TAG/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# $N slab(s), 15 frames: sum of kernel time {tot / 15e6:.2f} ms per frame = {tot / 15e6 / $N:.2f} ms per slab")
    print("kernel,calls_per_frame_per_slab,ms_per_frame,avg_us")
    for r in rows[:28]:
        print('"%s",%.1f,%.3f,%.1f' % (r["Name"][:90], int(r["Calls"]) / 15 / $N, float(r["TotalDurationNs"]) / 15e6, float(r["AverageNs"]) / 1e3))
PY
cat gpurun_out/${TAG}_kernel_stats.txt
rm -rf gpurun_out/prof_$TAG
