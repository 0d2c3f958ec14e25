# kernel-level breakdown of the LATE frames of a 300-frame dam break (the collapse phase): rocprofv3 kernel trace of
# scripts/long_run.py, launches grouped by frame (one k_cell_key_hist per frame)
#   usage (on the GPU box): bash scripts/late_phase_profile.sh TAG [exact|tolerance] [first_frame] [frames]
TAG=${1:-late}; MODE=${2:-exact}; F0=${3:-276}; FRAMES=${4:-300}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/late_$TAG -- python3 scripts/long_run.py 16777216 $FRAMES 25 lists 0 0 $MODE > gpurun_out/late_$TAG.log 2>&1 || { tail -5 gpurun_out/late_$TAG.log; exit 1; }
cat gpurun_out/late_$TAG.log
python3 scripts/late_phase_summary.py gpurun_out/late_$TAG $F0 > gpurun_out/late_${TAG}_kernels.txt
cat gpurun_out/late_${TAG}_kernels.txt
rm -rf gpurun_out/late_$TAG
