# quick look on the GPU box: list parity tests + the bench in both modes (no CPU baseline)
#   usage: bash scripts/gpu_quick.sh [pytest targets...]
mkdir -p gpurun_out
T=${@:-tests/test_gpu_lists.py}
timeout -k 10 900 python -m pytest $T -m gpu -x -q > gpurun_out/quick_tests.log 2>&1; tail -4 gpurun_out/quick_tests.log
python3 bench.py --steps 15 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/b_exact.log 2>&1
python3 bench.py --steps 15 --warmup 5 --no-extras --no-cpu-baseline --mode tolerance > gpurun_out/b_tol.log 2>&1
for f in gpurun_out/b_exact.log gpurun_out/b_tol.log; do
  echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"kernel_ms": [0-9.]*' $f) $(grep -o '"passes_ms": {[^}]*}' $f)"
done
