# rehearsal of bench.py --gpus 2 on a ONE-GPU box: gloo + both ranks on cuda:0 (the driver's real runs use nccl)
mkdir -p gpurun_out
export SPH_DIST_TRANSPORT=host SPH_DIST_SINGLE_DEVICE=1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --particles ${1:-1048576} > gpurun_out/dist_bench.log 2>&1
echo rc=$?; grep -v "Gloo\|socket.cpp\|amdgpu.ids" gpurun_out/dist_bench.log | tail -5 | cut -c1-1800
