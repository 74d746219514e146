export LSA_BENCH_SINGLE_DEVICE=1 LSA_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 2>&1 | grep -v "amdgpu.ids\|W1002\|^\*\*\*\|OMP_NUM" | tail -5
