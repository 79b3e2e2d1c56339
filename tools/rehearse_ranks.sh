set -o pipefail
export GPSR_LOCAL_DEVICE=0 GPSR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 4 --steps 2 --warmup 1 > gpurun_out/rehearse4_set14.json 2> gpurun_out/rehearse4_set14.err || { tail -20 gpurun_out/rehearse4_set14.err; exit 1; }
tail -c 1500 gpurun_out/rehearse4_set14.json; echo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 4 --workload 4kx2 --steps 1 --warmup 1 --max-workspace-gb 12 > gpurun_out/rehearse4_4k.json 2> gpurun_out/rehearse4_4k.err || { tail -20 gpurun_out/rehearse4_4k.err; exit 1; }
tail -c 1800 gpurun_out/rehearse4_4k.json
