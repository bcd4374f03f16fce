set -u
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1600 --warmup 64 2>&1 | tail -1 | cut -c1-220
BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --workload colpart --colpart-vars 8192 --steps 256 --warmup 32 2>&1 | tail -1 | cut -c1-220
