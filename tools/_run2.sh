cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python tools/shard_la_timing.py 1 > gpurun_out/shard_la_timing_hop1.txt 2>&1; echo rc=$? >> gpurun_out/shard_la_timing_hop1.txt)
(timeout 900 python tools/shard_la_timing.py 0 > gpurun_out/shard_la_timing_hop0.txt 2>&1; echo rc=$? >> gpurun_out/shard_la_timing_hop0.txt)
cat gpurun_out/shard_la_timing_hop1.txt gpurun_out/shard_la_timing_hop0.txt
