bash tools/probe_partition.sh
mkdir -p gpurun_out/gap
python tools/steady_gap.py --repeat 3 > gpurun_out/gap/unloaded.log 2>&1
python tools/steady_gap.py --repeat 3 --events 0 > gpurun_out/gap/unloaded_noev.log 2>&1
python tools/steady_gap.py --repeat 3 --load 1 > gpurun_out/gap/load1.log 2>&1
python tools/steady_gap.py --repeat 3 --load 2 > gpurun_out/gap/load2.log 2>&1
python tools/steady_gap.py --repeat 3 --load 2 --events 0 > gpurun_out/gap/load2_noev.log 2>&1
tail -n 4 gpurun_out/gap/*.log
nproc; uptime
