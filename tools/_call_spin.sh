set -x
mkdir -p gpurun_out/spin
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/spin/b20.err | tail -1 > gpurun_out/spin/b20.json
python bench.py 2>gpurun_out/spin/bdef.err | tail -1 > gpurun_out/spin/bdef.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/spin/b20_second.json
python - <<'PY'
import json
for f in ("b20","bdef","b20_second"):
    d=json.load(open("gpurun_out/spin/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d.get("steady_state_pivots_per_s"), d.get("steady_state_pivots_per_s_gpu_clock"), d.get("host_wait_after_gpu_ms"), d["roofline"]["frac"])
PY
python tools/steady_gap.py --repeat 3 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
