for w in 2 1 0 2 0; do echo "== wait=$w"; python tools/steady_gap.py --repeat 5 --pivots 4200 --wait $w 2>&1 | grep load= | cut -c1-215; done
for w in 2 0; do echo "== wait=$w pivots=20"; python tools/steady_gap.py --repeat 5 --pivots 20 --events 0 --wait $w 2>&1 | grep load= | cut -c1-215; done
echo "== bench x3"
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['other_configs']['cfg3_steady_state']; print('b20', d['value'], d['ms_per_step'], s['value'], s['gpu_clock']['pivots_per_s'], s['host_wait_after_gpu_ms'])"; done
