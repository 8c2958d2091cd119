mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullscale.py tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/r05/pytest_part.log 2>&1; grep -E "passed|failed" gpurun_out/r05/pytest_part.log
for v in sort hash sort hash; do FFH_SLAB_TOTALS=$v timeout 900 python bench.py --no-traffic --cpu-seconds 0 --no-verify --no-c2 --steps 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d.get('skewed', {})
print('$v', round(d['ms_per_step'], 3), 'skewed', round(s.get('ms_per_step', 0), 3), {k: round(v, 3) for k, v in s.get('breakdown_ms', {}).items()}, s.get('raw_hits'), s.get('kept_hits'))"; done
