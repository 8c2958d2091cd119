# robustness sweep of bench.py argument combinations (each must print one JSON line)
set -u
run() { echo "== $*"; timeout 300 python bench.py "$@" --no-traffic --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), '%.3g' % d['value'], d['hits'], d['plan'])" || echo FAILED; }
run --steps 1 --warmup 0
run --guides 1000 --targets 4.5e6 --workload chr22-scale
run --guides 1 --targets 1e5
run --guides 100000 --targets 3e8 --max-mismatch 5 --steps 2
run --guides 100000 --targets 3e8 --max-mismatch 3
run --guides 5000 --targets 3e8 --max-mismatch 0
run --guides 100000 --targets 3e8 --max-offtargets 10
run --guides 300000 --targets 1e8 --steps 2
