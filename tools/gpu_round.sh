#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_round.sh <tag> [pytest-args...]
# GPU test suite, then — unless the tests were killed — the C2 and C3 bench lines.  Everything goes to gpurun_out/<tag>/.
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q "$@" > $out/pytest.log 2>&1
rc=$?
tail -n 15 $out/pytest.log
if [ $rc -gt 1 ]; then echo "pytest rc=$rc: no further GPU steps"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err || exit $?
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --splats 10000000 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err || exit $?
python - <<PY
import json
for f in ("$out/bench_c2.json", "$out/bench_c3.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 4), "ms", "%.3g" % d["value"], "splats/s", d["roofline"]["frame"]["frac"] if d.get("roofline") else None, d["config"].get("tile_list_entries"), d["roofline"]["stage_ms_warmup_all_stages_timed"] if d.get("roofline") else None)
PY
exit $rc
