cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 bench.py --no-cpu-baseline --no-c3 --no-latency --steps 20 --warmup 10 --windows 9 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],5), b['windows_ms_per_step'][:5])"; }
run GS4D_LANES=4
run GS4D_LANES=4 GS4D_ABLATE_SORT=1
run GS4D_LANES=4
run GS4D_LANES=4 GS4D_ABLATE_SORT=1
run GS4D_LANES=4 GS4D_ABLATE_COMPOSITE=1
run GS4D_LANES=4 GS4D_ABLATE_COMPOSITE=1 GS4D_ABLATE_SORT=1
run GS4D_LANES=1
run GS4D_LANES=1 GS4D_ABLATE_SORT=1
run GS4D_LANES=1 GS4D_ABLATE_COMPOSITE=1
