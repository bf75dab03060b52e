#!/bin/bash
# usage: tools/profile_round.sh <tag> <splats> <steps> [extra bench.py flags, e.g. --four-d]   (on the GPU box, from the repo root)  -> gpurun_out/<tag>/
# one evidence set for profiles/: the bench line of the same command, rocprofv3 --kernel-trace --stats with the frame lanes overlapping and
# with one lane (every kernel alone), and the two --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, no trace domains) reduced by pmc_traffic.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; n=$2; steps=$3; warm=5; frames=$((steps+warm))
out=gpurun_out/$tag; mkdir -p $out
extra="${@:4}"
common="--splats $n --no-cpu-baseline --no-c3 --no-latency $extra"
python3 bench.py $common --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $common --steps $steps --warmup $warm --windows 1 > $out/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $common --steps $steps --warmup $warm --windows 1 > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $common --steps $steps --warmup $warm --windows 1 > $out/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $frames $out/pmc_traffic.json > $out/pmc_traffic.txt 2>&1
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
# the same kernels running alone (one lane: nothing overlaps): what each launch costs by itself
export GS4D_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1 -- python3 bench.py $common --steps $steps --warmup $warm --windows 1 --no-stage-events > $out/trace1.log 2>&1 || exit 1
unset GS4D_LANES
cp $(ls $out/trace1/*/*kernel_stats.csv | head -1) $out/kernel_stats_alone.csv
rm -rf $out/trace1 $out/trace $out/pmc_fetch $out/pmc_write
tail -c 700 $out/bench.json; echo; cat $out/pmc_traffic.txt
