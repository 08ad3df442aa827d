#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and the two HBM PMC passes for bench.py,
# plus the same two PMC passes on a known-size float4 copy (calibration of FETCH_SIZE on gfx950).
# Counters are collected in their own runs (no --kernel-trace / --stats beside --pmc).
#   usage: tools/collect_profiles.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --no-cpu-baseline "$@" > $OUT/kt.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --batch 4 --streams 1 --no-cpu-baseline > $OUT/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 bench.py --steps 2 --warmup 1 --batch 4 --streams 1 --no-cpu-baseline > $OUT/write.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cal -- ./tools/microbench/membench > $OUT/cal_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cal -- ./tools/microbench/membench > $OUT/cal_write.log 2>&1
grep '^{' $OUT/kt.log | tail -n 1 > $OUT/bench_line.json
