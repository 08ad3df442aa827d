#!/bin/bash
# first GPU call of round 2: full GPU test suite, default bench, baseline profiles at 8192^2 (kernel trace + PMC)
set -o pipefail
mkdir -p gpurun_out/r2a
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2a/status.txt
python bench.py > gpurun_out/r2a/bench_default.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/r2a/status.txt
tail -c 600 gpurun_out/r2a/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a/kt8192 -o kt -- python3 bench.py --size 8192 --batch 8 --steps 5 --warmup 2 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute > gpurun_out/r2a/kt8192.log 2>&1; echo "kt8192 rc=$?" | tee -a gpurun_out/r2a/status.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2a/fetch8192 -o fetch -- python3 bench.py --size 8192 --batch 4 --steps 2 --warmup 1 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute > gpurun_out/r2a/fetch8192.log 2>&1; echo "fetch rc=$?" | tee -a gpurun_out/r2a/status.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2a/write8192 -o write -- python3 bench.py --size 8192 --batch 4 --steps 2 --warmup 1 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute > gpurun_out/r2a/write8192.log 2>&1; echo "write rc=$?" | tee -a gpurun_out/r2a/status.txt
grep '^{' gpurun_out/r2a/bench_default.log | tail -n 1 | cut -c1-1500
