#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2f
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
PB=tools/microbench/passbench
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2f/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2f/status.txt )
tail -c 300 gpurun_out/r2f/tests.log
for cfg in "4096 24 10 2 2" "8192 6 6 3 2" "2048 32 10 2 4" "8192 6 6 1 1"; do
  timeout -k 10 120 $PB $PKG/libfdr.so $cfg >> gpurun_out/r2f/passbench.log 2>&1 || echo "FAILED $cfg" >> gpurun_out/r2f/passbench.log
done
timeout -k 10 120 $PB $PKG/build_dbg/libfdr_noswap.so 8192 6 6 3 2 >> gpurun_out/r2f/passbench.log 2>&1
for S in 8192; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2f/fetch_$S -o f -- $PB $PKG/libfdr.so $S 4 2 1 1 > gpurun_out/r2f/fetch_$S.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2f/write_$S -o w -- $PB $PKG/libfdr.so $S 4 2 1 1 > gpurun_out/r2f/write_$S.log 2>&1
done
grep -E "^==|us/image|batched|FAILED" gpurun_out/r2f/passbench.log | sed -e 's#/tmp/code/[^ ]*/##' -e 's/(real->complex)//' -e 's/launches)  */l) /' | cut -c1-150
