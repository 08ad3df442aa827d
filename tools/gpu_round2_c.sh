#!/bin/bash
# GPU call 3: full GPU tests (with the slab rehearsals), passbench over default / non-persistent / 16-value row variants
set -o pipefail
mkdir -p gpurun_out/r2c
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
PB=tools/microbench/passbench
( timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2c/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2c/status.txt )
tail -c 600 gpurun_out/r2c/tests.log
for lib in $PKG/libfdr.so $PKG/build_dbg/libfdr_np.so $PKG/build_dbg/libfdr_l4.so; do
  for cfg in "4096 24 10 3 1" "8192 6 6 3 1" "2048 32 10 2 4"; do
    timeout -k 10 120 $PB $lib $cfg >> gpurun_out/r2c/passbench.log 2>&1 || echo "FAILED $lib $cfg" >> gpurun_out/r2c/passbench.log
  done
done
timeout -k 10 120 $PB $PKG/libfdr.so 4096 24 10 3 2 >> gpurun_out/r2c/passbench.log 2>&1
timeout -k 10 120 $PB $PKG/libfdr.so 2048 32 10 2 1 >> gpurun_out/r2c/passbench.log 2>&1
timeout -k 10 120 $PB $PKG/libfdr.so 1024 64 10 2 4 >> gpurun_out/r2c/passbench.log 2>&1
grep -E "^==|us/image|batched|FAILED|checksum" gpurun_out/r2c/passbench.log | sed -e 's#/tmp/code/[^ ]*/##' | cut -c1-170
