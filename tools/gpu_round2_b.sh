#!/bin/bash
# GPU call 2: full GPU tests on the default build, then passbench over the variant builds
set -o pipefail
mkdir -p gpurun_out/r2b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
PB=tools/microbench/passbench
( timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2b/status.txt )
tail -c 400 gpurun_out/r2b/tests.log
for lib in $PKG/libfdr.so $PKG/build_dbg/libfdr_np.so $PKG/build_dbg/libfdr_pc2.so $PKG/build_dbg/libfdr_sc.so $PKG/build_dbg/libfdr_l4.so; do
  for cfg in "4096 24 10 3 1" "8192 6 6 3 1"; do
    timeout -k 10 120 $PB $lib $cfg >> gpurun_out/r2b/passbench.log 2>&1 || echo "FAILED $lib $cfg" >> gpurun_out/r2b/passbench.log
  done
done
for cfg in "2048 32 10 2 1" "2048 32 10 2 4" "1024 64 10 2 4" "512 64 10 2 4"; do
  timeout -k 10 120 $PB $PKG/libfdr.so $cfg >> gpurun_out/r2b/passbench.log 2>&1 || echo "FAILED default $cfg" >> gpurun_out/r2b/passbench.log
  timeout -k 10 120 $PB $PKG/build_dbg/libfdr_np.so $cfg >> gpurun_out/r2b/passbench.log 2>&1 || echo "FAILED np $cfg" >> gpurun_out/r2b/passbench.log
done
grep -E "^==|us/image|batched|FAILED|checksum" gpurun_out/r2b/passbench.log | cut -c1-200
