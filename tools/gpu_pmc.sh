#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the product build's passes through passbench: tools/gpu_pmc.sh <tag> "<sizes>"
set -o pipefail
TAG=$1; SIZES=$2
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
for S in $SIZES; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/fetch_$S -o f -- tools/microbench/passbench $PKG/libfdr.so $S 4 2 1 1 > gpurun_out/$TAG/fetch_$S.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/write_$S -o w -- tools/microbench/passbench $PKG/libfdr.so $S 4 2 1 1 > gpurun_out/$TAG/write_$S.log 2>&1
done
echo done
