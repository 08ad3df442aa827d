#!/bin/bash
# panel-stride skew sweep: tools/gpu_padsweep.sh <tag> "<lib suffix>" "<cfg>" "<pads>"
set -o pipefail
TAG=$1; L=$2; CFG=$3; PADS=$4
mkdir -p gpurun_out/$TAG
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
if [ "$L" = "prod" ]; then lib=$PKG/libfdr.so; else lib=$PKG/build_dbg/libfdr_$L.so; fi
for pad in $PADS; do
  echo "## pad $pad" >> gpurun_out/$TAG/padsweep.log
  FDR_DEBUG_PSTRIDE_PAD=$pad timeout -k 10 120 tools/microbench/passbench $lib $CFG >> gpurun_out/$TAG/padsweep.log 2>&1 || echo "FAILED pad $pad" >> gpurun_out/$TAG/padsweep.log
done
grep -E "^##|us/image|batched|FAILED" gpurun_out/$TAG/padsweep.log | sed -e 's/(real->complex)//' | cut -c1-120
