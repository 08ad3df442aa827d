#!/bin/bash
# generic passbench call: tools/gpu_pb.sh <tag> "<lib suffixes>" "<cfg1>;<cfg2>;..."   (lib suffix "" = product build)
set -o pipefail
TAG=$1; LIBS=$2; CFGS=$3
mkdir -p gpurun_out/$TAG
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
IFS=';' read -ra CF <<< "$CFGS"
for l in $LIBS; do
  if [ "$l" = "prod" ]; then lib=$PKG/libfdr.so; else lib=$PKG/build_dbg/libfdr_$l.so; fi
  for cfg in "${CF[@]}"; do
    timeout -k 10 120 tools/microbench/passbench $lib $cfg >> gpurun_out/$TAG/passbench.log 2>&1 || echo "FAILED $lib $cfg" >> gpurun_out/$TAG/passbench.log
  done
done
grep -E "^==|us/image|batched|FAILED" gpurun_out/$TAG/passbench.log | sed -e 's#/tmp/code/[^ ]*/##' -e 's/(real->complex)//' -e 's/launches)  */l) /' | cut -c1-150
