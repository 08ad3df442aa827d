#!/bin/bash
# decomposition of pass B' with the timing-only builds (memory only / compute only) next to the product build
P="$PWD/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
for v in normal skipfft skipmem; do
  if [ $v = normal ]; then unset FDR_LIB_PATH; else export FDR_LIB_PATH=$P/build_dbg/libfdr_$v.so; fi
  echo "== $v"; tools/bench_matrix.sh "$@"
done
