#!/bin/bash
# A/B timing of kernel variants on one box: normal build, non-pipelined pass B', FFT-less (memory only) build
P="parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
for v in normal skipfft; do
  if [ $v = skipfft ]; then export FDR_LIB_PATH=$PWD/$P/build_dbg/libfdr_skipfft.so; else unset FDR_LIB_PATH; fi
  for np in 0 1; do
    FDR_NO_PIPELINE=$np python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'nopipe=$np', j['value'], j['roofline']['all_passes_ms'])
"
  done
done
