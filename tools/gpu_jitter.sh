#!/bin/bash
# Race fuzzer run (on the GPU box, through gpurun): the GPU test suite against a -DFDR_DEBUG_JITTER build of the library
# (every wave sleeps 0..7 us before it touches shared LDS state; see fdr_fft_core.hpp).   tools/gpu_jitter.sh <tag> <variant> [pytest -k expr]
#   variant = jit     : the product code with jitter -- everything must pass
#   variant = jitbad  : the same with pass A's separation barrier left out -- the suite must FAIL (the fuzzer's own check)
# build first:  make OBJDIR=build_dbg/jit LIB=build_dbg/libfdr_jit.so EXTRA=-DFDR_DEBUG_JITTER   (in the package directory)
set -o pipefail
TAG=$1; VAR=$2; KEXPR=${3:-}
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp $PKG/libfdr.so /tmp/libfdr_product.so && cp $PKG/build_dbg/libfdr_$VAR.so $PKG/libfdr.so || exit 1
if [ -n "$KEXPR" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -k "$KEXPR" > gpurun_out/$TAG/tests_$VAR.log 2>&1
else
  timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/$TAG/tests_$VAR.log 2>&1
fi
rc=$?
cp /tmp/libfdr_product.so $PKG/libfdr.so
echo "jitter run $VAR rc=$rc" | tee -a gpurun_out/$TAG/status.txt
grep -E "passed|failed" gpurun_out/$TAG/tests_$VAR.log | tail -n 3
grep -E "^FAILED" gpurun_out/$TAG/tests_$VAR.log | cut -c1-160 | head -n 40
exit 0
