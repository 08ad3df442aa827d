#!/bin/bash
# Issue-side attribution of the passes' wave-cycles: which counters does `rocprofv3 -L` offer on this
# gfx950 box for instruction fetch, VMEM address / data path back-pressure, LDS issue and VALU dependency, and what do
# they read for each pass?  Counters are collected in their own passes (no trace options beside --pmc), through the
# torch-free passbench so that a pass costs a second, not a Python start-up.
#   usage: tools/gpu_issue_pmc.sh <tag> "<passbench cfg>;<passbench cfg>;..."   e.g. "4096 8 3 1 4;4096 16 3 2 4;8192 4 3 1 2"
set -o pipefail
TAG=$1; CFGS=${2:-"4096 8 3 1 4"}; LIBSFX=${3:-prod}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
if [ "$LIBSFX" = "prod" ]; then LIB=$PKG/libfdr.so; else LIB=$PKG/build_dbg/libfdr_$LIBSFX.so; fi
make -s -C tools/microbench passbench
rocprofv3 -L > $OUT/counters_available.txt 2>&1; echo "list rc=$?"
python3 tools/issue_pmc_sets.py $OUT/counters_available.txt > $OUT/sets.txt
cat $OUT/sets.txt | head -n 60
IFS=';' read -ra CF <<< "$CFGS"
c=0
for cfg in "${CF[@]}"; do
  c=$((c+1)); i=0
  while read -r SET; do
    [ -z "$SET" ] && continue
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $SET --output-format csv -d $OUT/c${c}_p$i -o pmc -- tools/microbench/passbench $LIB $cfg > $OUT/c${c}_p$i.log 2>&1
    echo "cfg$c [$cfg] pass$i rc=$? : $SET"
  done < $OUT/sets.txt
  python3 tools/issue_pmc_report.py $OUT "c${c}_p" > $OUT/report_c$c.txt 2>&1
  echo "== cfg $c: $cfg" ; cat $OUT/report_c$c.txt
done
