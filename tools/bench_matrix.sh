#!/bin/bash
# A/B matrix of the batched fast path on one box: streams x images-per-B'-launch x column-kernel variant.
# usage: tools/bench_matrix.sh SIZE BATCH "s,g,lean s,g,lean ..."
S=${1:-4096}; B=${2:-16}; COMBOS=${3:-"1,1,0 2,1,0 1,2,0 1,4,0 2,2,0 2,4,0 1,1,1 2,1,1 1,4,1 2,4,1"}
for c in $COMBOS; do
  IFS=, read s g l <<< "$c"
  FDR_LEAN_COLS=$l FDR_COLS8=${COLS8:-0} python bench.py --size $S --batch $B --streams $s --group $g --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        j=json.loads(line); r=j['roofline']
        print('size=$S batch=$B streams=$s group=$g lean=$l cols8=${COLS8:-0}', j['value'], 'us/img=%.1f' % (j['ms_per_step']*1e3/$B), {k.split(':')[0]: round(v*1e3,1) for k,v in r['all_passes_ms'].items()})
"
done
