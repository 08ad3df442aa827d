#!/bin/bash
# SQ counter passes (stall / activity / LDS conflicts) of bench.py --streams 1 at one size: tools/gpu_sqpmc.sh <tag> <size> <batch>
set -o pipefail
TAG=$1; S=${2:-4096}; B=${3:-8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--size $S --steps 2 --warmup 1 --batch $B --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute --no-parity-leg --no-batch-check"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_VMEM" "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/sq$i -o sq -- python3 bench.py $ARGS > $OUT/sq$i.log 2>&1; echo "sq$i rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fdr::fft" not in k: continue
        name = k.replace("void fdr::", "").split("(")[0][:44] + " g" + r["Grid_Size"]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in sorted(acc.items()):
    if len(next(iter(d.values()))) < 3: continue
    print(name)
    for c, v in sorted(d.items()):
        v.sort(); print("   %-24s median %.4g  (n=%d)" % (c, v[len(v)//2], len(v)))
PY
