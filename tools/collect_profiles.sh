#!/bin/bash
# Runs on the GPU box (through gpurun), ONE call for the whole evidence set of a round, all from the build in the tree:
#   * python bench.py (defaults)                         -> bench_line.json  (the line the driver will see)
#   * rocprofv3 --kernel-trace --stats of bench.py --streams 1 at 4096^2 and 8192^2 (un-overlapped kernel durations;
#     the launch grouping is bench's default, so the averages are per launch as bench's roofline reports them)
#   * rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace options beside them) of
#     bench.py --streams 1 at both sizes (bench's own launch grouping: the counters are per LAUNCH of 4 / 2 images), and on a
#     known-size float4 copy (calibration of FETCH_SIZE on gfx950)
#   * the 2-rank rehearsal of the N > 1 bench path on this one GPU (backend gloo, --one-device), weak and strong scaling
#   usage: tools/collect_profiles.sh <tag>
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -s -C tools/microbench membench 2>/dev/null || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/microbench/membench tools/microbench/membench.hip
python3 bench.py > $OUT/bench_default.log 2>&1; echo "bench_default rc=$?" >> $OUT/status.txt
grep '^{' $OUT/bench_default.log | tail -n 1 > $OUT/bench_line.json
# the same with passes C' + E (raw real plane, 36 B/pixel) instead of the default two-sweep C1 + C2 (32 B/pixel)
python3 bench.py --raw-plane --no-cpu-baseline --no-psf-recompute > $OUT/bench_raw_plane.log 2>&1; echo "bench_raw_plane rc=$?" >> $OUT/status.txt
for S in 4096 8192; do
  if [ $S = 8192 ]; then B="--batch 12 --steps 6 --warmup 2"; else B="--batch 48 --steps 10 --warmup 3"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$S -o kt -- python3 bench.py --size $S $B --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute > $OUT/kt_$S.log 2>&1; echo "kt_$S rc=$?" >> $OUT/status.txt
  grep '^{' $OUT/kt_$S.log | tail -n 1 > $OUT/bench_line_streams1_$S.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$S -o fetch -- python3 bench.py --size $S --steps 2 --warmup 1 --batch 8 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute > $OUT/fetch_$S.log 2>&1; echo "fetch_$S rc=$?" >> $OUT/status.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$S -o write -- python3 bench.py --size $S --steps 2 --warmup 1 --batch 8 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute > $OUT/write_$S.log 2>&1; echo "write_$S rc=$?" >> $OUT/status.txt
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cal -- ./tools/microbench/membench > $OUT/cal_fetch.log 2>&1; echo "cal_fetch rc=$?" >> $OUT/status.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cal -- ./tools/microbench/membench > $OUT/cal_write.log 2>&1; echo "cal_write rc=$?" >> $OUT/status.txt
# N > 1 path of bench.py on the one GPU: two ranks, gloo, both on cuda:0 (torchrun is started before anything touches the GPU)
python3 bench.py --gpus 2 --backend gloo --one-device --size 2048 --batch 32 --steps 5 --warmup 2 --repeats 3 --no-psf-recompute > $OUT/two_rank_weak.log 2>&1; echo "two_rank_weak rc=$?" >> $OUT/status.txt
python3 bench.py --gpus 2 --backend gloo --one-device --size 2048 --total-batch 63 --steps 5 --warmup 2 --repeats 3 --no-psf-recompute > $OUT/two_rank_strong.log 2>&1; echo "two_rank_strong rc=$?" >> $OUT/status.txt
# BASELINE config 5 on the one GPU (512 x 2048^2, device resident) and config 2's size (1024^2)
python3 bench.py --size 2048 --total-batch 512 --steps 10 --warmup 2 --repeats 3 > $OUT/config5_one_gpu.log 2>&1; echo "config5_one_gpu rc=$?" >> $OUT/status.txt
python3 bench.py --size 1024 --batch 256 --steps 10 --warmup 2 --repeats 3 > $OUT/config2_size.log 2>&1; echo "config2_size rc=$?" >> $OUT/status.txt
python3 bench.py --size 8192 --batch 24 --steps 6 --warmup 2 --repeats 3 --cpu-size 8192 > $OUT/config4_size.log 2>&1; echo "config4_size rc=$?" >> $OUT/status.txt
cat $OUT/status.txt
cut -c1-400 $OUT/bench_line.json
