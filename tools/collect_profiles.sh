#!/bin/bash
# Runs on the GPU box (through gpurun), ONE call for the whole evidence set of a round, all from the build in the tree:
#   * python bench.py (defaults)                         -> bench_line.json  (the line the driver will see)
#   * rocprofv3 --kernel-trace --stats of bench.py --streams 1 at 4096^2 and 8192^2 (un-overlapped kernel durations;
#     the launch grouping is bench's default, so the averages are per launch as bench's roofline reports them; run with
#     --no-batch-check --no-parity-leg: the check's single-image launches of the SAME kernels would be averaged in)
#   * rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace options beside them) of
#     bench.py --streams 1 at both sizes (bench's own launch grouping: the counters are per LAUNCH of 4 / 2 images), and on a
#     known-size float4 copy (calibration of FETCH_SIZE on gfx950)
#   * the 2-rank rehearsal of the N > 1 bench path on this one GPU (backend gloo, --one-device), weak and strong scaling, and
#     the same with --bcast-filter (rank 0's filter W broadcast to the other rank instead of recomputed)
#   * BASELINE config 2 as written -- ONE 1024^2 image per step (bench.py --batch 1 --streams 1 --group 1), also 512^2 and
#     2048^2 -- with the rocprofv3 kernel trace of the 1024^2 run, and passbench's per-pass view of the same
#   * tools/microbench/rmw_bench: the passes' TRAFFIC alone as plain streaming kernels (what this device gives the bytes)
#   usage: tools/collect_profiles.sh <tag>
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
make -s -C tools/microbench membench
python3 -c "import importlib; print(importlib.import_module('parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd').csrc_fingerprint())" > $OUT/csrc_fingerprint.txt
export FDR_REQUIRE_REF_MAINS=1
python3 bench.py > $OUT/bench_default.log 2>&1; echo "bench_default rc=$?" >> $OUT/status.txt
grep '^{' $OUT/bench_default.log | tail -n 1 > $OUT/bench_line.json
# the same with passes C' + E (raw real plane, 36 B/pixel) instead of the default two-sweep C1 + C2 (32 B/pixel)
python3 bench.py --raw-plane --no-cpu-baseline --no-psf-recompute > $OUT/bench_raw_plane.log 2>&1; echo "bench_raw_plane rc=$?" >> $OUT/status.txt
for S in 4096 8192; do
  if [ $S = 8192 ]; then B="--batch 12 --steps 6 --warmup 2"; else B="--batch 48 --steps 10 --warmup 3"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$S -o kt -- python3 bench.py --size $S $B --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute --no-parity-leg --no-batch-check > $OUT/kt_$S.log 2>&1; echo "kt_$S rc=$?" >> $OUT/status.txt
  grep '^{' $OUT/kt_$S.log | tail -n 1 > $OUT/bench_line_streams1_$S.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$S -o fetch -- python3 bench.py --size $S --steps 2 --warmup 1 --batch 8 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute --no-parity-leg --no-batch-check > $OUT/fetch_$S.log 2>&1; echo "fetch_$S rc=$?" >> $OUT/status.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$S -o write -- python3 bench.py --size $S --steps 2 --warmup 1 --batch 8 --streams 1 --repeats 1 --no-cpu-baseline --no-psf-recompute --no-parity-leg --no-batch-check > $OUT/write_$S.log 2>&1; echo "write_$S rc=$?" >> $OUT/status.txt
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cal -- ./tools/microbench/membench > $OUT/cal_fetch.log 2>&1; echo "cal_fetch rc=$?" >> $OUT/status.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cal -- ./tools/microbench/membench > $OUT/cal_write.log 2>&1; echo "cal_write rc=$?" >> $OUT/status.txt
# N > 1 path of bench.py on the one GPU: two ranks, gloo, both on cuda:0 (torchrun is started before anything touches the GPU)
python3 bench.py --gpus 2 --backend gloo --one-device --size 2048 --batch 32 --steps 5 --warmup 2 --repeats 3 --no-psf-recompute > $OUT/two_rank_weak.log 2>&1; echo "two_rank_weak rc=$?" >> $OUT/status.txt
python3 bench.py --gpus 2 --backend gloo --one-device --size 2048 --total-batch 63 --steps 5 --warmup 2 --repeats 3 --no-psf-recompute > $OUT/two_rank_strong.log 2>&1; echo "two_rank_strong rc=$?" >> $OUT/status.txt
python3 bench.py --gpus 2 --backend gloo --one-device --size 2048 --batch 16 --steps 5 --warmup 2 --repeats 3 --no-psf-recompute --no-parity-leg --bcast-filter > $OUT/two_rank_bcast_filter.log 2>&1; echo "two_rank_bcast_filter rc=$?" >> $OUT/status.txt
# config 5 exactly as the driver will call it, at the largest rank count this box admits (its process guard allows six GPU
# processes and the launcher counts as one): five ranks on the one device, every image of every rank checked
python3 bench.py --gpus 5 --backend gloo --one-device --size 2048 --total-batch 512 --steps 3 --warmup 1 --repeats 2 --no-psf-recompute --no-parity-leg > $OUT/five_rank_config5.log 2>&1; echo "five_rank_config5 rc=$?" >> $OUT/status.txt
# BASELINE config 2 as written: ONE image per step, one stream, one image per launch (and the neighbouring sizes)
for S in 512 1024 2048; do
  python3 bench.py --size $S --batch 1 --streams 1 --group 1 --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-psf-recompute --no-parity-leg > $OUT/single_image_$S.log 2>&1; echo "single_image_$S rc=$?" >> $OUT/status.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_single_1024 -o kt -- python3 bench.py --size 1024 --batch 1 --streams 1 --group 1 --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-psf-recompute --no-parity-leg > $OUT/kt_single_1024.log 2>&1; echo "kt_single_1024 rc=$?" >> $OUT/status.txt
make -s -C tools/microbench passbench seam_bench
PKG="$GRAFT_REPO_ROOT/parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"
for S in 256 512 1024 2048; do tools/microbench/passbench $PKG/libfdr.so $S 8 20 1 1 >> $OUT/single_image_passbench.log 2>&1; done
tools/microbench/passbench $PKG/libfdr.so 4096 24 10 2 4 >> $OUT/passbench_4096.log 2>&1
tools/microbench/passbench $PKG/libfdr.so 8192 6 6 2 2 >> $OUT/passbench_8192.log 2>&1
# pass B''s phase timeline (in-kernel stamps; needs the -DFDR_DEBUG_STAMPS build made beside the product build) and the bit-identical mode per pass
if [ -f $PKG/build_dbg/libfdr_stamps.so ]; then
  (tools/microbench/passbench $PKG/build_dbg/libfdr_stamps.so 4096 8 5 1 4; tools/microbench/passbench $PKG/build_dbg/libfdr_stamps.so 8192 4 3 1 2) > $OUT/passB_phase_stamps.log 2>&1
fi
(tools/microbench/passbench $PKG/libfdr.so 4096 8 5 3 1 0; tools/microbench/passbench $PKG/libfdr.so 8192 4 3 3 1 0) > $OUT/passbench_parity_mode.log 2>&1
make -s -C tools/microbench rmw_bench
(timeout -k 5 60 tools/microbench/rmw_bench 4 64 20; timeout -k 5 60 tools/microbench/rmw_bench 2 256 10) > $OUT/rmw_bench.log 2>&1; echo "rmw_bench rc=$?" >> $OUT/status.txt
(timeout -k 5 60 tools/microbench/seam_bench 256 256 200; timeout -k 5 60 tools/microbench/seam_bench 128 256 200; timeout -k 5 60 tools/microbench/seam_bench 256 512 200) > $OUT/seam_bench.log 2>&1; echo "seam_bench rc=$?" >> $OUT/status.txt
# BASELINE config 5 on the one GPU (512 x 2048^2, device resident) and config 2's size (1024^2)
python3 bench.py --size 2048 --total-batch 512 --steps 10 --warmup 2 --repeats 3 > $OUT/config5_one_gpu.log 2>&1; echo "config5_one_gpu rc=$?" >> $OUT/status.txt
python3 bench.py --size 1024 --batch 256 --steps 10 --warmup 2 --repeats 3 > $OUT/config2_size.log 2>&1; echo "config2_size rc=$?" >> $OUT/status.txt
python3 bench.py --size 8192 --batch 24 --steps 6 --warmup 2 --repeats 3 --cpu-size 8192 > $OUT/config4_size.log 2>&1; echo "config4_size rc=$?" >> $OUT/status.txt
# the bench line once more, now WITH the PMC bytes of this very collection: summarize_profiles.py writes profiles/traffic.json (entries carry the
# fingerprint of csrc/ recorded above) into this copy of the tree, and bench.py reports roofline.traffic / frac_by_counters only for a matching tree
python3 tools/summarize_profiles.py $TAG > $OUT/summarize_on_box.log 2>&1; echo "summarize_on_box rc=$?" >> $OUT/status.txt
python3 bench.py > $OUT/bench_with_traffic.log 2>&1; echo "bench_with_traffic rc=$?" >> $OUT/status.txt
grep '^{' $OUT/bench_with_traffic.log | tail -n 1 > $OUT/bench_line.json
cat $OUT/status.txt
cut -c1-400 $OUT/bench_line.json
