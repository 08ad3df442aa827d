#!/usr/bin/env python3
"""bench.py -- Mpixels/s restored (FFT + Wiener + IFFT) on synthetic N x N fp32 images, MI355X.

    python bench.py --gpus N --steps K --warmup W [--size 4096] [--batch 4] [--mode fast|parity]

A "step" is one pass of the hot path (pad -> FFT2 -> Wiener -> IFFT2 -> real -> normalise -> crop)
over one batch of `--batch` device-resident synthetic images per GPU.  Inputs and outputs stay in
HBM; the PSF spectrum is prepared once per plan, outside the timed region.  N > 1: one process per
GPU (torch.distributed / RCCL), independent images sharded by rank, no data-path collective, weak
scaling (per-GPU batch fixed).  Rank 0 prints ONE JSON line.

The JSON line also carries
  roofline     : algorithmic bytes of the dominant kernel / its mean duration, measured with hipEvent
                 pairs on the launch stream over a repetition of the same K steps (the timed region
                 itself is left un-instrumented), against the 8 TB/s HBM peak.  The formulation is the
                 HALF spectrum (32 B/pixel), and a pass-B' launch of n images counts the
                 filter W ONCE (8 n + 4 B/pixel): `frac` is a bandwidth fraction, `frac_by_counters`
                 the same from the PMC bytes of profiles/traffic.json;
  cpu_baseline : the CPU oracle (oracle/, a restatement of the reference's serial path) timed on one
                 host core on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# Algorithmic bytes per padded pixel of each pass.  "full": the complex-to-complex
# formulation of SURVEY.md 8d (56 B/px fused, 72 B/px in the reference's pass order).  "half": the default fast
# path keeps only the non-redundant half of the Hermitian spectrum, so every spectrum / filter
# access is 4 B per image pixel instead of 8 -- the bytes that formulation actually has to move.
PASS_BYTES = {
    "full": {
        "A rows: pad+FFT (real->complex)": 12, "B cols: FFT+Wiener": 24, "C rows: IFFT (complex)": 16,
        "D cols: IFFT+real+minmax": 12, "B' cols: FFT*W*IFFT": 24, "C' rows: IFFT+real+minmax": 12, "E normalize+crop": 8,
    },
    "half": {
        "A rows: pad+FFT (real->complex)": 8, "B' cols: FFT*W*IFFT": 12, "C' rows: IFFT+real+minmax": 8, "E normalize+crop": 8,
        # two-sweep normalisation (default): the inverse row pass runs twice, no raw real plane
        "C1 rows: IFFT+minmax": 4, "C2 rows: IFFT+normalize+crop": 8,
    },
}
PIPELINE_BYTES = {("fast", "half"): 36, ("fast", "full"): 56, ("parity", "full"): 72}
SEEDS = {1024: 0x5EED0002, 4096: 0x5EED0003, 8192: 0x5EED0004, 2048: 0x5EED0005}


def split_pass_name(name):
    """'B' cols: FFT*W*IFFT [4 images]' -> ("B' cols: FFT*W*IFFT", 4)"""
    if name.endswith(" images]"):
        base, tail = name.rsplit(" [", 1)
        return base, int(tail.split()[0])
    return name, 1


def pass_bytes_per_launch(name, spectrum, P):
    """Algorithmic HBM bytes of ONE launch of pass `name` (which may cover several images).  Pass B' reads the filter W once
    per launch, not once per image: the workgroups that handle one tile for the n images of a launch are neighbours on one
    XCD and share the tile's slice of W in its L2 (the PMC traffic confirms it), so a launch moves
    n x (spectrum in + out) + 1 x W."""
    base, nimg = split_pass_name(name)
    bpp = PASS_BYTES[spectrum].get(base, 0)
    if base.startswith("B' cols"):
        w = 4 if spectrum == "half" else 8
        return ((bpp - w) * nimg + w) * P, nimg
    return bpp * P * nimg, nimg


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096, help="synthetic image edge (power of two)")
    ap.add_argument("--batch", type=int, default=96, help="images per GPU per step")
    ap.add_argument("--mode", choices=["fast", "parity"], default="fast")
    ap.add_argument("--streams", type=int, default=0, help="internal streams / workspaces the batch alternates over (0 = 2 in fast mode, 3 in parity mode)")
    ap.add_argument("--group", type=int, default=0, help="images per launch of every pass (fast mode; 1..8, streams*group <= 16; 0 = 8 up to 1024^2, 4 up to 4096^2, else 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals on a one-GPU box)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--cpu-size", type=int, default=0, help="edge of the CPU-baseline sample (default: --size, capped at 4096)")
    ap.add_argument("--repeats", type=int, default=5, help="repetitions of the K timed steps; value = median (SURVEY 8d)")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="strong scaling: a FIXED batch of T images per step sharded over the ranks by calculate_distribution "
                         "(fft/fft_mpi.cpp:89-100), e.g. BASELINE config 5: --size 2048 --total-batch 512; 0 = weak scaling with --batch per GPU")
    ap.add_argument("--raw-plane", action="store_true", help="passes C' + E (raw real plane, 36 B/pixel) instead of the two-sweep C1 + C2 (32 B/pixel)")
    ap.add_argument("--no-psf-recompute", action="store_true", help="skip the second figure (PSF spectrum rebuilt per image)")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip config.value_parity_mode (a short timed leg in the bit-identical mode)")
    ap.add_argument("--no-batch-check", action="store_true",
                    help="skip the one-image-at-a-time recomputation of every image of a step (profiling runs: its single-image launches "
                         "of the same kernels would be averaged into rocprofv3's per-kernel statistics)")
    ap.add_argument("--bcast-filter", action="store_true",
                    help="rank 0 alone prepares the PSF spectrum / filter W and broadcasts it (RCCL under nccl) instead of every rank "
                         "recomputing it: the MPI_Bcast / Scatterv of the padded PSF in the reference's MPI variant (fft/fft_mpi.cpp:334-378)")
    return ap.parse_args()


def cpu_baseline(size):
    """Oracle (kind 'port'), one thread PINNED to one core (taskset -c 0): as many size x size
    single-channel images as fit ~12 s.  Returns (json object, restored image 0) -- image 0 also serves as the
    reference of the GPU-vs-CPU error figures of the bench line."""
    from oracle import oracle as o
    o.build()
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, {min(old_aff)})
    except (AttributeError, OSError):
        old_aff = None
    try:
        return _cpu_baseline_pinned(o, size, min(old_aff) if old_aff else None)
    finally:
        if old_aff:
            os.sched_setaffinity(0, old_aff)


def _cpu_baseline_pinned(o, size, core):
    psf = o.motion_blur_kernel(50, 30.0)
    P = size * size
    img = o.synth_image(SEEDS.get(size, 0x5EED0000), 0, P).reshape(size, size)
    t0 = time.perf_counter()
    ref0 = o.serial_channel(img, psf, 0.01)
    first = time.perf_counter() - t0
    extra = max(0, min(7, int(12.0 / first) - 1))
    more = [o.synth_image(SEEDS.get(size, 0x5EED0000), (b + 1) * P, P).reshape(size, size) for b in range(extra)]
    t0 = time.perf_counter()
    for im in more:
        o.serial_channel(im, psf, 0.01)
    dt = first + (time.perf_counter() - t0 if extra else 0.0)
    n = 1 + extra
    return {
        "value": round(n * P / 1e6 / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": "%d image(s) %dx%d fp32, PSF 50/30deg, K=0.01, oracle/fdr_oracle.c (gcc -O2, no FMA), %.1f s on 1 of %d host cores%s"
                  % (n, size, size, dt, os.cpu_count() or 0, (", pinned to core %d" % core) if core is not None else ""),
    }, ref0


def main():
    args = parse()
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL / hipIpc fail with the legacy mode): keep the setting in
    # whatever environment this process and the ranks it may launch run in
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run as a child (nothing has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)]
        cmd += sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    fdr = importlib.import_module(PKG)  # raises if libfdr.so is missing: no CPU fallback
    from importlib import import_module
    batch_mod = import_module(PKG + ".batch")

    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = batch_mod.Comm(backend=args.backend, device=dev)
    rank, world = comm.rank, comm.world

    S = args.size
    mode = fdr.MODE_FAST if args.mode == "fast" else fdr.MODE_PARITY
    P = S * S
    seed = SEEDS.get(S, 0x5EED0000)
    if args.total_batch > 0:   # strong scaling: a fixed batch sharded by the reference's own distribution rule
        counts, displs = batch_mod.calculate_distribution(args.total_batch, world)
        B, first_image, scaling = counts[rank], displs[rank], "strong"
    else:                      # weak scaling: every rank restores --batch images per step
        B, first_image, scaling = args.batch, rank * args.batch, "weak"

    flags = fdr.FLAG_FULL_SPECTRUM if os.environ.get("FDR_FULL_SPECTRUM") == "1" else 0
    spectrum = "half" if (args.mode == "fast" and not (flags & fdr.FLAG_FULL_SPECTRUM) and S >= 32) else "full"
    plan = fdr.Plan(S, S, mode, device=local_rank, flags=flags)
    if args.raw_plane:
        plan.set_option(fdr.OPT_TWO_SWEEP_NORM, 0)
    stream = torch.cuda.current_stream().cuda_stream
    # measured best (tools/microbench/passbench, profiles/README.md): every pass runs 10-17 % faster per image when a
    # launch covers more than one image (launch gaps and the ramp-up / drain of a grid amortise; in pass B' the
    # workgroups of one tile share its slice of the filter W in their XCD's L2), so images go in groups: up to 1024^2
    # 2 streams x 8 images per launch (512^2: 56 k -> 89 k, 1024^2: 130 k -> 165 k Mpixels/s against 2 x 4), up to 4096^2
    # 2 x 4, 8192^2 2 x 2 (24 x 8192^2: 3 x 2 145 k, 2 x 4 142 k, 2 x 2 150 k Mpixels/s)
    if args.streams <= 0:
        args.streams = 2 if args.mode == "fast" else 3
    if args.group <= 0:
        args.group = 8 if S <= 1024 else (4 if S <= 4096 else 2)
    plan.set_batching(args.streams, args.group if args.mode == "fast" else 1)
    bcast = None
    if args.bcast_filter:
        # ONE rank prepares the filter, the others receive its bytes: dist.broadcast of the plan's filter block (RCCL over
        # xGMI under "nccl"); every rank then hashes the filter it holds and the end-of-run all-reduce proves they are rank
        # 0's bits.  Default stays local recompute (cheaper than the broadcast).
        nbytes = plan.filter_bytes()
        wbuf = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        if rank == 0:
            plan.set_psf_motion(50, 30.0, 0.01, stream=stream)
            plan.export_filter_dev(wbuf.data_ptr(), nbytes, stream=stream)
        t0b = time.perf_counter()
        comm.broadcast(wbuf, src=0)
        torch.cuda.synchronize()
        bcast_s = time.perf_counter() - t0b
        if rank != 0:
            plan.import_filter_dev(wbuf.data_ptr(), nbytes, 0.01, stream=stream)
        chk_buf = torch.empty_like(wbuf)
        plan.export_filter_dev(chk_buf.data_ptr(), nbytes, stream=stream)  # what the plan really holds now
        torch.cuda.synchronize()
        whash = int(chk_buf.to(torch.int64).sum().item()) & ((1 << 52) - 1)
        bcast = {"bytes": nbytes, "seconds": bcast_s, "hash": whash}
        del wbuf, chk_buf
    else:
        plan.set_psf_motion(50, 30.0, 0.01, stream=stream)  # PSF generated, padded and transformed on the device
    imgs = torch.empty((max(B, 1), S, S), dtype=torch.float32, device=dev)
    outs = torch.zeros((max(B, 1), S, S), dtype=torch.float32, device=dev)
    # image b of this rank is global image first_image + b of the counter-based generator (distinct per rank)
    if B > 0:
        fdr.synth_image_dev(imgs.data_ptr(), B * P, seed, first_index=first_image * P, device=local_rank, stream=stream)
    torch.cuda.synchronize()

    # (arguments converted once: with one small image per step the call itself is ~20 us)
    step = plan.prepared_batch_dev(imgs.data_ptr(), P, B, S, S, S, outs.data_ptr(), P, S, fdr.NORM_PADDED, stream=stream)

    # one untimed priming step as part of the setup (code objects load and the internal streams / workspaces are touched
    # on first use); the W warm-up steps of the contract follow inside timed_steps.  The K timed steps are repeated
    # `--repeats` times (each repetition bracketed by barrier + synchronize, MAX over ranks); `value` is the median.
    step()
    torch.cuda.synchronize()
    reps = []
    for r in range(max(1, args.repeats)):
        reps.append(batch_mod.timed_steps(comm, step, torch.cuda.synchronize, args.steps, args.warmup if r == 0 else 0))
    elapsed = sorted(reps)[len(reps) // 2]

    # ---- consistency check over RCCL (outside the timed region) ----
    images_mine = B * args.steps  # images this rank restored in one repetition of the timed region
    if B > 0:
        chk = float(outs[:B].double().sum().item())
        finite = bool(torch.isfinite(outs[:B]).all().item())
        omin, omax = float(outs[:B].min().item()), float(outs[:B].max().item())
        # every image is min-max normalised over its whole (padded = full) plane: each must span [0, 1]
        spans = bool(((outs[:B].amin(dim=(1, 2)) <= 1e-6) & (outs[:B].amax(dim=(1, 2)) > 1.0 - 1e-6)).all().item())
        # the overlapped, grouped batch must give the bits of the one-image-at-a-time path (one stream, one image per
        # launch) for EVERY image of the step (round 2's LDS race lived in the images nobody compared): one extra pass
        # outside the timed region, compared on the device
        one = torch.empty((S, S), dtype=torch.float32, device=dev)
        differing = []
        for k in range(0 if args.no_batch_check else B):
            plan.wiener_dev(imgs[k].data_ptr(), S, S, S, one.data_ptr(), S, fdr.NORM_PADDED, stream=stream)
            if not bool(torch.equal(one, outs[k])):
                differing.append(k)
        del one
        same = not differing
        # (dst = src * scale + shift in two float roundings, as cv::normalize / the oracle: the extremes land within an ulp or two
        # of 0 and 1, on either side -- the range test allows 1e-6)
        ok = finite and omin >= -1e-6 and omax <= 1.0 + 1e-6 and spans and same
    else:
        chk, ok, differing = 0.0, True, []
        finite, omin, omax, spans = True, 0.0, 1.0, True
    tot = comm.allreduce_sum([images_mine, chk, 1.0 if ok else 0.0, B])
    if bcast is not None:  # every rank's filter hash must be rank 0's
        bcast["hash_min"], bcast["hash_max"] = int(comm.allreduce_min(bcast["hash"])), int(comm.allreduce_max(bcast["hash"]))
        bcast["seconds_max"] = comm.allreduce_max(bcast["seconds"])
    out0 = outs[0].cpu().numpy() if (rank == 0 and B > 0) else None

    # ---- second figure: the PSF spectrum rebuilt for every image, as the reference's loop does per channel
    #      (fft/fft_gpu.cu:329-343,356): PSF generation + pad + 2-D FFT + filter inside the step, one stream ----
    psf_elapsed = None
    if not args.no_psf_recompute:  # (every rank takes part in the barriers, also one whose shard is empty)
        nb = min(B, 16)

        def step_psf():
            for b in range(nb):
                plan.set_psf_motion(50, 30.0, 0.01, stream=stream)
                plan.wiener_dev(imgs[b].data_ptr(), S, S, S, outs[b].data_ptr(), S, fdr.NORM_PADDED, stream=stream)

        psf_elapsed = batch_mod.timed_steps(comm, step_psf, torch.cuda.synchronize, args.steps, 1)
        psf_images = comm.allreduce_sum([nb * args.steps])[0]

    # ---- third figure: the bit-identical mode (FDR_MODE_PARITY: reference pass order, 72 B/pixel, 5 launches) on a bounded leg ----
    parity_elapsed, parity_images = None, 0
    if not args.no_parity_leg and args.mode == "fast":
        nbp = min(B, 12 if S <= 4096 else 4)
        psteps = max(1, min(args.steps, 4))
        pplan = fdr.Plan(S, S, fdr.MODE_PARITY, device=local_rank)
        pplan.set_batching(3, 1)
        pplan.set_psf_motion(50, 30.0, 0.01, stream=stream)
        pouts = torch.empty((max(nbp, 1), S, S), dtype=torch.float32, device=dev)

        def step_parity():
            if nbp > 0:
                pplan.wiener_batch_dev(imgs.data_ptr(), P, nbp, S, S, S, pouts.data_ptr(), P, S, fdr.NORM_PADDED, stream=stream)

        parity_elapsed = batch_mod.timed_steps(comm, step_parity, torch.cuda.synchronize, psteps, 1)
        parity_images = comm.allreduce_sum([nbp * psteps])[0]
        pplan.close()
        del pouts

    # ---- per-kernel durations: hipEvent pairs on the launch stream, same K steps again (one stream: un-overlapped) ----
    plan.profile(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    passes = plan.pass_times()
    plan.profile(False)

    # ---- what a plain streaming kernel reaches on THIS device, for context next to the nominal peak: an out-of-place
    # elementwise pass (read 4 B + write 4 B per element) over the floats of one launch group, torch's own kernel ----
    stream_ref = None
    if rank == 0:
        try:
            n_el = P * max(1, min(args.group, 8))
            src = torch.empty(n_el, dtype=torch.float32, device=dev).fill_(1.0)
            dst = torch.empty_like(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                torch.mul(src, 1.0, out=dst)
            e0.record()
            for _ in range(20):
                torch.mul(src, 1.0, out=dst)
            e1.record()
            torch.cuda.synchronize()
            stream_ref = 8.0 * n_el * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            stream_ref = None

    rc = 0
    if rank == 0:
        images = int(tot[0])
        value = images * P / 1e6 / elapsed
        # bytes per pixel of the passes that actually ran, per image, with W counted once per pass-B' launch
        per_image_bytes = 0.0
        for n, _, _ in passes:
            bl, ni = pass_bytes_per_launch(n, spectrum, P)
            per_image_bytes += bl / ni
        pipe_bpp = per_image_bytes / P if per_image_bytes else float(PIPELINE_BYTES[(args.mode, spectrum)])
        pipe_gbps = pipe_bpp * P * images / elapsed / 1e9
        dom = max(passes, key=lambda t: t[1]) if passes else None  # longest launch
        roofline = None
        if dom:
            name, ms, cnt = dom
            base_name, nimg = split_pass_name(name)
            alg, _ = pass_bytes_per_launch(name, spectrum, P)
            achieved = alg / (ms * 1e-3) / 1e9
            per_image_equiv = PASS_BYTES[spectrum].get(base_name, 0) * P * nimg  # every image counted with its own W
            all_frac = {}
            for n, m, _ in passes:
                if m > 0:
                    all_frac[n] = round(pass_bytes_per_launch(n, spectrum, P)[0] / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            roofline = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "frac_by_counters": None,
                "kernel": name, "kernel_ms": round(ms, 5), "launches_timed": cnt,
                "algorithmic_bytes_per_launch": alg,
                "formulation": ("%s spectrum%s; W counted once per pass-B' launch of n images ((8 n + 4) B/px in the half spectrum)"
                                % (spectrum, ", 32 B/px per image (SURVEY 8f-4), not the 56 B/px complex-to-complex count of SURVEY 8d"
                                   if spectrum == "half" else "")),
                "per_image_equivalent": {"note": "the same launch with W counted once per IMAGE (12 B/px in pass B'): bytes a one-image launch "
                                                 "would move, NOT bandwidth", "bytes_per_launch": per_image_equiv,
                                         "GBps": round(per_image_equiv / (ms * 1e-3) / 1e9, 1)},
                "all_passes_ms": {n: round(m, 5) for n, m, _ in passes},
                "all_passes_frac": all_frac,
                "pipeline": {"bytes_per_pixel": round(pipe_bpp, 3), "achieved": round(pipe_gbps, 1),
                             "frac": round(pipe_gbps / HBM_PEAK_GBPS, 4),
                             "c2c_56_Bpx_would_need_GBps": round(PIPELINE_BYTES[(args.mode, "full")] * P * images / elapsed / 1e9, 1)},
            }
            if stream_ref:
                roofline["streaming_reference"] = {
                    "what": "out-of-place elementwise pass (torch.mul, read 4 B + write 4 B per element) over %d x %dx%d floats, timed in this "
                            "process after the timed region: what a plain streaming kernel reaches on this device; context, not the peak"
                            % (max(1, min(args.group, 8)), S, S),
                    "GBps": round(stream_ref, 1), "frac_of_peak": round(stream_ref / HBM_PEAK_GBPS, 4),
                    "kernel_over_reference": round(achieved / stream_ref, 4), "pipeline_over_reference": round(pipe_gbps / stream_ref, 4)}
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                try:
                    # PMC bytes per launch come from a committed collection: they are reported only when that collection
                    # was made on the kernel sources this run was built from (csrc fingerprint) -- else traffic_stale
                    te = fdr.traffic_entry(json.load(open(tfile)), "%s/%s/%d" % (args.mode, spectrum, S), base_name, nimg,
                                           fdr.csrc_fingerprint(), P, spectrum)
                    roofline["traffic_stale"] = te["stale"]
                    if te["note"]:
                        roofline["traffic_note"] = te["note"]
                    if te["traffic"] is not None:
                        roofline["traffic"] = te["traffic"]
                        roofline["traffic_kernel"] = te["kernel"]
                        roofline["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2 on gfx950)"
                        roofline["frac_by_counters"] = round(roofline["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                except Exception as e:  # noqa: BLE001 -- a malformed evidence file must not cost the bench line
                    roofline["traffic_note"] = "profiles/traffic.json unreadable: %s" % e
        vals = sorted(images * P / 1e6 / t for t in reps)
        config = {"workload": "%dx%d synthetic fp32, PSF len=50 angle=30, K=0.01, single channel, device-resident" % (S, S),
                  "images_per_step": int(tot[3]), "images_per_gpu_per_step": B if scaling == "weak" else None,
                  "total_batch": args.total_batch or None,
                  "mode": args.mode, "spectrum": spectrum, "streams": args.streams, "images_per_launch": args.group,
                  "parallelism": "images sharded over %d rank(s), no data-path collective" % world,
                  "collectives": comm.collectives(),  # "rccl", or "gloo (...)" when asked for or when the RCCL group could not be built
                  "normalize_area": "padded (serial semantics)",
                  "repeats": len(reps), "value_min": round(vals[0], 1), "value_max": round(vals[-1], 1),
                  "value_with_psf_recompute": round(psf_images * P / 1e6 / psf_elapsed, 1) if (psf_elapsed and psf_images) else None,
                  "value_parity_mode": round(parity_images * P / 1e6 / parity_elapsed, 1) if (parity_elapsed and parity_images) else None,
                  "filter": ("broadcast from rank 0: %d bytes in %.3f ms (max over ranks), hash on every rank = rank 0's: %s"
                             % (bcast["bytes"], bcast["seconds_max"] * 1e3, bcast["hash_min"] == bcast["hash_max"] == bcast["hash"]))
                            if bcast is not None else "recomputed on every rank"}
        line = {
            "metric": "Mpixels/sec restored (FFT+Wiener+IFFT) at %dx%d fp32" % (S, S),
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config,
            "roofline": roofline,
            "check": {"images_done": images, "images_expected": int(tot[3]) * args.steps, "checksum": tot[1], "ranks_ok": int(tot[2]),
                      "ranks": world,
                      "rank0": {"finite": finite, "out_min": omin, "out_max": omax, "every_image_spans_0_1": spans},
                      "batch_vs_one_by_one": ("skipped (--no-batch-check)" if args.no_batch_check else
                                              "all %d images of a step recomputed one at a time on one stream on every rank: bit-identical required%s"
                                              % (B, "" if not differing else "; rank 0 DIFFERS at images %s" % differing[:16]))},
        }
        if int(tot[2]) != world or images != int(tot[3]) * args.steps:
            rc = 3
        if bcast is not None and not (bcast["hash_min"] == bcast["hash_max"] == bcast["hash"]):
            rc = 5  # some rank holds a filter that is not rank 0's
        if world == 1 and not args.no_cpu_baseline:
            cs = args.cpu_size or min(S, 4096)
            line["cpu_baseline"], ref0 = cpu_baseline(cs)
            if cs == S and out0 is not None:
                # GPU image 0 against the CPU oracle's image 0: the stated tolerance is 1e-4
                import numpy as np
                d = (out0.astype(np.float64) - ref0.astype(np.float64))
                line["parity"] = {"against": "oracle image 0 (serial path restatement)", "max_abs": float(np.abs(d).max()),
                                  "rel_l2": float(np.linalg.norm(d) / np.linalg.norm(ref0.astype(np.float64))), "tolerance": 1e-4}
                if not (line["parity"]["max_abs"] <= 1e-4 and line["parity"]["rel_l2"] <= 1e-4):
                    rc = 4
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    plan.close()
    comm.close()
    if rc:
        sys.exit(rc)  # the consistency check or the parity check failed: the line above says which


if __name__ == "__main__":
    main()
