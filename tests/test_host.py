"""CPU tests of the host side: the C ABI library loads and exports every symbol include/fdr.h
declares, argument validation works without a GPU, and the batched-mode sharding / timing logic
runs under world_size 2 (gloo).  No compute call is made here."""
import ctypes
import importlib
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import PKG, ROOT


def test_library_exports_every_declared_symbol(fdr):
    header = open(os.path.join(ROOT, "include", "fdr.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fdr_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 24
    lib = ctypes.CDLL(fdr.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(fdr.EXPORTED_SYMBOLS) == declared
    assert lib.fdr_version() == 300
    # ... and nothing else: every dynamic symbol of the library that looks like an entry point is declared in the header
    # (a debug hook exported by accident would show up here)
    nm = subprocess.run(["nm", "-D", "--defined-only", fdr.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in nm.splitlines() if l.split() and l.split()[-1].startswith("fdr_")})
    assert exported == declared, sorted(set(exported) ^ set(declared))


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "gpu.cpp")), reason="the reference checkout is not on this machine")
def test_reference_drivers_compile_unchanged_against_include(tmp_path):
    """The reference's own serial.cpp and gpu.cpp, read where they lie (nothing is copied), compile against include/
    without OpenCV: include/opencv2/opencv.hpp -> fdr_cv.hpp supplies cv::imread / cvtColor / norm / imshow / waitKey /
    copyMakeBorder / getRotationMatrix2D / warpAffine / Size == on the bundled Mat.  Two ways, both must be clean:
      direct : g++ -I include <reference>/x.cpp    -- "utils.hpp" / "fft/fft.hpp" are then the REFERENCE's own headers
               (the binding flavour of INTEGRATION.md section 1; tools/cli/fft_hip.cpp supplies the definitions)
      stdin  : g++ -I include -x c++ - < x.cpp     -- the same names resolve to include/utils.hpp and include/fft/fft.hpp
               (the header-swap flavour)
    gpu.cpp's two CUDA lines (#include <cuda_runtime.h>, cudaFree(0): gpu.cpp:7,94) get a one-line stub made here."""
    stub = tmp_path / "cuda_stub"
    stub.mkdir()
    (stub / "cuda_runtime.h").write_text("inline int cudaFree(void*) { return 0; }\n")
    inc = os.path.join(ROOT, "include")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", inc, "-I", str(stub)]
    for name in ("serial.cpp", "gpu.cpp"):
        src = os.path.join(REFERENCE, name)
        r = subprocess.run(base + [src], capture_output=True, text=True)
        assert r.returncode == 0 and "error" not in r.stderr, (name, "direct", r.stderr[-3000:])
        with open(src) as f:
            r = subprocess.run(base + ["-x", "c++", "-"], stdin=f, capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0 and "error" not in r.stderr, (name, "stdin", r.stderr[-3000:])
    # the binding file itself against the reference's headers
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", REFERENCE, "-I", inc,
                        os.path.join(ROOT, "tools", "cli", "fft_hip.cpp")], capture_output=True, text=True)
    assert r.returncode == 0 and "error" not in r.stderr, r.stderr[-3000:]
    # no reference source text lives in this repository: the recipe reads it from /root/reference (oracle/Makefile)
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "$(REF)/serial.cpp" in mk and "$(REF)/gpu.cpp" in mk


def test_opencv_stand_in_host_functions(fdr, tmp_path):
    """tools/cli/cv_shim_test.cpp: the cv:: free functions of include/fdr_cv.hpp that never touch the device -- PNG / PPM round
    trips through imwrite / imread, imshow's FDR_IMSHOW_DIR behaviour, convertTo, BGR <-> Lab, norm, copyMakeBorder,
    getRotationMatrix2D, Size comparison -- checked on the host (cv::warpAffine is a device call: GPU tests)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools", "cli"), "-s", "cv_shim_test"])
    r = subprocess.run([os.path.join(ROOT, "tools", "cli", "cv_shim_test"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "cv shim ok" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-800:])


def test_integer_helpers_match_reference_semantics(fdr):
    # utils.hpp:27-37 / :50-52
    assert [fdr.nextPowerOfTwo(n) for n in (0, 1, 2, 3, 782, 1920, 4097)] == [1, 1, 2, 4, 1024, 2048, 8192]
    assert [fdr.isPowerOfTwo(n) for n in (0, 1, 2, 3, 1024, -4)] == [False, True, True, False, True, False]


def test_argument_validation_precedes_any_device_work(fdr):
    h = ctypes.c_void_p()
    assert fdr.lib.fdr_plan_create(0, 100, 64, 0, 0, ctypes.byref(h)) == -2  # FDR_ERR_NOT_POW2
    assert b"powers of two" in fdr.lib.fdr_last_error()
    assert fdr.lib.fdr_plan_create(0, 0, 64, 0, 0, ctypes.byref(h)) == -1
    assert fdr.lib.fdr_plan_create(0, 64, 65536, 0, 0, ctypes.byref(h)) == -1  # above 32768 (8192 < n <= 32768: long row pass)
    assert b"above 32768" in fdr.lib.fdr_last_error()
    assert fdr.lib.fdr_plan_create(0, 64, 64, 7, 0, ctypes.byref(h)) == -1     # unknown mode
    assert fdr.lib.fdr_plan_create(0, 64, 64, 0, 0, None) == -1
    assert fdr.lib.fdr_plan_destroy(None) == 0
    assert fdr.lib.fdr_wiener_f32_dev(None, None, 1, 1, 1, None, 1, 1, None) == -1
    assert fdr.lib.fdr_psf_motion(0, 30.0, None) == -1


def test_new_entry_points_validate_arguments_before_device_work(fdr):
    """fdr_batch_run / fdr_slab_* / options / phase times: null and out-of-range arguments are rejected before any HIP
    call (so this runs without a GPU); fdr_optimal_dft_size is pure host arithmetic (cv::getOptimalDFTSize)."""
    L = fdr.lib
    assert [fdr.getOptimalDFTSize(n) for n in (1, 2, 7, 11, 17, 782, 1920, 4097)] == [1, 2, 8, 12, 18, 800, 1920, 4320]
    assert L.fdr_batch_run(None, None) == -1
    d = fdr.BatchDesc()
    assert L.fdr_batch_run(ctypes.byref(d), None) == -1 and b"device entries" in L.fdr_last_error()
    devs = (ctypes.c_int * 1)(0)
    d.n_devices, d.devices, d.M, d.N, d.rows, d.cols, d.count = 1, devs, 64, 64, 65, 64, 1
    assert L.fdr_batch_run(ctypes.byref(d), None) == -1 and b"batch shape" in L.fdr_last_error()
    d.rows, d.steps = 64, 0
    assert L.fdr_batch_run(ctypes.byref(d), None) == -1 and b"steps" in L.fdr_last_error()
    d.steps, d.psf_size = 1, 0
    assert L.fdr_batch_run(ctypes.byref(d), None) == -1 and b"PSF" in L.fdr_last_error()
    assert L.fdr_plan_set_option(None, 1, 5) == -1
    assert L.fdr_plan_phase_times(None, None, 0) == -1
    assert L.fdr_slab_rows_fft_dev(None, None, 1, 0, 0, None) == -1
    assert L.fdr_slab_pack_dev(None, 1, 4, 1, None, 8, None, None) == -1
    assert L.fdr_slab_transpose_dev(None, None, 1, 1, 8, None) == -1
    assert L.fdr_slab_pad_dev(None, 0, 0, 0, None, 1, 8, None) == -1
    assert L.fdr_slab_wiener_dev(None, None, None, 0, ctypes.c_float(0.01), None) == -1
    assert L.fdr_slab_minmax_dev(None, None, 1, 1, 1, 1, None, None) == -1
    assert L.fdr_slab_normalize_dev(None, 1, None, None, 1, 1, 1, None) == -1
    assert L.fdr_slab_real_dev(None, None, 0, None) == -1
    assert L.fdr_warp_affine_f32(None, 1, 1, 1, None, None, 1, 1, 1) == -1
    n = ctypes.c_size_t(0)
    assert L.fdr_plan_filter_bytes(None, ctypes.byref(n)) == -1 and L.fdr_plan_filter_bytes(None, None) == -1
    assert L.fdr_plan_export_filter_dev(None, None, 0, None) == -1 and b"null argument" in L.fdr_last_error()
    assert L.fdr_plan_import_filter_dev(None, None, 0, ctypes.c_float(0.01), None) == -1
    h = ctypes.c_void_p()
    assert L.fdr_plan_create(0, 100, 64, 0, 0, ctypes.byref(h)) == -2                       # not a power of two ...
    assert L.fdr_plan_create(0, 5000, 64, 0, fdr.FLAG_ANY_SIZE, ctypes.byref(h)) == -1     # ... and too long for the naive-DFT table


def test_no_cpu_fallback_in_product_package():
    """The product path must not reach into oracle/ (or any numpy FFT) -- checked textually."""
    pkg_dir = os.path.join(ROOT, PKG)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "fdr_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
                assert "np.fft" not in text and "numpy.fft" not in text and "torch.fft" not in text, f


def test_calculate_distribution():
    batch = importlib.import_module(PKG + ".batch")
    # fft/fft_mpi.cpp:89-100 applied to images
    assert batch.calculate_distribution(512, 8) == ([64] * 8, [64 * i for i in range(8)])
    assert batch.calculate_distribution(10, 4) == ([3, 3, 2, 2], [0, 3, 6, 8])
    assert batch.calculate_distribution(3, 4) == ([1, 1, 1, 0], [0, 1, 2, 3])
    assert batch.calculate_distribution(0, 2) == ([0, 0], [0, 0])


def test_single_process_comm():
    batch = importlib.import_module(PKG + ".batch")
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        c = batch.Comm()
        assert (c.rank, c.world) == (0, 1)
        assert c.allreduce_max(1.5) == 1.5 and c.allreduce_sum([1, 2]) == [1.0, 2.0]
        calls = []
        dt = batch.timed_steps(c, lambda: calls.append(1), lambda: None, steps=5, warmup=2)
        assert len(calls) == 7 and dt >= 0.0
    finally:
        os.environ.update(env)


@pytest.mark.timeout(180)
def test_two_rank_gloo():
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), "11"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["images"] == 11 and r["index_sum"] == sum(range(11))
    assert r["bcast_ok"] is True
    assert r["elapsed"] >= 3 * 0.004  # MAX over ranks: the slower rank (2 * 2 ms per step) bounds it


@pytest.mark.timeout(180)
def test_two_rank_gloo_alltoall_transpose_logic():
    """The exchange of the single-image multi-GPU mode (..._amd/slab.py: calculate_distribution + all_to_all_single with
    the block sizes of fft/fft_mpi.cpp:118-147) on CPU tensors under gloo, world_size 2 and 3: blocks packed and
    transposed with numpy here (the device kernels that do it in the product are covered by the GPU tests)."""
    for world in (2, 3):
        port = 29900 + (os.getpid() % 90) + world
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "_a2a_worker.py"), "10", "7"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert r["world"] == world and r["ok"] is True, r


def test_traffic_json_is_tied_to_the_kernel_sources(fdr, tmp_path):
    """profiles/traffic.json entries carry the fingerprint of the csrc/ tree the counters were collected on; bench.py's
    roofline (fdr.traffic_entry) reports the PMC bytes only for a matching tree and flags `traffic_stale` otherwise."""
    fp = fdr.csrc_fingerprint()
    assert re.fullmatch(r"[0-9a-f]{16}", fp)
    P = 4096 * 4096
    name = "B' cols: FFT*W*IFFT"
    good = {"fast/half/4096": {name: {"per_launch": 36.0 * P, "images": 4, "kernel": "fdr::fft_cols_panel_fused16_kernel<12>",
                                      "grid_threads": 524288, "workgroup_threads": 256, "csrc": fp, "collected": "rXX"}}}
    te = fdr.traffic_entry(good, "fast/half/4096", name, 4, fp, P, "half")
    assert te["stale"] is False and te["traffic"] == 36.0 * P and te["kernel"].endswith("fused16_kernel<12>")
    # another launch size: the per-image part scales, W stays once
    te = fdr.traffic_entry(good, "fast/half/4096", name, 2, fp, P, "half")
    assert te["stale"] is False and te["traffic"] == 20.0 * P and "scaled" in te["note"]
    # the same file, edited: counters from other sources -> flagged, no bytes reported
    path = tmp_path / "traffic.json"
    edited = json.loads(json.dumps(good))
    edited["fast/half/4096"][name]["csrc"] = "0" * 16
    path.write_text(json.dumps(edited))
    te = fdr.traffic_entry(json.load(open(path)), "fast/half/4096", name, 4, fp, P, "half")
    assert te["stale"] is True and te["traffic"] is None and "0000" in te["note"]
    # an entry without a fingerprint (collections before round 4) is stale too; a missing entry is just absent
    legacy = {"fast/half/4096": {name: {"per_launch": 36.0 * P, "images": 4}}}
    assert fdr.traffic_entry(legacy, "fast/half/4096", name, 4, fp, P, "half")["stale"] is True
    te = fdr.traffic_entry(good, "fast/half/8192", name, 2, fp, P, "half")
    assert te["stale"] is False and te["traffic"] is None
    # touching a kernel source changes the fingerprint (checked on a copy of the function's own recipe)
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, PKG, "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".hpp", ".h")):
            h.update(n.encode() + b"\0")
            h.update(open(os.path.join(d, n), "rb").read() + (b" " if n == "fdr_panel.hip" else b""))
    assert h.hexdigest()[:16] != fp


def test_plan_cache_of_the_cpp_surface_is_bounded_and_released():
    """include/fft/fft.hpp: the per-thread plan cache is an object with a destructor, has a capacity knob and a release
    call (it used to be a leaked pointer).  Source-level check here; the GPU test runs the code."""
    src = open(os.path.join(ROOT, "include", "fft", "fft.hpp")).read()
    assert "static thread_local PlanCache c;" in src and "new PlanCache" not in src
    assert "~PlanCache() { clear(); }" in src
    for name in ("release_cached_plans", "set_plan_cache_capacity", "plan_cache_capacity"):
        assert re.search(r"inline \w+ %s\(" % name, src), name


def test_collective_failure_is_loud_not_a_hang(tmp_path):
    """batch.Comm: a collective that cannot complete (the peer rank is gone) ends the process with a one-line message and
    exit code 13 inside the deadline, so that a launcher stops the job instead of waiting on it (first contact with an
    8-GPU node must not be able to hang: VERDICT r03 next #2)."""
    import socket
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", FDR_DIST_TIMEOUT_S="20")
    worker = os.path.join(ROOT, "tests", "_dist_fail_worker.py")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    dt = time.time() - t0
    assert procs[1].returncode == 0
    assert procs[0].returncode == 13, (procs[0].returncode, outs[0])
    assert "fdr.batch: rank 0/2: barrier failed (backend gloo)" in outs[0][1], outs[0][1]
    assert "UNREACHABLE" not in outs[0][0]
    assert dt < 120, dt
