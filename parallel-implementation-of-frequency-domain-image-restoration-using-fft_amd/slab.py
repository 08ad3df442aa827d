"""Single-image multi-GPU mode: ONE image split into row slabs over the ranks, the 2-D transform as
local row passes + all-to-all transposes -- the scheme of the reference's MPI variant (fft/fft_mpi.cpp:89-100
calculate_distribution, :170-279 distributed transpose, :284-307 rows -> transpose -> rows -> transpose, :316-470 the
operator), with RCCL's all-to-all over xGMI (torch.distributed, backend "nccl") in place of MPI_Alltoallv and the
per-rank steps as HIP kernels behind the fdr_slab_* entry points of libfdr.so.

Unlike the batched mode this one HAS a data-path collective: every transpose moves the whole array once (8 M N bytes
complex, 4 M N real), 1/G of it per rank.  Per channel: image and PSF forward (one exchange, both arrays together),
quotient back to rows, to columns again, the real plane back to rows = 4 exchanges (the reference does 6: it also
transposes both spectra back before the quotient, which is pointwise and needs no particular layout).

Arithmetic = the parity mode of the single-GPU path (rows first in both directions, the reference's operation order
for the quotient, min/max over the padded area by default), so the gathered result is bit-identical to
fdr.wienerDeblur_myfft(..., mode=MODE_PARITY): data movement is exact and min / max are order independent.

Unmeasured on multi-GPU hardware (no node in this environment); covered by a 2-rank rehearsal on one GPU
(backend gloo: tensors staged through the host) and by CPU tests of the exchange logic.
"""
import ctypes

import numpy as np

from . import MODE_PARITY, FLAG_TABLES_ONLY, NORM_PADDED, Plan, _check, lib, nextPowerOfTwo
from .batch import calculate_distribution


def _p(t):
    return ctypes.c_void_p(int(t.data_ptr()))


def alltoall_blocks(comm, send, send_counts, recv_counts):
    """send: flat tensor holding one block per destination rank (send_counts elements each, rank order); returns the
    flat tensor of the blocks received from every rank (recv_counts, rank order).  nccl: device tensors directly over
    RCCL; gloo (rehearsals / CPU tests): staged through host memory."""
    import torch
    if comm.dist is None:
        return send.clone()
    if comm.backend == "nccl":  # (Comm: the proven RCCL group; else its gloo group, staged through the host)
        recv = torch.empty(int(sum(recv_counts)), dtype=send.dtype, device=send.device)
        comm._guard("all_to_all_single", lambda: comm.dist.all_to_all_single(
            recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=comm.group))
        return recv
    h_send = send.detach().cpu().contiguous()
    h_recv = torch.empty(int(sum(recv_counts)), dtype=send.dtype)
    comm._guard("all_to_all_single", lambda: comm.dist.all_to_all_single(
        h_recv, h_send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=comm.group))
    return h_recv.to(send.device)


class SlabTransposer:
    """Distributed transpose of a global R x C array held as row slabs (fft/fft_mpi.cpp:170-279): rank g owns
    rows [rdis[g], rdis[g] + rcnt[g]) before and rows [cdis[g], ...) of the C x R transpose after."""

    def __init__(self, comm, R, C):
        self.comm, self.R, self.C = comm, int(R), int(C)
        self.rcnt, self.rdis = calculate_distribution(R, comm.world)
        self.ccnt, self.cdis = calculate_distribution(C, comm.world)
        self.lr, self.lc = self.rcnt[comm.rank], self.ccnt[comm.rank]

    def __call__(self, slab, elem_floats):
        """slab: device float32 tensor of lr x C elements (elem_floats = 2 complex, 1 real) -> lc x R elements."""
        import torch
        es = 4 * elem_floats
        st = _current_stream()  # the kernels, torch's allocations / copies and the collective all order on torch's current stream
        packed = torch.empty(self.lr * self.C * elem_floats, dtype=torch.float32, device=slab.device)
        counts = (ctypes.c_int * len(self.ccnt))(*self.ccnt)
        if self.lr > 0:
            _check(lib.fdr_slab_pack_dev(_p(slab), self.lr, self.C, len(self.ccnt), counts, es, _p(packed), st))
        torch.cuda.synchronize()
        send_counts = [self.lr * c * elem_floats for c in self.ccnt]
        recv_counts = [r * self.lc * elem_floats for r in self.rcnt]
        recv = alltoall_blocks(self.comm, packed, send_counts, recv_counts)
        # blocks arrive in rank order = global row order: recv is the R x lc matrix X[:, my columns], row-major
        out = torch.empty(self.lc * self.R * elem_floats, dtype=torch.float32, device=slab.device)
        if self.lc > 0:
            _check(lib.fdr_slab_transpose_dev(_p(recv), _p(out), self.R, self.lc, es, st))
        return out


def _current_stream():
    import torch
    return ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream))


def wiener_slab(comm, img_rows_local, rows, cols, psf, K, device=0, norm_area=NORM_PADDED, stream=None):
    """One channel of the operator on a row-slab decomposition.  img_rows_local: this rank's rows of the UNPADDED image
    (float32 [n, cols], rows [first, first + n) with the split of the PADDED row count M: ranks whose slab lies in the
    padding pass an empty array); returns this rank's rows of the restored, cropped image as a numpy array.
    Everything -- the fdr_slab_* kernels, torch's allocations and copies, the all-to-all -- is ordered on ONE stream:
    torch's current stream, or `stream` (a hipStream_t handle) made current for the duration of the call."""
    import torch
    if stream:
        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=torch.device("cuda", device))):
            return wiener_slab(comm, img_rows_local, rows, cols, psf, K, device=device, norm_area=norm_area, stream=None)
    M, N = nextPowerOfTwo(rows), nextPowerOfTwo(cols)
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    st = _current_stream()
    rcnt, rdis = calculate_distribution(M, comm.world)
    lr, first = rcnt[comm.rank], rdis[comm.rank]
    t_fwd = SlabTransposer(comm, M, N)   # rows -> columns
    t_bwd = SlabTransposer(comm, N, M)   # columns -> rows
    lc = t_fwd.lc
    with Plan(M, N, MODE_PARITY, device=device, flags=FLAG_TABLES_ONLY) as plan:
        h = plan._h
        # 1. pad: this rank's rows of the padded image and of the padded PSF (top-left anchored, fft_mpi.cpp:357-366)
        img_local = np.ascontiguousarray(img_rows_local, dtype=np.float32).reshape(-1, cols)
        valid = img_local.shape[0]
        assert valid == max(0, min(rows, first + lr) - first)
        psf = np.ascontiguousarray(psf, dtype=np.float32)
        p_lo, p_hi = min(first, psf.shape[0]), min(first + lr, psf.shape[0])
        psf_local = psf[p_lo:p_hi]
        d_img = torch.from_numpy(img_local.copy()).to(dev) if valid else torch.zeros(1, device=dev)
        d_psf = torch.from_numpy(psf_local.copy()).to(dev) if psf_local.size else torch.zeros(1, device=dev)
        G = torch.empty(max(lr, 1) * N * 2, dtype=torch.float32, device=dev)
        H = torch.empty(max(lr, 1) * N * 2, dtype=torch.float32, device=dev)
        _check(lib.fdr_slab_pad_dev(_p(d_img), valid, cols if valid else 0, cols, _p(G), lr, N, st))
        _check(lib.fdr_slab_pad_dev(_p(d_psf), psf_local.shape[0], psf.shape[1] if psf_local.size else 0, psf.shape[1], _p(H), lr, N, st))
        # 2. rows forward (length N), both arrays; 3. one exchange for both (stacked: 2 lr rows would break the block
        #    order, so they go one after the other); 4. rows forward along the former columns (length M)
        _check(lib.fdr_slab_rows_fft_dev(h, _p(G), lr, 0, 0, st))
        _check(lib.fdr_slab_rows_fft_dev(h, _p(H), lr, 0, 0, st))
        Gt, Ht = t_fwd(G, 2), t_fwd(H, 2)
        _check(lib.fdr_slab_rows_fft_dev(h, _p(Gt), lc, 1, 0, st))
        _check(lib.fdr_slab_rows_fft_dev(h, _p(Ht), lc, 1, 0, st))
        # 5. Wiener quotient, pointwise in the transposed layout (fft_serial.cpp:186-224 operation order)
        _check(lib.fdr_slab_wiener_dev(h, _p(Gt), _p(Ht), lc * M, ctypes.c_float(K), st))
        # 6. inverse, rows first as the serial path does (fft_serial.cpp:113-139): back to rows, rows inverse (N),
        #    to columns, rows inverse (M)
        Q = t_bwd(Gt, 2)
        _check(lib.fdr_slab_rows_fft_dev(h, _p(Q), lr, 0, 1, st))
        Rt = t_fwd(Q, 2)
        _check(lib.fdr_slab_rows_fft_dev(h, _p(Rt), lc, 1, 1, st))
        # 7. real part, back to rows (4 bytes per pixel on the wire)
        raw_t = torch.empty(max(lc, 1) * M, dtype=torch.float32, device=dev)
        _check(lib.fdr_slab_real_dev(_p(Rt), _p(raw_t), lc * M, st))
        raw = t_bwd(raw_t, 1)
        # 8. global min / max: local window, then MIN / MAX all-reduce (exact, order independent)
        mm_rows = M if norm_area == NORM_PADDED else rows
        mm_cols = N if norm_area == NORM_PADDED else cols
        d_mm = torch.tensor([float("inf"), float("-inf")], dtype=torch.float32, device=dev)
        local_mm_rows = max(0, min(mm_rows, first + lr) - first)
        if lr > 0 and local_mm_rows > 0:
            _check(lib.fdr_slab_minmax_dev(h, _p(raw), lr, N, local_mm_rows, mm_cols, _p(d_mm), st))
        torch.cuda.synchronize()
        mn, mx = comm.allreduce_min(float(d_mm[0].item())), comm.allreduce_max(float(d_mm[1].item()))
        d_mm = torch.tensor([mn, mx], dtype=torch.float32, device=dev)
        # 9. normalise + crop this rank's rows
        out = torch.empty(max(valid, 1) * cols, dtype=torch.float32, device=dev)
        if valid > 0:
            _check(lib.fdr_slab_normalize_dev(_p(raw), N, _p(d_mm), _p(out), valid, cols, cols, st))
        torch.cuda.synchronize()
        return out[:valid * cols].reshape(valid, cols).cpu().numpy()
