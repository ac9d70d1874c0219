"""One transform too large for (or simply spread over) several GPUs: the four-step scheme with the transposition done by
an all-to-all between the ranks (the reference's CPU statement of the scheme is
optimizations/parallel_fft.c:213-272 -- column transforms, twiddle W_N^(i j), row transforms, transpose).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the device, "gloo" in the CPU tests).  This is
the ONLY place of the library with a data-path collective: batched transforms shard by batch index and exchange nothing
(bench.py).  The local transforms are batched 1D plans of the engine behind the C ABI (fftlib.Plan on device pointers);
PyTorch carries the buffers, the local transposes and the all-to-all -- plumbing, not arithmetic of the transform
(the inter-step twiddle is one element-wise complex multiply by a table computed in fp64).

Layout.  N = N1 * N2, G ranks, G | N1 and G | N2.  Rank r owns the contiguous slab x[r N/G : (r+1) N/G] = rows
n1 in [r N1/G, (r+1) N1/G) of the N1 x N2 matrix A[n1][n2] = x[n1 N2 + n2].
  1. all-to-all: every rank gets N2/G whole columns            -> local [N2/G][N1]
  2. batched FFT of length N1 over n1 (batch N2/G)             -> B[n2][k1]
  3. twiddle B[n2][k1] *= W_N^(k1 n2)
  4. all-to-all: every rank gets N1/G values of k1, all n2     -> local [N1/G][N2]
  5. batched FFT of length N2 over n2 (batch N1/G)             -> C[k1][k2] = X[k1 + N1 k2]
  6. (natural_order=True) all-to-all: rank r ends up with the contiguous slab X[r N/G : (r+1) N/G]
The inverse runs the same steps with conjugate twiddles and inverse local transforms (each scaled by 1/length: 1/N overall).

ONE all-to-all (round 3; SURVEY.md 8f item 4: "one all-to-all between the passes"): steps 1 and 6 only re-distribute the
contiguous slabs a caller usually holds.  A caller that keeps the signal distributed BY COLUMNS (transposed_in=True: rank r holds
A[:, r N2/G : (r+1) N2/G] as a contiguous [N2/G][N1] array -- FFTW-MPI's TRANSPOSED_IN) and accepts C[k1][k2] = X[k1 + N1 k2]
distributed by k1 (natural_order=False: TRANSPOSED_OUT) runs steps 2-5: two local passes and the ONE exchange between them.  A
forward transform followed by an inverse one of the other orientation (convolutions, spectral filters) never needs the other two.
"""
import math

import numpy as np
import torch
import torch.distributed as dist


def _split(n, g):
    """N1 * N2 = n with g | N1, g | N2 and N1 as close to sqrt(n) as that allows (powers of two: always possible when g^2 | n)."""
    if n % (g * g) != 0:
        raise ValueError("a transform of length %d cannot be spread over %d ranks: %d^2 must divide it" % (n, g, g))
    best = None
    root = math.isqrt(n)
    for n1 in range(g, n // g + 1, g):
        if n % n1 == 0 and (n // n1) % g == 0:
            if best is None or abs(n1 - root) < abs(best - root):
                best = n1
    return best, n // best


class DistributedFFT1D:
    """local_fft(t, direction): batched 1D transform of a contiguous [batch, length] complex tensor on this rank's device,
    returning a tensor of the same shape.  On the GPU: `engine_local_fft` below (fftlib plans, cached per shape)."""

    def __init__(self, n, local_fft, group=None, natural_order=True, transposed_in=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = n
        self.n1, self.n2 = _split(n, self.world)
        self.local_fft = local_fft
        self.natural_order = natural_order
        self.transposed_in = transposed_in
        self.all_to_alls = 0  # exchanges of the last call (tests: 1 with transposed_in and natural_order=False, else 2 or 3)
        self._tw = {}

    def _all_to_all(self, t):
        """t: [G, chunk...] -> received [G, chunk...]: slice g goes to rank g (dist.all_to_all_single)."""
        self.all_to_alls += 1
        if self.world == 1:
            return t
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t.contiguous(), group=self.group)
        return out

    def _twiddle(self, direction, device, dtype):
        key = (direction, str(device), dtype)
        if key not in self._tw:
            g, n1, n2 = self.world, self.n1, self.n2
            cols = np.arange(self.rank * n2 // g, (self.rank + 1) * n2 // g, dtype=np.int64)  # this rank's n2 after step 1
            k1 = np.arange(n1, dtype=np.int64)
            m = (cols[:, None] * k1[None, :]) % self.n                                      # exact phase index, then fp64 trig
            ang = (2.0 * np.pi * direction / self.n) * m.astype(np.float64)
            w = (np.cos(ang) + 1j * np.sin(ang)).astype(np.complex64 if dtype == torch.complex64 else np.complex128)
            self._tw[key] = torch.from_numpy(w).to(device)
        return self._tw[key]

    def __call__(self, x_local, direction=-1):
        """x_local: this rank's slab, a contiguous complex tensor of N / G elements.  Returns this rank's slab of the
        spectrum (natural_order) or its [N1/G][N2] block of C[k1][k2] = X[k1 + N1 k2]."""
        g, n1, n2 = self.world, self.n1, self.n2
        self.all_to_alls = 0
        if self.transposed_in:
            cols = x_local.reshape(n2 // g, n1)                      # the caller holds whole columns: [my n2][n1]
        else:
            a = x_local.reshape(n1 // g, g, n2 // g)                 # [my n1][dest rank][its n2]
            send = a.permute(1, 0, 2).contiguous()                   # [dest][my n1][its n2]
            recv = self._all_to_all(send)                            # [src][src's n1][my n2]
            cols = recv.reshape(n1, n2 // g).t().contiguous()        # [my n2][n1]: whole columns, contiguous
        b = self.local_fft(cols.contiguous(), direction)             # [my n2][k1]
        b = b * self._twiddle(direction, b.device, b.dtype)
        send = b.reshape(n2 // g, g, n1 // g).permute(1, 0, 2).contiguous()   # [dest][my n2][its k1]
        recv = self._all_to_all(send)                                # [src][src's n2][my k1]
        rows = recv.reshape(n2, n1 // g).t().contiguous()            # [my k1][n2]
        c = self.local_fft(rows, direction)                          # [my k1][k2]  = X[k1 + N1 k2]
        if not self.natural_order:
            return c
        # X index = k1 + N1 k2: rank d owns k2 in [d N2/G, (d+1) N2/G); within its slab the order is k2-major, k1 fastest
        send = c.reshape(n1 // g, g, n2 // g).permute(1, 2, 0).contiguous()   # [dest][its k2][my k1]
        recv = self._all_to_all(send)                                # [src][my k2][src's k1]
        return recv.permute(1, 0, 2).reshape(-1).contiguous()        # [my k2][k1]


_plans = {}


def engine_local_fft(t, direction):
    """Batched transform of a contiguous CUDA tensor [batch, n] through the C ABI (plans cached per shape / direction / stream)."""
    import fftlib
    assert t.is_cuda and t.is_contiguous()
    batch, n = t.shape
    dt = np.complex64 if t.dtype == torch.complex64 else np.complex128
    # torch's CURRENT stream of that device.  Handle 0 -- torch's default stream -- would mean "the plan's own stream" to the C ABI:
    # name the legacy default stream explicitly (hipStreamLegacy = 1), or the transform would race with torch's kernels around it
    stream = torch.cuda.current_stream(t.device).cuda_stream or fftlib.HIP_STREAM_LEGACY
    # one plan per STREAM as well: a plan owns one set of scratch buffers, exchange windows and control words, so two torch streams
    # running the same shape at once must not share it
    key = (n, batch, direction, dt, t.device.index, stream)
    with torch.cuda.device(t.device):  # the plan lives on the TENSOR's device, not on whatever device is current
        prev = fftlib.get_device()  # the library's current device is global state of its own: put it back
        fftlib.set_device(t.device.index)
        try:
            if key not in _plans:
                plan = fftlib.Plan(n, batch, direction, dt)
                plan.set_stream(stream)
                # torch's caching allocator may hand `t` / `out` to somebody else as soon as the stream-ordered work is queued: the
                # library must never repeat an execute from these pointers after a team-kernel timeout (include/fft_hip.h)
                plan.set_option(fftlib.OPT_TEAM_NO_REPLAY, 1)
                _plans[key] = plan
            out = torch.empty_like(t)
            _plans[key].execute_ptr(t.data_ptr(), out.data_ptr())
        finally:
            if prev >= 0 and prev != t.device.index:
                fftlib.set_device(prev)
    return out
