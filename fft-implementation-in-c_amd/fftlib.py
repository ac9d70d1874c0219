"""ctypes binding of the C-ABI product library (libfft_mi355x.so).

This is the same boundary a C caller links against (include/fft_gpu.h,
include/fft_auto.h, include/fft_hip.h).  There is no fallback of any kind: a
missing library raises, and every entry point fails (NULL / -1) without a GPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FFT_LIB_PATH") or os.path.join(HERE, "libfft_mi355x.so")  # FFT_LIB_PATH: tools A/B-ing two builds

FFT_FORWARD = -1
FFT_INVERSE = 1
FFT_GPU_AUTO = -1
FFT_GPU_HIP = 4
PREC_F64 = 0
PREC_F32 = 1
ALGO_AUTO, ALGO_RADIX2, ALGO_RADIX4, ALGO_SPLIT_RADIX, ALGO_RADIX2_GLOBAL, ALGO_BLUESTEIN, ALGO_RADIX2_SHFL = range(7)
ALGO_NAMES = {"auto": 0, "radix2": 1, "radix4": 2, "split_radix": 3, "radix2_global": 4, "bluestein": 5, "radix2_shfl": 6}
FFT_PREFER_GPU = 1 << 9
HIP_STREAM_LEGACY = 1  # hipStreamLegacy: the NULL / default stream named explicitly (fft_gpu_plan_set_stream: NULL = the plan's own)


class PlanInfo(C.Structure):
    _fields_ = [("n", C.c_int), ("batch", C.c_int), ("direction", C.c_int), ("precision", C.c_int),
                ("algo", C.c_int), ("device", C.c_int), ("bluestein_m", C.c_int), ("n_passes", C.c_int),
                ("factors", C.c_int * 4), ("chunk_batch", C.c_int), ("workspace_bytes", C.c_size_t),
                ("team_tiles", C.c_int), ("fused", C.c_int), ("team_kernel", C.c_int)]


# every symbol include/*.h declares, with its ctypes signature
_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int
SIGNATURES = {
    # include/fft_gpu.h
    "fft_gpu_init": (_i, [_i]), "fft_gpu_cleanup": (None, []), "fft_gpu_available": (_i, []),
    "fft_gpu_get_backend": (_i, []), "fft_gpu_alloc": (_vp, [_sz]), "fft_gpu_free": (None, [_vp]),
    "fft_gpu_copy_h2d": (None, [_vp, _vp, _sz]), "fft_gpu_copy_d2h": (None, [_vp, _vp, _sz]),
    "fft_gpu_plan_1d": (_vp, [_i, _i, _i]), "fft_gpu_execute": (None, [_vp, _vp, _vp]),
    "fft_gpu_destroy_plan": (None, [_vp]), "fft_gpu_dft_1d": (_i, [_vp, _vp, _i, _i]),
    "fft_gpu_dft_1d_batch": (_i, [_vp, _vp, _i, _i, _i]), "fft_gpu_plan_2d": (_vp, [_i, _i, _i]),
    "fft_gpu_dft_2d": (_i, [_vp, _vp, _i, _i, _i]), "fft_gpu_get_device_name": (C.c_char_p, []),
    "fft_gpu_get_memory_info": (None, [C.POINTER(_sz), C.POINTER(_sz)]), "fft_gpu_set_device": (_i, [_i]),
    # include/fft_hip.h part 1 (backend set)
    "fft_gpu_init_hip": (_i, []), "fft_gpu_cleanup_hip": (None, []), "fft_gpu_available_hip": (_i, []),
    "fft_gpu_alloc_hip": (_vp, [_sz]), "fft_gpu_free_hip": (None, [_vp]),
    "fft_gpu_copy_h2d_hip": (None, [_vp, _vp, _sz]), "fft_gpu_copy_d2h_hip": (None, [_vp, _vp, _sz]),
    "fft_gpu_plan_1d_hip": (_vp, [_i, _i, _i]), "fft_gpu_execute_hip": (None, [_vp, _vp, _vp, _i]),
    "fft_gpu_destroy_plan_hip": (None, [_vp]), "fft_gpu_get_device_name_hip": (C.c_char_p, []),
    "fft_gpu_get_memory_info_hip": (None, [C.POINTER(_sz), C.POINTER(_sz)]),
    "fft_gpu_dft_1d_hip": (_i, [_vp, _vp, _i, _i]),
    # include/fft_hip.h part 2 (additive, backend level)
    "fft_gpu_device_count_hip": (_i, []), "fft_gpu_set_device_hip": (_i, [_i]), "fft_gpu_get_device_hip": (_i, []),
    "fft_gpu_alloc_bytes_hip": (_vp, [_sz]), "fft_gpu_copy_h2d_bytes_hip": (_i, [_vp, _vp, _sz]),
    "fft_gpu_copy_d2h_bytes_hip": (_i, [_vp, _vp, _sz]), "fft_gpu_memory_ptr_hip": (_vp, [_vp]),
    "fft_gpu_memory_bytes_hip": (_sz, [_vp]), "fft_gpu_plan_1d_ex_hip": (_vp, [_i, _i, _i, _i, _i]),
    "fft_gpu_plan_info_hip": (_i, [_vp, C.POINTER(PlanInfo)]), "fft_gpu_plan_set_stream_hip": (_i, [_vp, _vp]),
    "fft_gpu_execute_ptr_hip": (_i, [_vp, _vp, _vp]), "fft_gpu_plan_sync_hip": (_i, [_vp]),
    "fft_gpu_plan_set_option_hip": (_i, [_vp, _i, _i]), "fft_gpu_set_policy_hip": (_i, [_i, _i, _i]),
    "fft_gpu_plan_2d_hip": (_vp, [_i, _i, _i]), "fft_gpu_plan_2d_ex_hip": (_vp, [_i, _i, _i, _i, _i]),
    "fft_gpu_plan_r2c_1d_hip": (_vp, [_i, _i, _i]), "fft_gpu_plan_c2r_1d_hip": (_vp, [_i, _i, _i]),
    "fft_gpu_plan_fused_hip": (_vp, [_i, _i, _i, _vp, _i, _i]), "fft_gpu_fused_out_len_hip": (_i, [_vp]),
    "fft_gpu_execute_fused_hip": (_i, [_vp, _vp, _vp, _vp, C.c_double]),
    "fft_gpu_host_register_hip": (_i, [_vp, _sz]), "fft_gpu_host_unregister_hip": (_i, [_vp]),
    "fft_gpu_host_is_registered_hip": (_i, [_vp]), "fft_gpu_copy_bench_hip": (C.c_double, [_sz, _i]), "fft_gpu_stream_bench_hip": (C.c_double, [_sz, _i, _i]),
    "fft_gpu_debug_counters_hip": (None, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "fft_gpu_plan_measure_hip": (_i, [_vp, _i]),
    "fft_gpu_plan_team_status_hip": (_i, [_vp]), "fft_gpu_plan_team_trace_hip": (_i, [_vp, _vp, _i]),
    "fft_gpu_execute_timed_hip": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_float)]),
    "fft_gpu_dft_1d_batch_hip": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "fft_gpu_profile_passes_hip": (_i, [_vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int), _i]),
    "fft_gpu_bit_reverse_hip": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    # include/fft_hip.h part 2 (additive, public)
    "fft_gpu_device_count": (_i, []), "fft_gpu_alloc_f32": (_vp, [_sz]),
    "fft_gpu_copy_h2d_f32": (None, [_vp, _vp, _sz]), "fft_gpu_copy_d2h_f32": (None, [_vp, _vp, _sz]),
    "fft_gpu_memory_ptr": (_vp, [_vp]), "fft_gpu_plan_1d_f32": (_vp, [_i, _i, _i]),
    "fft_gpu_plan_1d_ex": (_vp, [_i, _i, _i, _i, _i]), "fft_gpu_plan_info": (_i, [_vp, C.POINTER(PlanInfo)]),
    "fft_gpu_plan_set_stream": (_i, [_vp, _vp]), "fft_gpu_execute_async": (_i, [_vp, _vp, _vp]),
    "fft_gpu_execute_ptr": (_i, [_vp, _vp, _vp]), "fft_gpu_plan_sync": (_i, [_vp]),
    "fft_gpu_execute_timed": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_float)]),
    "fft_gpu_dft_1d_f32": (_i, [_vp, _vp, _i, _i]), "fft_gpu_dft_1d_batch_f32": (_i, [_vp, _vp, _i, _i, _i]),
    "fft_gpu_bit_reverse": (_i, [_vp, _vp, _i, _i, _i]),
    # include/fft_auto.h
    "fft_plan_dft_1d": (_vp, [_i, _vp, _vp, _i, C.c_uint]), "fft_execute": (None, [_vp]),
    "fft_execute_dft": (None, [_vp, _vp, _vp]), "fft_destroy_plan": (None, [_vp]),
    "fft_auto": (_i, [_vp, _vp, _i, _i]), "fft_plan_r2c_1d": (_vp, [_i, _vp, _vp, C.c_uint]),
    "fft_plan_c2r_1d": (_vp, [_i, _vp, _vp, C.c_uint]), "fft_plan_dft_2d": (_vp, [_i, _i, _vp, _vp, _i, C.c_uint]),
    "fft_export_wisdom_to_string": (_vp, []), "fft_import_wisdom_from_string": (_i, [C.c_char_p]),
    "fft_get_hardware_capabilities": (C.c_uint, []), "fft_plan_with_nthreads": (None, [_i]),
    "fft_alloc_complex": (_vp, [_sz]), "fft_alloc_real": (_vp, [_sz]), "fft_free": (None, [_vp]),
    "fft_version": (C.c_char_p, []), "fft_plan_measured_algo": (_i, [_vp]), "fft_plan_last_error": (_i, [_vp]), "fft_auto_cleanup": (None, []),
    # include/fft_apps.h, include/fft_utils.h
    "fft_convolution_gpu": (_i, [_vp, _i, _vp, _i, _vp]), "circular_convolution_gpu": (_i, [_vp, _vp, _i, _vp]),
    "compute_periodogram_gpu": (_vp, [_vp, _i, C.c_double]), "autocorrelation_fft_gpu": (_vp, [_vp, _i]),
    "cross_correlation_fft_gpu": (_vp, [_vp, _vp, _i]),
    "save_complex_array": (_i, [C.c_char_p, _vp, _i]), "load_complex_array": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_i)]),
    # include/fft_algorithms.h
    "radix2_dit_fft_gpu": (_i, [_vp, _i, _i]), "radix2_fft_gpu": (_i, [_vp, _i, _i]),
    "radix4_fft_gpu": (_i, [_vp, _i, _i]), "split_radix_fft_gpu": (_i, [_vp, _i, _i]),
    "bluestein_fft_gpu": (_i, [_vp, _i, _i]),
}

_lib = None


def load():
    """Load the product library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP extension missing: %s (build it: make -C %s)" % (LIB_PATH, HERE))
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib


def _prec_of(dtype):
    if dtype == np.complex64:
        return PREC_F32
    if dtype == np.complex128:
        return PREC_F64
    raise TypeError("complex64 or complex128 expected")


def init():
    lib = load()
    if lib.fft_gpu_init(FFT_GPU_AUTO) != 0:
        raise RuntimeError("fft_gpu_init failed: no usable MI355X/HIP device")
    return lib


EXP_LIB_PATH = os.path.join(HERE, "libfft_mi355x_exp.so")  # -DFFT_EXPERIMENTS build: kernel-variant switches, ablation bits
OPT_TEAM_FORCE_FALLBACK, OPT_TEAM_ENABLE, OPT_NO_FUSION, OPT_NO_CHAIN, OPT_TEAM_NO_REPLAY = 1, 2, 3, 4, 5


def set_device(index):
    """Make `index` the device new plans and buffers live on (fft_gpu_set_device; an 8-GPU process sets each in turn)."""
    if load().fft_gpu_set_device(int(index)) != 0:
        raise RuntimeError("fft_gpu_set_device(%d) failed" % index)


def get_device():
    """The device new plans and buffers currently live on (fft_gpu_get_device_hip)."""
    return load().fft_gpu_get_device_hip()


def set_policy(team=-1, min_batch=-1, chunk_mb=-1):
    """Planner policy for plans created from now on (fft_gpu_set_policy_hip); -1 keeps a value."""
    load().fft_gpu_set_policy_hip(team, min_batch, chunk_mb)


class DeviceBuffer:
    """Device memory owned through fft_gpu_alloc_bytes_hip / fft_gpu_free."""

    def __init__(self, nbytes):
        self.lib = load()
        self.handle = self.lib.fft_gpu_alloc_bytes_hip(nbytes)
        if not self.handle:
            raise MemoryError("device allocation of %d bytes failed" % nbytes)
        self.nbytes = nbytes

    @property
    def ptr(self):
        return self.lib.fft_gpu_memory_ptr_hip(self.handle)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if self.lib.fft_gpu_copy_h2d_bytes_hip(self.handle, arr.ctypes.data, arr.nbytes) != 0:
            raise RuntimeError("h2d failed")

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        if self.lib.fft_gpu_copy_d2h_bytes_hip(out.ctypes.data, self.handle, out.nbytes) != 0:
            raise RuntimeError("d2h failed")
        return out

    def free(self):
        if self.handle:
            self.lib.fft_gpu_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Plan:
    def __init__(self, n, batch, direction=FFT_FORWARD, dtype=np.complex64, algo=ALGO_AUTO):
        self.lib = init()
        self.n, self.batch, self.direction, self.dtype = n, batch, direction, np.dtype(dtype)
        self.handle = self.lib.fft_gpu_plan_1d_ex(n, batch, direction, _prec_of(self.dtype), algo)
        if not self.handle:
            raise RuntimeError("fft_gpu_plan_1d_ex(n=%d, batch=%d) failed" % (n, batch))

    def info(self):
        pi = PlanInfo()
        self.lib.fft_gpu_plan_info(self.handle, C.byref(pi))
        return pi

    def execute(self, buf_in, buf_out=None):
        buf_out = buf_out or buf_in
        self.lib.fft_gpu_execute(self.handle, buf_in.handle, buf_out.handle)

    def execute_ptr(self, d_in, d_out):
        if self.lib.fft_gpu_execute_ptr(self.handle, d_in, d_out) != 0:
            raise RuntimeError("fft_gpu_execute_ptr failed")

    def sync(self):
        return self.lib.fft_gpu_plan_sync(self.handle)

    def set_option(self, option, value):
        if self.lib.fft_gpu_plan_set_option_hip(self.handle, option, value) != 0:
            raise RuntimeError("fft_gpu_plan_set_option_hip(%d) failed" % option)

    def team_status(self):
        """0 team kernel did the last execute, 1 its fallback did, 2 barrier timeout, -1 no team kernel / never launched."""
        return self.lib.fft_gpu_plan_team_status_hip(self.handle)

    def timed(self, d_in, d_out, iters):
        ms = C.c_float()
        if self.lib.fft_gpu_execute_timed(self.handle, d_in, d_out, iters, C.byref(ms)) != 0:
            raise RuntimeError("fft_gpu_execute_timed failed")
        return ms.value

    def profile_passes(self, d_in, d_out, max_passes=4):
        ms = (C.c_float * max_passes)()
        cnt = (C.c_int * max_passes)()
        if self.lib.fft_gpu_profile_passes_hip(self.handle, d_in, d_out, ms, cnt, max_passes) != 0:
            raise RuntimeError("fft_gpu_profile_passes_hip failed")
        return [(ms[i], cnt[i]) for i in range(max_passes) if cnt[i]]

    def set_stream(self, stream_ptr):
        self.lib.fft_gpu_plan_set_stream(self.handle, stream_ptr)

    def destroy(self):
        if self.handle:
            self.lib.fft_gpu_destroy_plan(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


FUSED_KINDS = {"conv": 0, "circ": 1, "autocorr": 2, "xcorr": 3, "psd": 4}


class ExtPlan:
    """2D / r2c / c2r / fused plans (fft_hip.h): a thin owner of the handle; execute with raw device pointers."""

    def __init__(self, handle):
        self.lib = init()
        if not handle:
            raise RuntimeError("plan creation failed")
        self.handle = handle

    @classmethod
    def fft2d(cls, rows, cols, n_matrices=1, direction=FFT_FORWARD, dtype=np.complex128):
        return cls(init().fft_gpu_plan_2d_ex_hip(rows, cols, n_matrices, direction, _prec_of(np.dtype(dtype))))

    @classmethod
    def r2c(cls, n, batch=1, dtype=np.float64):
        return cls(init().fft_gpu_plan_r2c_1d_hip(n, batch, PREC_F32 if np.dtype(dtype) == np.float32 else PREC_F64))

    @classmethod
    def c2r(cls, n, batch=1, dtype=np.float64):
        return cls(init().fft_gpu_plan_c2r_1d_hip(n, batch, PREC_F32 if np.dtype(dtype) == np.float32 else PREC_F64))

    @classmethod
    def fused(cls, kind, nx, batch=1, h=None, dtype=np.complex128):
        dt = np.dtype(dtype)
        hh = None if h is None else np.ascontiguousarray(np.asarray(h).astype(dt))
        p = cls(init().fft_gpu_plan_fused_hip(FUSED_KINDS[kind], nx, 0 if hh is None else len(hh),
                                              None if hh is None else hh.ctypes.data, batch, _prec_of(dt)))
        p.out_len = p.lib.fft_gpu_fused_out_len_hip(p.handle)
        return p

    def execute_ptr(self, d_in, d_out):
        if self.lib.fft_gpu_execute_ptr(self.handle, d_in, d_out) != 0:
            raise RuntimeError("fft_gpu_execute_ptr failed")

    def set_option(self, option, value):
        if self.lib.fft_gpu_plan_set_option_hip(self.handle, option, value) != 0:
            raise RuntimeError("fft_gpu_plan_set_option_hip(%d) failed" % option)

    def info(self):
        pi = PlanInfo()
        self.lib.fft_gpu_plan_info(self.handle, C.byref(pi))
        return pi

    def execute_fused(self, d_x, d_y, d_out, sample_rate=1.0):
        if self.lib.fft_gpu_execute_fused_hip(self.handle, d_x, d_y, d_out, sample_rate) != 0:
            raise RuntimeError("fft_gpu_execute_fused_hip failed")

    def sync(self):
        return self.lib.fft_gpu_plan_sync(self.handle)

    def timed(self, d_in, d_out, iters):
        ms = C.c_float()
        if self.lib.fft_gpu_execute_timed(self.handle, d_in, d_out, iters, C.byref(ms)) != 0:
            raise RuntimeError("fft_gpu_execute_timed failed")
        return ms.value

    def destroy(self):
        if self.handle:
            self.lib.fft_gpu_destroy_plan(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _roundtrip(plan, x, out_shape, out_dtype, y=None, fused=False, fs=1.0):
    a = DeviceBuffer(x.nbytes)
    a.upload(x)
    out = np.empty(out_shape, dtype=out_dtype)
    o = DeviceBuffer(max(out.nbytes, 16))
    b = None
    if y is not None:
        b = DeviceBuffer(y.nbytes)
        b.upload(y)
    if fused:
        plan.execute_fused(a.ptr, b.ptr if b else None, o.ptr, fs)
    else:
        plan.execute_ptr(a.ptr, o.ptr)
    if plan.sync() != 0:
        raise RuntimeError("execute failed")
    res = o.download(out_shape, out_dtype)
    for buf in (a, o, b):
        if buf:
            buf.free()
    return res


def fft2d(x, direction=FFT_FORWARD):
    """x: [rows, cols] or [matrices, rows, cols] complex -> 2D transform of every matrix."""
    x = np.ascontiguousarray(x)
    x3 = x.reshape((-1,) + x.shape[-2:])
    plan = ExtPlan.fft2d(x3.shape[1], x3.shape[2], x3.shape[0], direction, x3.dtype)
    y = _roundtrip(plan, x3, x3.shape, x3.dtype)
    plan.destroy()
    return y.reshape(x.shape)


def rfft(x):
    """x: [batch, n] float32/float64 -> [batch, n//2 + 1] complex."""
    x = np.ascontiguousarray(x)
    batch, n = x.shape
    cdt = np.complex64 if x.dtype == np.float32 else np.complex128
    plan = ExtPlan.r2c(n, batch, x.dtype)
    y = _roundtrip(plan, x, (batch, n // 2 + 1), cdt)
    plan.destroy()
    return y


def irfft(X, n):
    """X: [batch, n//2 + 1] complex -> [batch, n] real, scaled by 1/n."""
    X = np.ascontiguousarray(X)
    rdt = np.float32 if X.dtype == np.complex64 else np.float64
    plan = ExtPlan.c2r(n, X.shape[0], rdt)
    y = _roundtrip(plan, X, (X.shape[0], n), rdt)
    plan.destroy()
    return y


def fused(kind, x, y=None, h=None, fs=1.0):
    """Batched fused consumer on host arrays: x [batch, nx] complex (see fft_hip.h fft_gpu_fused_t)."""
    x = np.ascontiguousarray(x)
    batch, nx = x.shape
    plan = ExtPlan.fused(kind, nx, batch, h, x.dtype)
    if kind == "psd":
        odt = np.float32 if x.dtype == np.complex64 else np.float64
    else:
        odt = x.dtype
    yy = None if y is None else np.ascontiguousarray(np.asarray(y).astype(x.dtype))
    res = _roundtrip(plan, x, (batch, plan.out_len), odt, y=yy, fused=True, fs=fs)
    plan.destroy()
    return res


def fft(x, direction=FFT_FORWARD, algo=ALGO_AUTO, inplace=True):
    """Host-array convenience: x [batch, n] (or [n]) complex64/complex128 -> transform on the GPU."""
    x = np.ascontiguousarray(x)
    squeeze = x.ndim == 1
    x2 = x.reshape(1, -1) if squeeze else x.reshape(-1, x.shape[-1])
    batch, n = x2.shape
    plan = Plan(n, batch, direction, x2.dtype, algo)
    a = DeviceBuffer(x2.nbytes)
    a.upload(x2)
    if inplace:
        plan.execute(a, a)
        out = a.download(x2.shape, x2.dtype)
    else:
        b = DeviceBuffer(x2.nbytes)
        plan.execute(a, b)
        out = b.download(x2.shape, x2.dtype)
        b.free()
    a.free()
    plan.destroy()
    return out.reshape(x.shape)
