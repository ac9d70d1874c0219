"""ctypes loader for the CPU emulation of the HIP kernels (tests/emu).  Test infra only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "fft-implementation-in-c_amd", "csrc")
_lib = None


def _needs_build(so):
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    srcs = [os.path.join(EMU_DIR, f) for f in os.listdir(EMU_DIR) if f.endswith((".cpp", ".h"))]
    srcs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("FFT_EMU_SO", os.path.join(EMU_DIR, "libfft_emu.so"))  # FFT_EMU_SO: a sanitizer build
        if "FFT_EMU_SO" not in os.environ and _needs_build(so):
            # several processes may get here at once (the ranks of a gloo test): one builds, under a lock, into a temporary
            # file that is renamed into place; the others wait for the lock and find the library fresh
            import fcntl
            with open(so + ".lock", "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                if _needs_build(so):
                    tmp = "%s.%d.tmp" % (so, os.getpid())
                    subprocess.run(["g++", "-O1", "-std=c++17", "-DFFT_EMU", "-DFFT_EXPERIMENTS", "-fPIC", "-shared", "-pthread", "-I" + CSRC,
                                    os.path.join(EMU_DIR, "emu_fft.cpp"), "-o", tmp], check=True)
                    os.replace(tmp, so)
        _lib = C.CDLL(so)
        _lib.emu_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_int)]
        _lib.emu_fft.restype = C.c_int
        _lib.emu_fft_team.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int)]
        _lib.emu_fft_team.restype = C.c_int
        _lib.emu_bitrev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    return _lib


def emu_fft(x, direction=-1, algo=0, lds_budget=0, inplace=False):
    """x: [batch, n] complex64/complex128.  Returns (result, info)."""
    x = np.ascontiguousarray(x)
    prec = 1 if x.dtype == np.complex64 else 0
    batch, n = x.shape
    info = (C.c_int * 8)()
    if inplace:
        out = x.copy()
        rc = lib().emu_fft(out.ctypes.data, out.ctypes.data, n, batch, direction, prec, algo, lds_budget, info)
    else:
        out = np.full_like(x, np.nan)
        rc = lib().emu_fft(x.ctypes.data, out.ctypes.data, n, batch, direction, prec, algo, lds_budget, info)
    if rc != 0:
        raise RuntimeError("emu_fft failed")
    return out, list(info)


def emu_fft_team(x, direction=-1, log2seats=2, n_xcc=2, threads=16, lds_budget=0, inplace=False, skew=False, tiles=4):
    """The team kernel (fft_team.h) with a small geometry: n_xcc "XCDs" of 2^log2seats workgroups of `threads` threads,
    all running concurrently.  The planner cuts each "XCD" into teams of n / (tiles * tile elements) workgroups
    (tiles = 4 as on the device; 1 or 2 force the few-tiles variants of the kernel).  info[0] = 100*tiles + passes
    when the team kernel was planned.  skew=True makes workgroup 0 report the wrong XCD: the kernel must give up and
    the two-pass fallback run."""
    os.environ["FFT_HIP_TEAM"] = "2"  # every size, any batch (the default only plans it where it measured faster)
    os.environ["FFT_HIP_TEAM_TILES"] = str(tiles)
    x = np.ascontiguousarray(x)
    prec = 1 if x.dtype == np.complex64 else 0
    batch, n = x.shape
    info = (C.c_int * 8)()
    mode = (log2seats + 1) | (n_xcc << 4) | (threads << 8) | ((1 << 20) if skew else 0)
    if inplace:
        out = x.copy()
        rc = lib().emu_fft_team(out.ctypes.data, out.ctypes.data, n, batch, direction, prec, lds_budget, mode, info)
    else:
        out = np.full_like(x, np.nan)
        rc = lib().emu_fft_team(x.ctypes.data, out.ctypes.data, n, batch, direction, prec, lds_budget, mode, info)
    if rc != 0:
        raise RuntimeError("emu_fft_team failed")
    return out, list(info)


def emu_fft2d(x, direction=-1, lds_budget=0, inplace=False):
    """x: [matrices, rows, cols] (or [rows, cols]) complex.  Returns (result, info); info[0] = 1 direct column pass, 2 transpose path."""
    x = np.ascontiguousarray(x)
    x3 = x.reshape((-1,) + x.shape[-2:])
    nm, rows, cols = x3.shape
    prec = 1 if x.dtype == np.complex64 else 0
    info = (C.c_int * 8)()
    out = x3.copy() if inplace else np.full_like(x3, np.nan)
    src = out if inplace else x3
    lib().emu_fft2d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_int)]
    if lib().emu_fft2d(src.ctypes.data, out.ctypes.data, rows, cols, nm, direction, prec, lds_budget, info) != 0:
        raise RuntimeError("emu_fft2d failed")
    return out.reshape(x.shape), list(info)


def emu_r2c(x):
    """x: [batch, n] float32/float64 -> [batch, n//2 + 1] complex."""
    x = np.ascontiguousarray(x)
    batch, n = x.shape
    prec = 1 if x.dtype == np.float32 else 0
    out = np.full((batch, n // 2 + 1), np.nan, dtype=np.complex64 if prec else np.complex128)
    lib().emu_real.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4
    if lib().emu_real(x.ctypes.data, out.ctypes.data, n, batch, 1, prec) != 0:
        raise RuntimeError("emu_real failed")
    return out


def emu_c2r(X, n):
    """X: [batch, n//2 + 1] complex -> [batch, n] real (scaled by 1/n)."""
    X = np.ascontiguousarray(X)
    batch = X.shape[0]
    prec = 1 if X.dtype == np.complex64 else 0
    out = np.full((batch, n), np.nan, dtype=np.float32 if prec else np.float64)
    lib().emu_real.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4
    if lib().emu_real(X.ctypes.data, out.ctypes.data, n, batch, 0, prec) != 0:
        raise RuntimeError("emu_real failed")
    return out


FUSED = {"conv": 0, "circ": 1, "autocorr": 2, "xcorr": 3, "psd": 4}


def emu_fused(kind, x, y=None, h=None, lds_budget=0, no_fusion=False, fs=1.0):
    """x: [batch, nx] complex; h: [nh] kernel (conv / circ); y: [batch, nx] (xcorr).  Returns (result, info);
    info = [passes, fused?, log2 m]."""
    x = np.ascontiguousarray(x)
    batch, nx = x.shape
    prec = 1 if x.dtype == np.complex64 else 0
    k = FUSED[kind]
    nh = 0 if h is None else len(h)
    if h is not None:
        h = np.ascontiguousarray(h.astype(x.dtype))
    if y is not None:
        y = np.ascontiguousarray(y.astype(x.dtype))
    if kind == "conv":
        out = np.full((batch, nx + nh - 1), np.nan, dtype=x.dtype)
    elif kind == "psd":
        out = np.full((batch, nx // 2 + 1), np.nan, dtype=np.float32 if prec else np.float64)
    else:
        out = np.full((batch, nx), np.nan, dtype=x.dtype)
    info = (C.c_int * 8)()
    f = lib().emu_fused
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                  C.POINTER(C.c_int)]
    rc = f(k, x.ctypes.data, None if y is None else y.ctypes.data, None if h is None else h.ctypes.data, nx, nh, out.ctypes.data, batch, prec,
           lds_budget, 1 if no_fusion else 0, fs, info)
    if rc != 0:
        raise RuntimeError("emu_fused failed")
    return out, list(info)
