#!/usr/bin/env python3
"""Golden vectors for the "next" rows from the REAL reference's consumers of the transform, compiled by
`make -C oracle ref` into oracle/_ref/libref_{conv,psd,image,utils}.so (applications/convolution.c,
applications/power_spectrum.c, applications/image_fft.c, utils/fft_utils.c).  Build container only.  The outputs are
DATA -- inputs and what the reference computed for them, including one text file written by its save_complex_array.

    python tests/golden/make_golden_apps.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REFDIR = os.path.join(O.ORACLE_DIR, "_ref")
vp = C.c_void_p


def lcg(n, seed):
    return O.gen_lcg(n, seed, 1)[0]


def main():
    out = {}
    conv = C.CDLL(os.path.join(REFDIR, "libref_conv.so"))
    psd = C.CDLL(os.path.join(REFDIR, "libref_psd.so"))
    img = C.CDLL(os.path.join(REFDIR, "libref_image.so"))
    utils = C.CDLL(os.path.join(REFDIR, "libref_utils.so"))

    # fft_convolution(x, nx, h, nh, y)  applications/convolution.c:34-69
    for nx, nh in ((100, 17), (256, 256), (1, 1), (33, 5)):
        x, h = lcg(nx, nx), lcg(nh, nh + 1000)
        y = np.zeros(nx + nh - 1, dtype=np.complex128)
        conv.fft_convolution.argtypes = [vp, C.c_int, vp, C.c_int, vp]
        conv.fft_convolution(x.ctypes.data, nx, h.ctypes.data, nh, y.ctypes.data)
        out["conv_%d_%d_x" % (nx, nh)] = x
        out["conv_%d_%d_h" % (nx, nh)] = h
        out["conv_%d_%d_y" % (nx, nh)] = y
    # circular_convolution(x, h, n, y)  :72-96
    for n in (64, 1024):
        x, h = lcg(n, n + 1), lcg(n, n + 2)
        y = np.zeros(n, dtype=np.complex128)
        conv.circular_convolution.argtypes = [vp, vp, C.c_int, vp]
        conv.circular_convolution(x.ctypes.data, h.ctypes.data, n, y.ctypes.data)
        out["circ_%d_x" % n], out["circ_%d_h" % n], out["circ_%d_y" % n] = x, h, y

    # compute_periodogram(signal, n, fs) -> malloc'd double[n/2+1]  applications/power_spectrum.c:58-86
    for n in (256, 4096):
        x = lcg(n, n + 3)
        psd.compute_periodogram.argtypes = [vp, C.c_int, C.c_double]
        psd.compute_periodogram.restype = C.POINTER(C.c_double)
        p = psd.compute_periodogram(x.ctypes.data, n, 48000.0)
        out["psd_%d_x" % n] = x
        out["psd_%d_out" % n] = np.array([p[i] for i in range(n // 2 + 1)])
    # autocorrelation_fft / cross_correlation_fft -> allocate_complex_array(n)  :133-190
    for n in (100, 1000):
        x, y = lcg(n, n + 4), lcg(n, n + 5)
        for name, args in (("autocorrelation_fft", (x,)), ("cross_correlation_fft", (x, y))):
            f = getattr(psd, name)
            f.argtypes = [vp] * len(args) + [C.c_int]
            f.restype = C.POINTER(C.c_double)
            r = f(*[a.ctypes.data for a in args], n)
            out["%s_%d" % (name, n)] = np.array([r[2 * i] + 1j * r[2 * i + 1] for i in range(n)])
        out["corr_%d_x" % n], out["corr_%d_y" % n] = x, y

    # fft_2d(complex_t** data, rows, cols, dir)  applications/image_fft.c:35-72 (in place on an array of row pointers)
    for rows, cols in ((32, 64), (128, 32)):  # both >= 32: the reference's own 1D transform is wrong for n in {4, 8, 16}
        x = lcg(rows * cols, rows * 7 + cols).reshape(rows, cols)
        for d, tag in ((-1, "fwd"), (1, "inv")):
            m = np.ascontiguousarray(x.copy())
            ptrs = (vp * rows)(*[m[i].ctypes.data for i in range(rows)])
            img.fft_2d.argtypes = [vp, C.c_int, C.c_int, C.c_int]
            img.fft_2d(ptrs, rows, cols, d)
            out["fft2d_%dx%d_%s" % (rows, cols, tag)] = m
        out["fft2d_%dx%d_in" % (rows, cols)] = x

    # save_complex_array(filename, data, n)  utils/fft_utils.c:77-97: the text the reference writes
    n = 12
    x = lcg(n, 99) * 1000.0
    path = os.path.join(HERE, "reference_saved_array.txt")
    utils.save_complex_array.argtypes = [C.c_char_p, vp, C.c_int]
    assert utils.save_complex_array(path.encode(), x.ctypes.data, n) == 0
    out["saved_array_values"] = x

    dst = os.path.join(HERE, "reference_apps_vectors.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "arrays;", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
