"""CPU tests of the oracle (oracle/oracle_fft.c): pinned against the golden vectors generated from the REAL
reference (tests/golden/make_golden.py) and, where oracle/_ref has been built, against the reference itself."""
import numpy as np
import pytest

import oracle_lib as O


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.float64), np.asarray(b).view(np.float64))


def test_golden_n1024_bit_exact(golden):
    """The restatement reproduces the reference's outputs bit for bit (recurrence twiddles, C99, no contraction)."""
    x = golden["n1024_in"]
    for algo in ("dit", "dif", "radix4", "split_radix"):
        assert bits_equal(O.oracle_fft(x, -1, algo), golden["n1024_%s_fwd" % algo]), algo
        assert bits_equal(O.oracle_fft(x, +1, algo), golden["n1024_%s_inv" % algo]), algo


def test_golden_small_sizes_bit_exact(golden):
    for n in (32, 64, 128, 256, 512):
        x = golden["n%d_in" % n]
        assert bits_equal(O.oracle_fft(x, -1, "dit"), golden["n%d_dit_fwd" % n])
        assert bits_equal(O.oracle_fft(x, +1, "dit"), golden["n%d_dit_inv" % n])


@pytest.mark.parametrize("n", [65536, 1 << 18])
def test_golden_large_sampled(golden, n):
    bins = golden["n%d_bins" % n]
    for name, x in (("tone", O.gen_two_tone(n, 3, 1)[0]), ("lcg", O.gen_lcg(n, 3, 1)[0])):
        assert np.array_equal(x[:16], golden["n%d_%s_in_head" % (n, name)])  # generators are pinned too
        X = O.oracle_fft(x, -1, "dit")
        assert bits_equal(X[bins], golden["n%d_%s_fwd_bins" % (n, name)])
        assert float(np.linalg.norm(X)) == float(golden["n%d_%s_fwd_norm" % (n, name)])
        assert bits_equal(O.oracle_fft(x, +1, "dit")[bins], golden["n%d_%s_inv_bins" % (n, name)])
    f, g = O.two_tone_bins(n, 3)
    assert list(golden["n%d_tone_peaks" % n]) == [f, g]
    assert np.allclose(golden["n%d_tone_peak_vals" % n], [n, n / 2], rtol=1e-9)


def test_golden_bluestein(golden):
    for n in (31, 97, 1009):
        x = golden["blu%d_in" % n]
        assert bits_equal(O.oracle_fft(x, -1, "bluestein"), golden["blu%d_fwd" % n])
        assert bits_equal(O.oracle_fft(x, +1, "bluestein"), golden["blu%d_inv" % n])
        assert rel(golden["blu%d_fwd" % n], np.fft.fft(x)) < 1e-12


def test_bit_reverse_tables_and_reference_bug(golden):
    """The oracle's permutation equals the reference's bit_reverse() wherever the reference is right
    (log2n >= 5) and documents the reference bug for log2n <= 4, where it returns 0 for every input
    (harmless at n = 2, whose permutation is the identity)."""
    lib = O.oracle()
    for log2n in range(1, 13):
        ref_table = golden["bitrev_ref_log2n_%d" % log2n]
        mine = O.bit_reverse_table(log2n)
        asref = np.array([lib.oracle_bit_reverse_asref(i, log2n) for i in range(1 << log2n)], dtype=np.uint32)
        assert np.array_equal(asref, ref_table)  # bug-for-bug restatement matches the real thing
        assert sorted(mine) == list(range(1 << log2n))  # a permutation
        assert np.array_equal(mine[mine], np.arange(1 << log2n))  # an involution
        if log2n >= 5:
            assert np.array_equal(mine, ref_table), log2n
        else:
            assert not np.array_equal(mine, ref_table) and not ref_table.any()  # the reference returns 0 everywhere


def test_twiddle_and_size_helpers(golden):
    import ctypes as C
    lib = O.oracle()
    for k, n, d, re_, im_ in golden["twiddle_samples"]:
        a, b = C.c_double(), C.c_double()
        lib.oracle_twiddle_factor(int(k), int(n), int(d), C.byref(a), C.byref(b))
        assert (a.value, b.value) == (re_, im_)
    for v, w in zip(golden["np2_in"], golden["np2_out"]):
        assert lib.oracle_next_power_of_two(int(v)) == int(w)


def test_known_answer_n8(golden):
    """fft/fft.c:75 -- the reference's own radix-2 is wrong at n = 8 (bit_reverse bug), its N=8 demo is right."""
    x = golden["n8_known_in"]
    assert np.max(np.abs(O.oracle_fft(x, -1, "naive") - golden["n8_known_out_3dp"])) < 1e-3
    assert np.max(np.abs(O.oracle_fft(x, -1, "exact") - golden["n8_known_out_3dp"])) < 1e-3


@pytest.mark.parametrize("n", [32, 256, 4096])
def test_oracle_variants_agree_with_numpy(n):
    x = O.gen_lcg(n, 1, 2)
    want = np.fft.fft(x, axis=-1)
    for algo in ("dit", "dif", "radix4", "split_radix", "exact"):
        assert rel(O.oracle_fft(x, -1, algo), want) < 1e-13, algo
        assert rel(O.oracle_fft(x, +1, algo), np.fft.ifft(x, axis=-1)) < 1e-13, algo
    assert rel(O.oracle_fft_f32(x.astype(np.complex64), -1), want) < 5e-6
    assert O.oracle_fft(x[:, :24], -1, "bluestein").shape == (2, 24)
    with pytest.raises(ValueError):
        O.oracle_fft(x[:, :24], -1, "dit")  # non power of two: the reference exit()s, the oracle reports


def test_two_tone_generator_is_analytic():
    n = 4096
    x = O.gen_two_tone(n, 5, 3)
    X = np.fft.fft(x, axis=-1)
    for i in range(3):
        f, g = O.two_tone_bins(n, 5 + i)
        assert abs(X[i, f] - n) < 1e-8 and abs(X[i, g] - n / 2) < 1e-8
        X[i, f] = X[i, g] = 0
    assert np.max(np.abs(X)) < 1e-8
    assert np.array_equal(O.gen_two_tone(n, 5, 3, np.complex64), x.astype(np.complex64))


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference on this machine)")
def test_against_real_reference_when_present():
    rng = np.random.default_rng(3)
    for n in (32, 2048, 1 << 16):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for algo in ("dit", "dif", "radix4", "split_radix"):
            for d in (-1, 1):
                assert bits_equal(O.oracle_fft(x, d, algo), O.ref_fft(x, d, algo)), (n, algo, d)
    for n in (9, 100, 641):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for d in (-1, 1):
            assert bits_equal(O.oracle_fft(x, d, "bluestein"), O.ref_fft(x, d, "bluestein"))
    # n <= 8 pads to m <= 16, where the reference's own radix-2 is wrong (bit_reverse bug): the reference's
    # Bluestein is then wrong too, the oracle (correct permutation) is right -- parity is claimed for n >= 9 only
    x = rng.standard_normal(7) + 1j * rng.standard_normal(7)
    assert rel(O.oracle_fft(x, -1, "bluestein"), np.fft.fft(x)) < 1e-13
    assert rel(O.ref_fft(x, -1, "bluestein"), np.fft.fft(x)) > 0.1
