"""GPU: rtus_selftest — the forward trace's correctly rounded division / square root sequences against the compiler's own
(same bits on 16 M operand pairs incl. every special value), and the structure of the depth-first bounding-box records
rtus_tree_kernel builds, for polyline lengths around every level boundary of the hierarchy."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _selftest(rtus, n_rays, n_math):
    counts = (C.c_ulonglong * 4)()
    lens = rtus.Params().lens()
    st = rtus.lib().rtus_selftest(C.byref(lens), n_rays, n_math, counts, 0)
    assert st == 0, st
    return list(counts)


def test_div_sqrt_sequences_return_the_correctly_rounded_bits(rtus):
    div_bad, sqrt_bad, tree_bad, n = _selftest(rtus, 905, 1 << 24)
    assert n == 1 << 24 and div_bad == 0 and sqrt_bad == 0 and tree_bad == 0


@pytest.mark.parametrize("n_rays", [8, 9, 63, 64, 65, 511, 512, 513, 905, 1810, 4095, 4096, 4097, 8192, 32768, 32769, 100003])
def test_tree_records_form_a_tree(rtus, n_rays):
    assert _selftest(rtus, n_rays, 0)[2] == 0


def test_selftest_rejects_bad_arguments(rtus):
    counts = (C.c_ulonglong * 4)()
    lens = rtus.Params().lens()
    assert rtus.lib().rtus_selftest(C.byref(lens), 4, 0, counts, 0) != 0
    assert rtus.lib().rtus_selftest(None, 905, 0, counts, 0) != 0
