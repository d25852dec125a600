"""Device-side self tests of the arithmetic building blocks."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("emax", [40, 200, 400])
def test_division_by_precomputed_reciprocal_is_correctly_rounded(gpu, emax):
  """div_by_recip (common.hip.h) == IEEE `/`, bit for bit, on ~4e9 random operand pairs
  per exponent range (3/8 of the mantissas at the all-ones / all-zeros / half edges)."""
  from pymoc_amd._lib import lib, check
  tested, bad = C.c_uint64(0), C.c_uint64(0)
  check(lib.pm_selftest_fastdiv(20240 + emax, 2048, 2000, emax, C.byref(tested), C.byref(bad)))
  assert tested.value == 4 * 2000 * 256 * 2048
  assert bad.value == 0, "%d of %d quotients differ" % (bad.value, tested.value)
