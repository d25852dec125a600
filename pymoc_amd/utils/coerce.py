"""Host-side coercion helpers with the reference's contract.

Mirrors `pymoc.utils.make_func` (src/pymoc/utils/make_func.py:4-45) and
`pymoc.utils.make_array` (src/pymoc/utils/make_array.py:4-37): accepted types are
callable / numpy.ndarray / float (ints are rejected), ndarray inputs are NOT copied
(the returned closure / array aliases the caller's object), and the error is the
2-tuple TypeError whose text the reference's tests pin.
"""
import numpy as np

_MSG = 'needs to be either function, numpy array, or float'


def make_func(myst, axis, name):
  """callable | ndarray on `axis` | float  ->  callable f(x)."""
  if callable(myst):
    return myst
  if isinstance(myst, np.ndarray):
    fn = lambda x: np.interp(x, axis, myst)
    fn._pm_source = myst  # the aliased array (lets callers detect in-place edits cheaply)
    return fn
  if isinstance(myst, float):
    fn = lambda x: myst + 0 * x
    fn._pm_source = myst
    return fn
  raise TypeError(name, _MSG)


def make_array(myst, axis, name):
  """ndarray | callable | float  ->  ndarray along `axis`."""
  if isinstance(myst, np.ndarray):
    return myst
  if callable(myst):
    return myst(axis)
  if isinstance(myst, float):
    return myst + 0 * axis
  raise TypeError(name, _MSG)


def check_numpy_version():
  """True when NumPy can differentiate on non-uniform grids (np.gradient with coordinate
  arrays, NumPy >= 1.13) -- the reference's guard (utils/check_numpy_version.py:4-22)."""
  major, minor = (int(p) for p in np.version.version.split('.')[:2])
  return (major, minor) >= (1, 13)
