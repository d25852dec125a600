"""Device memory, streams and events on top of the C-ABI (no PyTorch)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class Stream(object):
  def __init__(self, high_priority=False):
    """`high_priority`: the device dispatches this stream's workgroups before those of normal
    streams when both have work pending (hipStreamCreateWithPriority)."""
    _lib.require_device()
    h = C.c_void_p()
    if high_priority:
      check(lib.pm_stream_create_priority(C.byref(h), 1))
    else:
      check(lib.pm_stream_create(C.byref(h)))
    self.handle = h

  def sync(self):
    check(lib.pm_stream_sync(self.handle))

  def wait(self, event):
    """Work submitted to this stream from now on waits for `event` (hipStreamWaitEvent)."""
    check(lib.pm_stream_wait_event(self.handle, event.handle))

  def __del__(self):
    try:
      if self.handle:
        lib.pm_stream_destroy(self.handle)
        self.handle = None
    except Exception:
      pass


def _sh(stream):
  return stream.handle if stream is not None else None


class Event(object):
  def __init__(self):
    _lib.require_device()
    h = C.c_void_p()
    check(lib.pm_event_create(C.byref(h)))
    self.handle = h

  def record(self, stream=None):
    check(lib.pm_event_record(self.handle, _sh(stream)))

  def sync(self):
    check(lib.pm_event_sync(self.handle))

  def elapsed_ms(self, later):
    ms = C.c_float(0)
    check(lib.pm_event_elapsed_ms(self.handle, later.handle, C.byref(ms)))
    return ms.value

  def __del__(self):
    try:
      if self.handle:
        lib.pm_event_destroy(self.handle)
        self.handle = None
    except Exception:
      pass


class LaunchTimer(object):
  """HIP events around every launch of a driver run (`ens.timer = LaunchTimer()`): the run-average
  duration of each kernel IN SITU -- every launch on the state and forcing the run gives it --
  which is what a roofline figure of a coupled config has to be computed from (replaying a
  kernel on the run's final state is up to 25 % off).  Events are recorded on the stream the
  kernel is launched on; `summary()` synchronises."""

  class _Span(object):
    def __init__(self, timer, name, stream):
      self.timer, self.name, self.stream = timer, name, stream

    def __enter__(self):
      self.e0, self.e1 = Event(), Event()
      self.e0.record(self.stream)
      return self

    def __exit__(self, et, ev, tb):
      self.e1.record(self.stream)
      self.timer.spans.append((self.name, self.e0, self.e1))
      return False

  def __init__(self):
    self.spans = []

  def span(self, name, stream=None):
    return LaunchTimer._Span(self, name, stream)

  def null_span_ms(self, stream=None, reps=32):
    """What a span measures with NOTHING between its two event records (the events' own
    packets in the queue): the median of `reps` empty spans on `stream`.  `summary()` subtracts
    it, so that a span's duration is the kernel's own -- what rocprofv3's kernel trace
    reports -- and not kernel + event overhead (2-3 us, 7-10 % of a 30 us launch)."""
    spans = []
    for _ in range(reps):
      e0, e1 = Event(), Event()
      e0.record(stream)
      e1.record(stream)
      spans.append((e0, e1))
    synchronize()
    v = sorted(e0.elapsed_ms(e1) for e0, e1 in spans)
    return v[len(v) // 2]

  def summary(self, stream=None, subtract_null=True):
    """{name: (launches, total ms)}; every span less the empty-span time (`null_span_ms`)."""
    synchronize()
    null = self.null_span_ms(stream) if (subtract_null and self.spans) else 0.0
    self.null_ms = null
    out = {}
    for name, e0, e1 in self.spans:
      n, t = out.get(name, (0, 0.0))
      out[name] = (n + 1, t + max(e0.elapsed_ms(e1) - null, 0.0))
    return out


class _NoSpan(object):
  def __enter__(self):
    return self

  def __exit__(self, et, ev, tb):
    return False


_NO_SPAN = _NoSpan()


def launch_span(timer, name, stream=None):
  """`with launch_span(self.timer, "k_thermwind", stream): launch` -- a no-op without a timer."""
  return _NO_SPAN if timer is None else timer.span(name, stream)


class Graph(object):
  """A captured launch sequence (hipGraph): `with Graph.capture(stream) as g: ...`."""

  def __init__(self, handle):
    self.handle = handle

  class _Capture(object):
    def __init__(self, stream):
      self.stream = stream
      self.graph = None

    def __enter__(self):
      check(lib.pm_graph_begin_capture(_sh(self.stream)))
      return self

    def __exit__(self, et, ev, tb):
      h = C.c_void_p()
      rc = lib.pm_graph_end_capture(_sh(self.stream), C.byref(h))
      if et is None:
        check(rc)
        self.graph = Graph(h)
      return False

  @staticmethod
  def capture(stream=None):
    return Graph._Capture(stream)

  def launch(self, stream=None):
    check(lib.pm_graph_launch(self.handle, _sh(stream)))

  def __del__(self):
    try:
      if self.handle:
        lib.pm_graph_destroy(self.handle)
        self.handle = None
    except Exception:
      pass


class DeviceArray(object):
  """A dense fp64 / int32 array in HBM, owned by this object."""

  def __init__(self, shape, dtype=np.float64):
    _lib.require_device()
    self.shape = tuple(int(s) for s in np.atleast_1d(shape))
    self.dtype = np.dtype(dtype)
    self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
    p = C.c_void_p()
    check(lib.pm_malloc(C.byref(p), self.nbytes))
    self.ptr = p.value or 0

  @classmethod
  def from_host(cls, arr, dtype=None, stream=None):
    arr = np.ascontiguousarray(arr, dtype=dtype if dtype is not None else
                               (np.int32 if np.asarray(arr).dtype.kind in "iub"
                                else np.float64))
    d = cls(arr.shape, arr.dtype)
    d.upload(arr, stream)
    return d

  @classmethod
  def zeros(cls, shape, dtype=np.float64, stream=None):
    d = cls(shape, dtype)
    check(lib.pm_memset(d.ptr, 0, d.nbytes, _sh(stream)))
    # complete before returning: a fill enqueued on one stream must not be overtaken by a
    # kernel that a caller launches on another (streams here are non-blocking)
    check(lib.pm_stream_sync(_sh(stream)))
    return d

  def upload(self, arr, stream=None):
    arr = np.ascontiguousarray(arr, dtype=self.dtype)
    if arr.nbytes != self.nbytes:
      raise ValueError("upload size mismatch: %r vs %r" % (arr.shape, self.shape))
    check(lib.pm_memcpy_h2d(self.ptr, arr.ctypes.data, self.nbytes, _sh(stream)))

  def download(self, out=None, stream=None):
    if out is None:
      out = np.empty(self.shape, dtype=self.dtype)
    if out.nbytes != self.nbytes or not out.flags.c_contiguous or out.dtype != self.dtype:
      raise ValueError("download buffer mismatch")
    check(lib.pm_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, _sh(stream)))
    return out

  def copy_from(self, other, stream=None):
    if other.nbytes != self.nbytes:
      raise ValueError("copy size mismatch")
    check(lib.pm_memcpy_d2d(self.ptr, other.ptr, self.nbytes, _sh(stream)))

  def view(self, first_row, nrows):
    """Rows [first_row, first_row + nrows) of a 2-D array as an array of their own that does
    not own the memory (the parent must outlive it)."""
    v = object.__new__(DeviceArray)
    v.shape = (int(nrows),) + self.shape[1:]
    v.dtype = self.dtype
    row = int(np.prod(self.shape[1:])) * self.dtype.itemsize
    v.nbytes = int(nrows) * row
    v.ptr = self.ptr + int(first_row) * row
    v._parent = self
    return v

  def free(self):
    if getattr(self, "_parent", None) is not None:
      self.ptr = 0
      return
    if getattr(self, "ptr", 0):
      lib.pm_free(self.ptr)
      self.ptr = 0

  def __del__(self):
    try:
      self.free()
    except Exception:
      pass


class PinnedArray(object):
  """A page-locked host array (hipHostMalloc) as the target of asynchronous device-to-host
  copies: `dev.download_async(pinned, stream)` returns at once, `pinned.array` is valid after
  the stream (or an event recorded behind the copy) has been synchronised."""

  def __init__(self, shape, dtype=np.float64):
    _lib.require_device()
    self.shape = tuple(int(s) for s in np.atleast_1d(shape))
    self.dtype = np.dtype(dtype)
    self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
    p = C.c_void_p()
    check(lib.pm_host_alloc(C.byref(p), self.nbytes))
    self.ptr = p.value or 0
    if self.nbytes:
      buf = (C.c_char * self.nbytes).from_address(self.ptr)
      self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)
    else:
      self.array = np.empty(self.shape, self.dtype)

  def free(self):
    if getattr(self, "ptr", 0):
      self.array = None
      lib.pm_host_free(self.ptr)
      self.ptr = 0

  def __del__(self):
    try:
      self.free()
    except Exception:
      pass


def download_async(src_ptr, nbytes, pinned, stream=None, offset=0):
  """`nbytes` from device address `src_ptr` into `pinned` at byte `offset`, ordered on `stream`,
  no host synchronisation."""
  if offset + nbytes > pinned.nbytes:
    raise ValueError("pinned buffer too small")
  check(lib.pm_memcpy_d2h_async(pinned.ptr + offset, src_ptr, nbytes, _sh(stream)))


def rows_pack(items, nrows, sel=None, stream=None):
  """One launch of pm_rows_pack: items = [(src address, dst address, nlev, src_stride)], rows
  `sel` (a device int32 array / address, or None = rows 0..nrows-1)."""
  from ._lib import pm_row_copy, PM_PACK_MAX_ITEMS
  for i0 in range(0, len(items), PM_PACK_MAX_ITEMS):
    chunk = items[i0:i0 + PM_PACK_MAX_ITEMS]
    arr = (pm_row_copy * len(chunk))()
    for k, (src, dst, nlev, stride) in enumerate(chunk):
      arr[k].src, arr[k].dst, arr[k].nlev, arr[k].src_stride = src, dst, int(nlev), int(stride)
    selp = None if sel is None else (sel if isinstance(sel, int) else sel.ptr)
    check(lib.pm_rows_pack(arr, len(chunk), selp, int(nrows), _sh(stream)))


def _addr(x):
  return x if isinstance(x, int) else x.ptr


def synchronize():
  check(lib.pm_device_sync())
