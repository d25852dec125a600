"""ctypes binding of libpymoc_hip.so (the C-ABI declared in include/pymoc_hip.h).

The engine has no CPU fallback: if the shared library is missing, importing this module
raises; if no HIP device is visible, the first call that needs one raises.
"""
import ctypes as C
import os

# dmabuf IPC (the only kind this platform's driver supports): RCCL and any cross-process
# sharing of device memory need it, and the HSA runtime reads it when the engine is loaded --
# so it is set here, before the dlopen below, whoever started this process (pymoc_amd.launch,
# torch.distributed.run, a plain shell); an explicit setting by the caller wins
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYMOC_HIP_LIB: another build of the same sources (the phase-clock build of profiles/, an A/B
# variant); the product loads the library next to this file
LIB_PATH = os.environ.get("PYMOC_HIP_LIB") or os.path.join(_HERE, "libpymoc_hip.so")

PM_OK, PM_EINVAL, PM_EHIP, PM_ENCCL, PM_ENODEV = 0, 1, 2, 3, 4

PM_COL_DO_CONV, PM_COL_BZBOT, PM_COL_STATIC_IN_RANGE, PM_COL_UNIFORM_AREA = 1, 2, 4, 8
PM_EQ_HFREE, PM_EQ_HAS_BBOT, PM_EQ_KAPPA_ARRAY, PM_EQ_PSI_ARRAY = 1, 2, 4, 8
PM_OP_CONVECT, PM_OP_VERTADVDIFF, PM_OP_HORADV, PM_OP_TIMESTEP, PM_OP_WEFF = 1, 2, 4, 7, 8
PM_OP_CONTRACTED = 16
PM_OP_WA_PSI = 32
PM_OP_WA_TWOBASIN = 64
PM_COLS_ALL_UNIFORM_AREA = 1
PM_COLS_DIV3_PROVEN = 2

c_dp = C.c_void_p  # device pointers travel as plain addresses


class PmError(RuntimeError):
  def __init__(self, code, text):
    super().__init__("pymoc_hip error %d: %s" % (code, text))
    self.code = code


class pm_columns(C.Structure):
  """Mirror of `struct pm_columns` (include/pymoc_hip.h)."""
  _fields_ = [
      ("ncols", C.c_int32), ("nz", C.c_int32), ("nsel", C.c_int32),
      ("reserved", C.c_int32), ("z", c_dp), ("b", c_dp), ("kappa", c_dp),
      ("area", c_dp), ("dAkappa", c_dp), ("bs", c_dp), ("bbot", c_dp), ("bzbot", c_dp),
      ("N2min", c_dp), ("flags", c_dp), ("ksel", c_dp), ("nonfinite", c_dp),
      ("kappa_base", c_dp), ("kappa_profile", c_dp)
  ]


PM_TW_SOLVE, PM_TW_PSIB, PM_TW_PSIBZ, PM_TW_WA_PSI = 1, 2, 4, 8


class pm_thermwind(C.Structure):
  """Mirror of `struct pm_thermwind` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("nz", C.c_int32), ("nb", C.c_int32), ("reserved", C.c_int32),
      ("z", c_dp), ("b1", c_dp), ("b2", c_dp), ("f", c_dp), ("Psi", c_dp),
      ("bgrid", c_dp), ("psib", c_dp), ("psibz1", c_dp), ("psibz2", c_dp),
      ("Psi_SO", c_dp), ("wA1", c_dp), ("wA2", c_dp), ("b1_mid", c_dp), ("b2_mid", c_dp),
      ("dPsi", c_dp)
  ]


(PM_SO_HAS_C, PM_SO_BVP_WITH_EK, PM_SO_HAS_HSILL, PM_SO_HAS_HEK, PM_SO_HAS_HTAPERTOP,
 PM_SO_HAS_HTAPERBOT, PM_SO_TAU_ARRAY) = 1, 2, 4, 8, 16, 32, 64
PM_SO_OP_EKMAN, PM_SO_OP_GM, PM_SO_OP_SOLVE = 1, 2, 3
PM_JN_UNIFORM_AREA, PM_JN_CONTRACTED, PM_JN_SHARED_COEF, PM_JN_SPLIT_LANES = 1, 2, 4, 8
PM_JN_DIV3_PROVEN = 16


class pm_psi_so(C.Structure):
  """Mirror of `struct pm_psi_so` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("nz", C.c_int32), ("ny", C.c_int32), ("flags", C.c_int32),
      ("bvp_refine", C.c_int32), ("reserved", C.c_int32),
      ("z", c_dp), ("y", c_dp), ("b", c_dp), ("bs", c_dp), ("tau", c_dp), ("KGM", c_dp),
      ("f", C.c_double), ("rho", C.c_double), ("L", C.c_double), ("smax", C.c_double),
      ("c", C.c_double), ("Hsill", C.c_double), ("HEk", C.c_double),
      ("Htapertop", C.c_double), ("Htaperbot", C.c_double),
      ("Psi", c_dp), ("Psi_Ek", c_dp), ("Psi_GM", c_dp), ("Ek_raw", c_dp),
      ("GM_raw", c_dp), ("ys", c_dp), ("status", c_dp), ("ys_in", c_dp),
      ("tau_ave_in", c_dp)
  ]


class pm_so_ml(C.Structure):
  """Mirror of `struct pm_so_ml` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("nz", C.c_int32), ("ny", C.c_int32), ("reserved", C.c_int32),
      ("y", c_dp), ("bs", c_dp), ("Psi_s", c_dp), ("b_basin", c_dp), ("Psi_b", c_dp),
      ("surflux", c_dp), ("rest_mask", c_dp), ("b_rest", c_dp),
      ("Ks", C.c_double), ("h", C.c_double), ("L", C.c_double), ("v_pist", C.c_double),
      ("status", c_dp)
  ]


class pm_column_equi(C.Structure):
  """Mirror of `struct pm_column_equi` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("nz", C.c_int32), ("mmax", C.c_int32), ("reserved", C.c_int32),
      ("m", c_dp), ("active", c_dp), ("x", c_dp), ("Ak", c_dp), ("dAk", c_dp), ("wA", c_dp),
      ("wA_z", c_dp), ("z", c_dp), ("bs", c_dp), ("bbot", c_dp), ("bzbot", c_dp),
      ("flags", c_dp), ("zidx", c_dp), ("tol", C.c_double), ("y", c_dp), ("rms", c_dp),
      ("nadd", c_dp), ("b", c_dp), ("bz", c_dp)
  ]


class pm_equi_column(C.Structure):
  """Mirror of `struct pm_equi_column` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("nzg", C.c_int32), ("mmax", C.c_int32), ("reserved", C.c_int32),
      ("m", c_dp), ("active", c_dp), ("x", c_dp), ("y", c_dp), ("yp", c_dp), ("p", c_dp),
      ("f", c_dp), ("A", c_dp), ("bs", c_dp), ("bb", c_dp), ("kappa", c_dp), ("flags", c_dp),
      ("zg", c_dp), ("kappa_z", c_dp), ("dkappa_z", c_dp), ("psi_z", c_dp),
      ("tol", C.c_double), ("rms", c_dp), ("nadd", c_dp), ("status", c_dp), ("niter", c_dp),
      ("info", c_dp), ("scratch", c_dp)
  ]


class pm_jn2018_bc(C.Structure):
  """Mirror of `struct pm_jn2018_bc` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("nz", C.c_int32), ("ny", C.c_int32), ("reserved", C.c_int32),
      ("Psi_SO", c_dp), ("Psi_res_b", c_dp), ("Psi_res_n", c_dp), ("b_basin", c_dp),
      ("b_north", c_dp), ("bs_SO", c_dp), ("bbot", c_dp), ("ksel", c_dp)
  ]


class pm_jn2018(C.Structure):
  """Mirror of `struct pm_jn2018` (include/pymoc_hip.h)."""
  _fields_ = [
      ("n", C.c_int32), ("hints", C.c_int32), ("reserved1", C.c_int32),
      ("reserved2", C.c_int32), ("cols", pm_columns), ("wA", c_dp), ("Psi_SO", c_dp),
      ("Psi_res_b", c_dp), ("Psi_res_n", c_dp), ("ml", pm_so_ml)
  ]


PM_PACK_MAX_ITEMS = 8


class pm_row_copy(C.Structure):
  """Mirror of `struct pm_row_copy` (include/pymoc_hip.h)."""
  _fields_ = [("src", c_dp), ("dst", c_dp), ("nlev", C.c_int32), ("src_stride", C.c_int32)]


class pm_run_schedule(C.Structure):
  """Mirror of `struct pm_run_schedule` (include/pymoc_hip.h)."""
  _fields_ = [("n_first", C.c_int32), ("n_updates", C.c_int32), ("m_steps", C.c_int32),
              ("n_last", C.c_int32)]


class pm_twocol_loop(C.Structure):
  """Mirror of `struct pm_twocol_loop` (include/pymoc_hip.h)."""
  _fields_ = [("cols", pm_columns), ("tw", pm_thermwind), ("wA", c_dp), ("dt", C.c_double),
              ("sched", pm_run_schedule), ("status", c_dp)]


class pm_jn2018_loop(C.Structure):
  """Mirror of `struct pm_jn2018_loop` (include/pymoc_hip.h)."""
  _fields_ = [("jn", pm_jn2018), ("tw", pm_thermwind), ("so", pm_psi_so), ("dt", C.c_double),
              ("sched", pm_run_schedule)]


if not os.path.exists(LIB_PATH):
  raise ImportError(
      "pymoc_amd: %s is missing. Build it with `make lib` (hipcc --offload-arch=gfx950) "
      "or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU "
      "fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

# name -> (restype, argtypes); every symbol include/pymoc_hip.h declares
SIGNATURES = {
    "pm_version": (C.c_char_p, []),
    "pm_last_error": (C.c_char_p, []),
    "pm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pm_set_device": (C.c_int, [C.c_int]),
    "pm_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int),
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "pm_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "pm_free": (C.c_int, [C.c_void_p]),
    "pm_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "pm_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "pm_host_free": (C.c_int, [C.c_void_p]),
    "pm_memcpy_d2h_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_rows_pack": (C.c_int, [C.c_void_p, C.c_int32, c_dp, C.c_int32, C.c_void_p]),
    "pm_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "pm_stream_create_priority": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "pm_stream_destroy": (C.c_int, [C.c_void_p]),
    "pm_stream_sync": (C.c_int, [C.c_void_p]),
    "pm_device_sync": (C.c_int, []),
    "pm_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "pm_event_destroy": (C.c_int, [C.c_void_p]),
    "pm_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pm_event_sync": (C.c_int, [C.c_void_p]),
    "pm_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pm_twocol_forcing": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "pm_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "pm_graph_begin_capture": (C.c_int, [C.c_void_p]),
    "pm_graph_end_capture": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pm_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pm_graph_destroy": (C.c_int, [C.c_void_p]),
    "pm_column_steps": (C.c_int, [C.POINTER(pm_columns), c_dp, c_dp, c_dp, C.c_double,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pm_column_weff": (C.c_int, [C.POINTER(pm_columns), c_dp, c_dp, C.c_void_p]),
    "pm_column_kernel_shape": (C.c_int, [C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pm_column_kernel_name": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_char_p, C.c_size_t]),
    "pm_thermwind_update": (C.c_int, [C.POINTER(pm_thermwind), C.c_int32, C.c_void_p]),
    "pm_thermwind_residuals": (C.c_int, [C.c_int32] + [c_dp] * 6 + [C.c_void_p]),
    "pm_psi_so_update": (C.c_int, [C.POINTER(pm_psi_so), C.c_int32, C.c_void_p]),
    "pm_so_ml_step": (C.c_int, [C.POINTER(pm_so_ml), C.c_double, C.c_void_p]),
    "pm_jn2018_bc_switch": (C.c_int, [C.POINTER(pm_jn2018_bc), C.c_void_p]),
    "pm_jn2018_steps": (C.c_int, [C.POINTER(pm_jn2018), C.c_double, C.c_int32, C.c_void_p]),
    "pm_twocol_run": (C.c_int, [C.POINTER(pm_twocol_loop), C.c_void_p]),
    "pm_jn2018_run": (C.c_int, [C.POINTER(pm_jn2018_loop), C.c_void_p]),
    "pm_so_tw_update": (C.c_int, [C.POINTER(pm_psi_so), C.POINTER(pm_thermwind), C.c_int32,
                                  C.c_void_p]),
    "pm_run_lds_bytes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_size_t)]),
    "pm_twobasin_forcing": (C.c_int, [C.c_int32, C.c_int32] + [c_dp] * 9 + [C.c_void_p]),
    "pm_column_equi_pass": (C.c_int, [C.POINTER(pm_column_equi), C.c_void_p]),
    "pm_equi_column_scratch_doubles": (C.c_size_t, [C.c_int32]),
    "pm_equi_column_newton": (C.c_int, [C.POINTER(pm_equi_column), C.c_void_p]),
    "pm_axpby": (C.c_int, [C.c_size_t, C.c_double, c_dp, C.c_double, c_dp, c_dp, C.c_void_p]),
    "pm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pm_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "pm_comm_destroy": (C.c_int, [C.c_void_p]),
    "pm_comm_allgather": (C.c_int, [C.c_void_p, c_dp, c_dp, C.c_size_t, C.c_void_p]),
    "pm_comm_gather_root": (C.c_int, [C.c_void_p, c_dp, c_dp, C.c_size_t, C.c_int32, C.c_int32,
                                      C.c_void_p]),
    "pm_comm_allreduce_max": (C.c_int, [C.c_void_p, c_dp, c_dp, C.c_size_t, C.c_void_p]),
    "pm_comm_barrier": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pm_selftest_fastdiv": (C.c_int, [C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pm_selftest_lane_shift": (C.c_int, [C.POINTER(C.c_int32)]),
    "pm_div3_proven": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "pm_recip_check": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "pm_selftest_div3": (C.c_int, [C.c_uint64, C.c_int32, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "pm_selftest_so_scans": (C.c_int, [C.c_int32, C.c_uint64, C.POINTER(C.c_double),
                                       C.POINTER(C.c_int32)]),
}

for _name, (_res, _args) in SIGNATURES.items():
  _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch
  _fn.restype = _res
  _fn.argtypes = _args


def check(rc):
  if rc != PM_OK:
    raise PmError(rc, lib.pm_last_error().decode("utf-8", "replace"))


_device_ready = False


def require_device(device=None):
  """Select the HIP device once; raises PmError(PM_ENODEV) when none is visible."""
  global _device_ready
  if _device_ready and device is None:
    return
  n = C.c_int(0)
  check(lib.pm_device_count(C.byref(n)))
  if n.value <= 0:
    raise PmError(PM_ENODEV, "no HIP device visible; pymoc_amd has no CPU fallback")
  if device is None:
    device = int(os.environ.get("LOCAL_RANK", "0")) % n.value
  check(lib.pm_set_device(int(device)))
  _device_ready = True


def device_info():
  require_device()
  name = C.create_string_buffer(256)
  cus, mem, clk = C.c_int(0), C.c_size_t(0), C.c_int(0)
  check(lib.pm_device_info(name, 256, C.byref(cus), C.byref(mem), C.byref(clk)))
  return {"name": name.value.decode(), "compute_units": cus.value,
          "hbm_bytes": mem.value, "clock_mhz": clk.value}
