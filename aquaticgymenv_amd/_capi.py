"""ctypes binding of libaqua_hip.so (include/aqua_hip.h).  No fallback: if the HIP library is
missing or does not load, importing this module raises -- the product has no CPU path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AQUA_HIP_LIB selects a tuning build of the same library (aquaticgymenv_amd/build.py --variants)
LIB_PATH = os.environ.get("AQUA_HIP_LIB") or os.path.join(_HERE, "lib", "libaqua_hip.so")

ABI_VERSION = 8
ACT_U8, ACT_I32, ACT_I64, ACT_F32X2, ACT_SAMPLE_D, ACT_SAMPLE_C, ACT_BEARING = range(7)
TERM_NONE, TERM_COLLIDED, TERM_TIME, TERM_SUCCESS = range(4)
MAX_OBSTACLES = 64

# every symbol include/aqua_hip.h declares (tests/test_capi_cpu.py checks the library exports them all)
SYMBOLS = (
    "aqua_version", "aqua_last_error", "aqua_obstacle_blob_bytes", "aqua_pack_obstacles", "aqua_step_f32",
    "aqua_reset_f32", "aqua_rollout_f32", "aqua_rollout_fused_f32", "aqua_tick_advance", "aqua_graph_begin",
    "aqua_graph_end", "aqua_graph_launch", "aqua_graph_upload", "aqua_graph_destroy",
    "aqua_discrete_constants", "aqua_obs_norm_f32",
    "aqua_ring_write_f32", "aqua_ring_write_u8", "aqua_pack_tables", "aqua_tables32_floats", "aqua_step_tables_f32", "aqua_reset_tables_f32",
    "aqua_event_create", "aqua_event_record", "aqua_event_elapsed_ms", "aqua_event_destroy", "aqua_graph_end_timed", "aqua_rollout_tables_f32",
    "aqua_rollout_tables_fused_f32",
    "aqua_ipc_buffer_create", "aqua_ipc_buffer_ptr", "aqua_ipc_buffer_handle", "aqua_ipc_buffer_destroy", "aqua_ipc_open",
    "aqua_ipc_close", "aqua_copy_async", "aqua_copy_fanout_async", "aqua_rollout_events_f32",
)
IPC_HANDLE_BYTES = 64
COPY_ENGINE_WAVES, COPY_ENGINE_DMA = 0, 1


class AquaParams(ctypes.Structure):
    _fields_ = [("waves", ctypes.c_int32), ("continuous", ctypes.c_int32), ("random_boat", ctypes.c_int32),
                ("random_goal", ctypes.c_int32), ("time_limit", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]


class AquaError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libaqua_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`python -m aquaticgymenv_amd.build` (needs hipcc); there is no CPU fallback")
    # torch ships its own libamdhip64 (soname libamdhip64.so.7, requested as "libamdhip64.so"); loading it
    # FIRST makes the dynamic loader satisfy our NEEDED libamdhip64.so.7 with that same runtime.  In the
    # other order two HIP runtimes end up in the process and the second one finds no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, u64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int
    pp = ctypes.POINTER(AquaParams)
    lib.aqua_version.restype = ci
    lib.aqua_last_error.restype = ctypes.c_char_p
    lib.aqua_obstacle_blob_bytes.argtypes = [ci]
    lib.aqua_obstacle_blob_bytes.restype = ctypes.c_size_t
    lib.aqua_pack_obstacles.argtypes = [vp, ci, vp, ctypes.c_size_t]
    lib.aqua_step_f32.argtypes = [pp, vp, ci, i64, i64, vp, i64, vp, vp, ci, i64, vp, i64, u64, u64, vp, vp, vp, vp,
                                  vp, ci, vp]
    lib.aqua_reset_f32.argtypes = [pp, vp, ci, i64, i64, vp, i64, vp, vp, u64, u64, vp, vp]
    lib.aqua_rollout_f32.argtypes = [pp, vp, ci, i64, i64, vp, i64, vp, i64, vp, ci, i64, i64, u64, u64, vp, vp, vp,
                                     i64, vp, i64, vp, ci, ci, vp]
    lib.aqua_rollout_events_f32.argtypes = [pp, vp, ci, i64, i64, vp, i64, vp, i64, vp, ci, i64, i64, u64, u64, vp, vp, vp,
                                            i64, vp, i64, vp, ci, ci, vp, vp, vp]
    lib.aqua_rollout_fused_f32.argtypes = [pp, vp, ci, i64, i64, vp, i64, vp, i64, vp, ci, i64, i64, u64, u64, vp,
                                           vp, vp, i64, ci, vp]
    lib.aqua_tick_advance.argtypes = [vp, u64, vp]
    lib.aqua_obs_norm_f32.argtypes = [vp, i64, i64, vp, vp, vp]
    lib.aqua_ring_write_f32.argtypes = [vp, i64, i64, i64, vp, i64, ci, i64, vp]
    lib.aqua_ring_write_u8.argtypes = [vp, i64, i64, i64, vp, i64, ci, i64, vp]
    lib.aqua_pack_tables.argtypes = [vp, ci, i64, i64, vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.aqua_tables32_floats.argtypes = [ci, i64]
    lib.aqua_tables32_floats.restype = ctypes.c_size_t
    lib.aqua_step_tables_f32.argtypes = [pp, vp, vp, ci, i64, ctypes.c_float, i64, i64, vp, i64, vp, vp, ci, i64, vp, i64,
                                         u64, u64, vp, vp, vp, vp, vp, ci, vp]
    lib.aqua_rollout_tables_f32.argtypes = [pp, vp, vp, ci, i64, ctypes.c_float, i64, i64, vp, i64, vp, i64, vp, ci, i64, i64,
                                            u64, u64, vp, vp, vp, i64, vp, i64, vp, ci, ci, vp]
    lib.aqua_rollout_tables_fused_f32.argtypes = [pp, vp, vp, ci, i64, ctypes.c_float, i64, i64, vp, i64, vp, i64, vp, ci, i64, i64,
                                                  u64, u64, vp, vp, vp, i64, ci, vp]
    lib.aqua_reset_tables_f32.argtypes = [pp, vp, ci, i64, i64, i64, vp, i64, vp, vp, u64, u64, vp, vp]
    lib.aqua_graph_begin.argtypes = [vp]
    lib.aqua_graph_end.argtypes = [vp, ctypes.POINTER(vp)]
    lib.aqua_graph_launch.argtypes = [vp, vp]
    lib.aqua_graph_upload.argtypes = [vp, vp]
    lib.aqua_graph_destroy.argtypes = [vp]
    lib.aqua_event_create.argtypes = [ctypes.POINTER(vp)]
    lib.aqua_event_record.argtypes = [vp, vp]
    lib.aqua_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.aqua_event_destroy.argtypes = [vp]
    lib.aqua_graph_end_timed.argtypes = [vp, ctypes.POINTER(vp), vp, vp]
    lib.aqua_ipc_buffer_create.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.aqua_ipc_buffer_ptr.argtypes = [vp]
    lib.aqua_ipc_buffer_ptr.restype = vp
    lib.aqua_ipc_buffer_handle.argtypes = [vp, ctypes.c_char_p]
    lib.aqua_ipc_buffer_destroy.argtypes = [vp]
    lib.aqua_ipc_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    lib.aqua_ipc_close.argtypes = [vp]
    lib.aqua_copy_async.argtypes = [vp, vp, ctypes.c_size_t, ci, vp]
    lib.aqua_copy_fanout_async.argtypes = [ctypes.POINTER(vp), ci, vp, ctypes.c_size_t, vp]
    lib.aqua_discrete_constants.argtypes = [ctypes.POINTER(ctypes.c_float)]
    lib.aqua_discrete_constants.restype = None
    for name in ("aqua_pack_obstacles", "aqua_step_f32", "aqua_reset_f32", "aqua_rollout_f32",
                 "aqua_rollout_fused_f32", "aqua_tick_advance", "aqua_graph_begin", "aqua_graph_end",
                 "aqua_graph_launch", "aqua_graph_upload", "aqua_graph_destroy",
                 "aqua_obs_norm_f32", "aqua_ring_write_f32", "aqua_ring_write_u8", "aqua_pack_tables", "aqua_tables32_floats",
                 "aqua_step_tables_f32", "aqua_reset_tables_f32", "aqua_event_create", "aqua_event_record",
                 "aqua_event_elapsed_ms", "aqua_event_destroy", "aqua_graph_end_timed", "aqua_rollout_tables_f32",
                 "aqua_rollout_tables_fused_f32", "aqua_ipc_buffer_create", "aqua_ipc_buffer_handle", "aqua_ipc_buffer_destroy",
                 "aqua_ipc_open", "aqua_ipc_close", "aqua_copy_async", "aqua_copy_fanout_async",
                 "aqua_rollout_events_f32"):
        getattr(lib, name).restype = ci
    if lib.aqua_version() != ABI_VERSION:
        raise ImportError("libaqua_hip.so ABI %d != binding %d: rebuild" % (lib.aqua_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        msg = lib.aqua_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        raise AquaError("%s failed (code %d): %s" % (what, rc, msg))


def pack_obstacles(rows):
    """rows: numpy float64 [K][5] -> bytes of the device-format blob (empty for K == 0)."""
    import numpy as np
    rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 5)
    k = rows.shape[0]
    if k > MAX_OBSTACLES:
        raise ValueError("at most %d obstacles, got %d" % (MAX_OBSTACLES, k))
    n = lib.aqua_obstacle_blob_bytes(k)
    buf = (ctypes.c_uint8 * max(n, 1))()
    check(lib.aqua_pack_obstacles(rows.ctypes.data, k, ctypes.addressof(buf), n), "aqua_pack_obstacles")
    return bytes(buf)[:n]


def discrete_constants():
    out = (ctypes.c_float * 9)()
    lib.aqua_discrete_constants(out)
    return [float(v) for v in out]
