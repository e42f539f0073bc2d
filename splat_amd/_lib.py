"""ctypes binding of libsplat_hip.so (include/splat_hip.h).  Fails loudly if the library is
missing: there is no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPLAT_AMD_LIB") or os.path.join(_HERE, "libsplat_hip.so")   # override: A/B builds

SPLAT_OK, ERR_INVALID, ERR_HIP, ERR_NO_SCENE, ERR_CAPACITY = 0, -1, -2, -3, -4
TILE = 16


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("mode", C.c_int32), ("y_up", C.c_int32), ("sample_half", C.c_int32),
                ("zclip", C.c_int32), ("zmin", C.c_float), ("zmax", C.c_float), ("pair_capacity", C.c_uint64)]


class CameraC(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("w", C.c_float), ("h", C.c_float),
                ("htanx", C.c_float), ("htany", C.c_float), ("focal", C.c_float), ("cam_pos", C.c_float * 3),
                ("lowpass", C.c_float), ("sh_dim", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_gaussians", C.c_uint64), ("n_visible", C.c_uint64), ("n_singular", C.c_uint64),
                ("n_pairs", C.c_uint64), ("max_tile_len", C.c_uint64), ("bytes_algorithmic", C.c_uint64),
                ("ms_preprocess", C.c_float), ("ms_scan", C.c_float), ("ms_emit", C.c_float),
                ("ms_sort", C.c_float), ("ms_composite", C.c_float), ("ms_total", C.c_float),
                ("n_fallback", C.c_uint64), ("n_sort_fallback", C.c_uint64), ("n_iter_scan", C.c_uint64), ("n_iter_blend", C.c_uint64),
                ("n_blocks_culled", C.c_uint64), ("flops_algorithmic", C.c_uint64),
                ("n_near_tiles", C.c_uint64), ("n_near_fallback", C.c_uint64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Record(C.Structure):
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("hx", C.c_float), ("hy", C.c_float),
                ("conic_a", C.c_float), ("conic_b", C.c_float), ("conic_c", C.c_float), ("opacity", C.c_float),
                ("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("depth", C.c_float),
                ("px0", C.c_int32), ("px1", C.c_int32), ("py0", C.c_int32), ("py1", C.c_int32)]


# every symbol include/splat_hip.h declares: (name, restype, argtypes)
_fp = C.POINTER(C.c_float)
SYMBOLS = [
    ("splat_abi_version", C.c_uint32, []),
    ("splat_stats_size", C.c_uint64, []),
    ("splat_default_config", None, [C.POINTER(Config)]),
    ("splat_create", C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    ("splat_destroy", None, [C.c_void_p]),
    ("splat_last_error", C.c_char_p, [C.c_void_p]),
    ("splat_upload_scene", C.c_int, [C.c_void_p, C.c_uint64, _fp, _fp, _fp, _fp]),
    ("splat_compute_cov3d", C.c_int, [C.c_void_p, C.c_uint64, _fp, _fp, _fp]),
    ("splat_set_slab", C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ("splat_tile_row_loads", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.POINTER(C.c_uint64), C.c_int32]),
    ("splat_render", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.POINTER(C.c_uint32), C.POINTER(Stats)]),
    ("splat_render_frame", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.POINTER(C.c_uint32), C.POINTER(Stats)]),
    ("splat_render_device", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    ("splat_render_frame_device", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    ("splat_sync", C.c_int, [C.c_void_p]),
    ("splat_frames_dropped", C.c_uint64, [C.c_void_p]),
    ("splat_device_bytes", C.c_uint64, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("splat_set_frame_overlap", C.c_int, [C.c_void_p, C.c_int32]),
    ("splat_set_option", C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    ("splat_get_option", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    ("splat_stream", C.c_void_p, [C.c_void_p]),
    ("splat_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("splat_get_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int32]),
    ("splat_get_records", C.c_int, [C.c_void_p, C.POINTER(Record), C.c_uint64]),
    ("splat_get_tile_lists", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32),
                                        C.c_uint64]),
    ("splat_binning_mode", C.c_int64, [C.c_void_p]),
    ("splat_render_stream", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.c_void_p]),
    ("splat_stream_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("splat_device_alloc", C.c_void_p, [C.c_void_p, C.c_uint64]),
    ("splat_device_free", None, [C.c_void_p, C.c_void_p]),
    ("splat_device_upload", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("splat_device_download", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("splat_host_alloc", C.c_void_p, [C.c_uint64]),
    ("splat_host_free", None, [C.c_void_p]),
    ("splat_host_register", C.c_int, [C.c_void_p, C.c_uint64]),
    ("splat_host_unregister", C.c_int, [C.c_void_p]),
    # multi-GPU
    ("splat_slab_partition", C.c_int, [C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_int32)]),
    ("splat_comm_unique_id", C.c_int, [C.POINTER(C.c_uint8)]),
    ("splat_comm_init_rank", C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
    ("splat_comm_set_slabs", C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    ("splat_comm_gather", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    ("splat_comm_loopback", C.c_int, [C.c_void_p, C.c_int32]),
    ("splat_comm_destroy", None, [C.c_void_p]),
    ("splat_multi_create", C.c_int, [C.POINTER(Config), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    ("splat_multi_destroy", None, [C.c_void_p]),
    ("splat_multi_last_error", C.c_char_p, [C.c_void_p]),
    ("splat_multi_upload_scene", C.c_int, [C.c_void_p, C.c_uint64, _fp, _fp, _fp, _fp]),
    ("splat_multi_balance", C.c_int, [C.c_void_p, C.POINTER(CameraC)]),
    ("splat_multi_set_frame_overlap", C.c_int, [C.c_void_p, C.c_int32]),
    ("splat_multi_get_slabs", C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    ("splat_multi_render", C.c_int, [C.c_void_p, C.POINTER(CameraC), C.POINTER(C.c_uint32), C.POINTER(Stats)]),
    ("splat_multi_render_frame", C.c_int, [C.c_void_p, C.POINTER(CameraC)]),
    ("splat_multi_sync", C.c_int, [C.c_void_p]),
    ("splat_multi_image", C.c_void_p, [C.c_void_p]),
    ("splat_multi_download", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int32, C.c_int32]),
    ("splat_multi_ctx", C.c_void_p, [C.c_void_p, C.c_int32]),
]
UNIQUE_ID_BYTES = 128
# SPLAT_OPT_* (include/splat_hip.h): tuning options of a context
OPT_PIPELINE_DEPTH = 1
OPT_FUSED_SORT_MAX = 2
OPT_REGION_SPARE = 3
OPT_EARLY_OUT_EPS = 4
OPT_EARLY_OUT_MIN_LIST = 5
OPT_EARLY_OUT_SCAN_EIGHTHS = 6
OPT_SORT_IN_COMPOSITOR = 7
OPT_PAIR_WALK = 8
OPT_TIMING_EVERY = 9
OPT_BLOCK_CULLING = 10
OPT_ONE_PASS_BINNING = 11
OPT_KEY_BUFFER_BYTES = 12
OPT_FAST_CLOSE_WIDTH = 13
OPT_PRIORITY_LIST_LEN = 14
OPT_FRAME_OVERLAP = 15
OPT_NEAR_SELECT_KEYS = 16
OPT_OVERFLOW_REDO = 17
OPT_START_HINTS = 18
OPT_HOST_ZERO_COPY = 19
OPT_KEYS_PER_GAUSSIAN = 20
OPT_COUNT_FIRST = 21
OPT_LARGE_SPLAT_TILES = 22
OPT_LARGE_LIST_MIN = 23
ABI_VERSION = 6          # SPLAT_ABI_VERSION of the header these structures were written against

_LIB = None


def lib():
    """Load libsplat_hip.so.  Raises if it has not been built (__graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "splat_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C splat_amd/csrc). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # a library built from another header would write a splat_stats of another size into ours
        if L.splat_abi_version() != ABI_VERSION or L.splat_stats_size() != C.sizeof(Stats):
            raise RuntimeError("splat_amd: %s speaks ABI version %d (splat_stats of %d bytes), this binding %d (%d bytes): rebuild"
                               % (LIB_PATH, L.splat_abi_version(), L.splat_stats_size(), ABI_VERSION, C.sizeof(Stats)))
        _LIB = L
    return _LIB
