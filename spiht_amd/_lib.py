"""ctypes binding of libspiht_hip.so (include/spiht_hip.h).

The HIP library is the only compute path of this package: if it is missing or cannot be loaded the
import fails loudly -- there is no CPU fallback (the CPU restatement under oracle/ is test
infrastructure and is never imported from here).
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPIHT_HIP_LIB") or os.path.join(_HERE, "libspiht_hip.so")  # override: diagnostic builds

OK, ERR_LL, ERR_EMPTY, ERR_SHAPE, ERR_CAPACITY, ERR_HIP, ERR_ARG, ERR_MAGNITUDE, ERR_INTERNAL, ERR_TOO_LARGE, \
    ERR_NOMEM = range(11)

MODES = {"reflect": 0, "symmetric": 1, "periodic": 2, "zero": 3, "constant": 4}


class SpihtHipError(RuntimeError):
    """A HIP runtime failure or an internal guard of libspiht_hip."""


class PanicException(BaseException):
    """Counterpart of pyo3_runtime.PanicException, which the reference raises when the Rust core panics
    (assert!(ll_h > 1), encoder_decoder.rs:160-161; out-of-bounds index; unwrap on an empty array).
    Like PyO3's, it derives from BaseException."""


_lib = None
_lib_lock = threading.Lock()

# every symbol include/spiht_hip.h declares
SYMBOLS = [
    "spiht_strerror", "spiht_last_hip_error", "spiht_abi_version", "spiht_ctx_create", "spiht_ctx_create_priority", "spiht_ctx_resident_ticket", "spiht_ctx_wait_resident",
    "spiht_ctx_destroy",
    "spiht_ctx_synchronize", "spiht_ctx_wait_on", "spiht_event_create", "spiht_event_destroy", "spiht_event_record",
    "spiht_ctx_wait_event", "spiht_ctx_stream", "spiht_dwt_pyramid_batch_f64", "spiht_encode_lists_batch_i32",
    "spiht_decode_lists_batch_i32", "spiht_unscatter_lists_batch_i32", "spiht_ctx_set_timing", "spiht_ctx_reset_timing", "spiht_ctx_num_stages",
    "spiht_ctx_stage_name", "spiht_ctx_get_timing", "spiht_encode_i32", "spiht_encode_bound", "spiht_decode_i32",
    "spiht_decode_with_metadata_i32", "spiht_encode_batch_i32", "spiht_decode_batch_i32", "spiht_wavelet_id", "spiht_mode_id", "spiht_geometry",
    "spiht_encode_image_batch_f64", "spiht_decode_image_batch_f64", "spiht_dwt_quant_batch_f64",
    "spiht_encode_image_batch_f32", "spiht_dwt_quant_batch_f32",
    "spiht_dequant_idwt_batch_f64", "spiht_pyramid_batch_i32", "spiht_color3_batch_f64", "spiht_ctx_set_color3", "spiht_ctx_get_color3", "spiht_ctx_set_decoder_waves", "spiht_decode_budgets_i32", "spiht_decode_budgets_dev_i32", "spiht_nbits_to_nbytes", "spiht_dev_alloc", "spiht_dev_free", "spiht_host_alloc", "spiht_host_free",
    "spiht_dev_upload", "spiht_dev_download", "spiht_dev_memset", "spiht_dev_copy",
    "spiht_idwt_coarse_batch_f64", "spiht_idwt_level1_batch_f64", "spiht_idwt_level1_flags_batch_f64", "spiht_idwt_approx_shape",
    "spiht_encode_image_host_f64", "spiht_encode_image_host_f32", "spiht_decode_image_host_f64",
    "spiht_dequant_idwt_host_f64",
    "spiht_comm_unique_id", "spiht_comm_create", "spiht_comm_destroy", "spiht_comm_info", "spiht_gather_streams", "spiht_gather_row_offsets",
    "spiht_comm_barrier", "spiht_comm_allreduce_max_f64", "spiht_rccl_library", "spiht_ctx_lock", "spiht_ctx_unlock",
    "spiht_pipeline_create", "spiht_pipeline_create_on", "spiht_pipeline_destroy", "spiht_pipeline_info", "spiht_pipeline_set_color3", "spiht_pipeline_submit",
    "spiht_pipeline_submit_gather", "spiht_pipeline_flush", "spiht_pipeline_synchronize", "spiht_pipeline_contexts",
    "spiht_geometry_mode", "spiht_wavelet_taps", "spiht_ctx_set_option", "spiht_ctx_get_option", "spiht_ctx_wide_stats", "spiht_l1_flags_words", "spiht_decode_lists_flags_batch_i32", "spiht_dequant_idwt_flags_batch_f64",
]


def lib():
    """Load libspiht_hip.so (once).  Raises ImportError if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libspiht_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C spiht_amd/csrc`; spiht_amd has no CPU fallback" % LIB_PATH)
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise ImportError("cannot load %s: %s" % (LIB_PATH, e))
        i64, u64, u8, i32 = C.c_int64, C.c_uint64, C.c_uint8, C.c_int
        vp = C.c_void_p
        L.spiht_strerror.restype = C.c_char_p
        L.spiht_strerror.argtypes = [i32]
        L.spiht_last_hip_error.restype = C.c_char_p
        L.spiht_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.spiht_ctx_create_priority.argtypes = [i32, i32, C.POINTER(vp)]
        L.spiht_ctx_resident_ticket.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32)]
        L.spiht_ctx_wait_resident.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
        L.spiht_ctx_destroy.argtypes = [vp]
        L.spiht_ctx_destroy.restype = None
        L.spiht_ctx_synchronize.argtypes = [vp]
        L.spiht_ctx_wait_on.argtypes = [vp, vp]
        L.spiht_event_create.argtypes = [vp, C.POINTER(vp)]
        L.spiht_event_destroy.argtypes = [vp]
        L.spiht_event_destroy.restype = None
        L.spiht_event_record.argtypes = [vp, vp]
        L.spiht_ctx_wait_event.argtypes = [vp, vp]
        L.spiht_ctx_stream.argtypes = [vp, C.POINTER(vp)]
        L.spiht_dwt_pyramid_batch_f64.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp, vp, vp, vp]
        L.spiht_encode_lists_batch_i32.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, u64, vp, u64, vp, vp]
        L.spiht_decode_lists_batch_i32.argtypes = [vp, vp, u64, vp, vp, i64, i64, i64, i64, i64, i64, vp]
        L.spiht_unscatter_lists_batch_i32.argtypes = [vp, vp, i64, i64, i64, i64]
        L.spiht_color3_batch_f64.argtypes = [vp, vp, vp, i64, i64, vp, vp, C.c_double]
        L.spiht_ctx_set_color3.argtypes = [vp, vp, vp, C.c_double, vp, vp, C.c_double]
        L.spiht_ctx_get_color3.argtypes = [vp, C.POINTER(i32), vp, vp, C.POINTER(C.c_double), vp, vp, C.POINTER(C.c_double)]
        L.spiht_ctx_set_decoder_waves.argtypes = [vp, C.c_int]
        L.spiht_decode_budgets_i32.argtypes = [vp, vp, u64, C.c_uint8, i64, i64, i64, i64, i64, vp, i64, vp]
        L.spiht_decode_budgets_dev_i32.argtypes = [vp, vp, u64, C.c_uint8, i64, i64, i64, i64, i64, vp, i64, vp]
        L.spiht_ctx_set_timing.argtypes = [vp, i32]
        L.spiht_ctx_reset_timing.argtypes = [vp]
        L.spiht_ctx_stage_name.restype = C.c_char_p
        L.spiht_ctx_stage_name.argtypes = [i32]
        L.spiht_ctx_get_timing.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)]
        L.spiht_encode_i32.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, u64, vp, u64, C.POINTER(u64),
                                       C.POINTER(u8)]
        L.spiht_encode_bound.argtypes = [i64, i64, i64, i64, i64, C.c_uint32, u64, C.POINTER(u64)]
        L.spiht_decode_i32.argtypes = [vp, vp, u64, u8, i64, i64, i64, i64, i64, vp]
        L.spiht_decode_with_metadata_i32.argtypes = [vp, vp, u64, u8, i64, i64, i64, i64, i64, vp, vp, i64, vp, vp]
        L.spiht_encode_batch_i32.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, u64, vp, u64, vp, vp]
        L.spiht_decode_batch_i32.argtypes = [vp, vp, u64, vp, vp, i64, i64, i64, i64, i64, i64, vp]
        L.spiht_wavelet_id.argtypes = [C.c_char_p]
        L.spiht_mode_id.argtypes = [C.c_char_p]
        L.spiht_wavelet_taps.argtypes = [i32]
        L.spiht_geometry.argtypes = [i64, i64, i32, i32, C.POINTER(i32)] + [C.POINTER(i64)] * 6
        L.spiht_encode_image_batch_f64.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, u64, vp,
                                                   u64, vp, vp, vp]
        L.spiht_decode_image_batch_f64.argtypes = [vp, vp, u64, vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double,
                                                   vp, vp, vp]
        L.spiht_dwt_quant_batch_f64.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_dwt_quant_batch_f32.argtypes = L.spiht_dwt_quant_batch_f64.argtypes
        L.spiht_encode_image_batch_f32.argtypes = L.spiht_encode_image_batch_f64.argtypes
        L.spiht_dequant_idwt_batch_f64.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_pyramid_batch_i32.argtypes = [vp, vp, i64, i64, i64, i64, i64, i64, vp, vp, vp]
        L.spiht_nbits_to_nbytes.argtypes = [vp, vp, i64, vp]
        L.spiht_dev_alloc.argtypes = [vp, u64, C.POINTER(vp)]
        L.spiht_dev_free.argtypes = [vp, vp]
        L.spiht_dev_upload.argtypes = [vp, vp, vp, u64]
        L.spiht_dev_download.argtypes = [vp, vp, vp, u64]
        L.spiht_dev_memset.argtypes = [vp, vp, i32, u64]
        L.spiht_dev_copy.argtypes = [vp, vp, vp, u64]
        L.spiht_idwt_coarse_batch_f64.argtypes = [vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_idwt_level1_batch_f64.argtypes = [vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_idwt_level1_flags_batch_f64.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_idwt_approx_shape.argtypes = [i64, i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
        L.spiht_encode_image_host_f64.argtypes = [vp, vp, i64, i64, i64, i32, i32, i32, C.c_double, vp, u64, vp, u64,
                                                  C.POINTER(u64), C.POINTER(u8)]
        L.spiht_encode_image_host_f32.argtypes = L.spiht_encode_image_host_f64.argtypes
        L.spiht_decode_image_host_f64.argtypes = [vp, vp, u64, u8, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_dequant_idwt_host_f64.argtypes = [vp, vp, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_comm_unique_id.argtypes = [vp]
        L.spiht_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.spiht_comm_destroy.argtypes = [vp]
        L.spiht_comm_destroy.restype = None
        L.spiht_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.spiht_gather_streams.argtypes = [vp, vp, vp, vp, vp, i64, u64, vp, vp, vp]
        L.spiht_gather_row_offsets.argtypes = [i32, i32, i64, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.spiht_comm_barrier.argtypes = [vp, vp]
        L.spiht_comm_allreduce_max_f64.argtypes = [vp, vp, C.POINTER(C.c_double)]
        L.spiht_rccl_library.restype = C.c_char_p
        L.spiht_rccl_library.argtypes = []
        L.spiht_host_alloc.argtypes = [u64, C.POINTER(vp)]
        L.spiht_host_free.argtypes = [vp]
        L.spiht_ctx_set_option.argtypes = [vp, C.c_char_p, i64]
        L.spiht_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
        L.spiht_ctx_wide_stats.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.spiht_l1_flags_words.argtypes = [i64, i64, i64, i32, i32, i32, C.POINTER(u64)]
        L.spiht_decode_lists_flags_batch_i32.argtypes = [vp, vp, u64, vp, vp, i64, i64, i64, i64, i32, i32, i32, vp, vp]
        L.spiht_geometry_mode.argtypes = [i64, i64, i32, i32, i32, C.POINTER(i32)] + [C.POINTER(i64)] * 6
        L.spiht_dequant_idwt_flags_batch_f64.argtypes = [vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, vp]
        L.spiht_pipeline_create.argtypes = [i32, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, u64, C.POINTER(vp)]
        L.spiht_pipeline_create_on.argtypes = [vp, i32, i64, i64, i64, i64, i32, i32, i32, C.c_double, vp, u64, C.POINTER(vp)]
        L.spiht_pipeline_destroy.argtypes = [vp]
        L.spiht_pipeline_destroy.restype = None
        L.spiht_pipeline_info.argtypes = [vp, C.POINTER(u64), C.POINTER(i64), C.POINTER(i64)]
        L.spiht_pipeline_set_color3.argtypes = [vp, vp, vp, C.c_double, vp, vp, C.c_double]
        L.spiht_pipeline_submit.argtypes = [vp, vp, vp, vp, vp, vp]
        L.spiht_pipeline_submit_gather.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32]
        L.spiht_pipeline_flush.argtypes = [vp]
        L.spiht_pipeline_synchronize.argtypes = [vp]
        L.spiht_pipeline_contexts.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.spiht_ctx_lock.argtypes = [vp]
        L.spiht_ctx_unlock.argtypes = [vp]
        _lib = L
        return _lib


def check(status):
    """Map a C status to the exception the reference would raise."""
    if status == OK:
        return
    L = lib()
    msg = L.spiht_strerror(status).decode()
    if status in (ERR_LL, ERR_EMPTY, ERR_SHAPE):
        raise PanicException(msg)
    if status == ERR_HIP:
        raise SpihtHipError("%s: %s" % (msg, L.spiht_last_hip_error().decode()))
    if status in (ERR_INTERNAL, ERR_NOMEM):
        raise SpihtHipError("%s (%s)" % (msg, L.spiht_last_hip_error().decode()))
    if status == ERR_TOO_LARGE:
        raise OverflowError(msg)
    raise ValueError(msg)


class Context:
    """One spiht_ctx (device id, stream, scratch).  Created lazily, one per device."""

    def __init__(self, device=0, priority=0):
        self._lib = lib()
        h = C.c_void_p()
        check(self._lib.spiht_ctx_create_priority(int(device), int(priority), C.byref(h)))
        self.handle = h
        self.device = int(device)

    @classmethod
    def borrowed(cls, handle, device=0):
        """a view of a context someone else owns (e.g. a pipeline's): never destroyed from here"""
        self = cls.__new__(cls)
        self._lib = lib()
        self.handle = handle if isinstance(handle, C.c_void_p) else C.c_void_p(handle)
        self.device = int(device)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "handle", None):
            if not getattr(self, "_borrowed", False):
                self._lib.spiht_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lock(self):
        """hold the context's (recursive) mutex across several calls of this thread: other threads' calls on the context
        wait until unlock() -- for settings that are state of the context (color_models.fused)"""
        check(self._lib.spiht_ctx_lock(self.handle))

    def unlock(self):
        check(self._lib.spiht_ctx_unlock(self.handle))

    def set_option(self, name, value):
        """a switch of the library (spiht_ctx_set_option): "l1_flags", "pads_persist", "wide_encode", "wide_groups",
        "wide_solo", "idwt_groups"; results do not depend on them"""
        check(self._lib.spiht_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64()
        check(self._lib.spiht_ctx_get_option(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def wide_stats(self):
        """(images, images that fell back to the single-workgroup encoder) of the last several-CUs-per-image encode call"""
        g, u = C.c_uint32(), C.c_uint32()
        check(self._lib.spiht_ctx_wide_stats(self.handle, C.byref(g), C.byref(u)))
        return int(g.value), int(u.value)

    def set_decoder_waves(self, waves):
        """wavefronts per decoder workgroup on this context: 12 (default, fastest alone) or 8 (lighter beside HBM-bound
        kernels of other contexts)"""
        check(self._lib.spiht_ctx_set_decoder_waves(self.handle, int(waves)))

    def wait_on(self, other):
        """device-side ordering: work queued on self after this call waits for work queued on `other` before it"""
        check(self._lib.spiht_ctx_wait_on(self.handle, other.handle))

    def synchronize(self):
        check(self._lib.spiht_ctx_synchronize(self.handle))

    def stream_ptr(self):
        """the context's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)"""
        p = C.c_void_p()
        check(self._lib.spiht_ctx_stream(self.handle, C.byref(p)))
        return p.value or 0

    def record(self, event=None):
        """mark the point this context's queue has reached; returns the Event (a new one unless given)"""
        if event is None:
            event = Event(self)
        check(self._lib.spiht_event_record(event.handle, self.handle))
        return event

    def wait_event(self, event):
        """work queued on this context from now on starts only after the recorded point"""
        check(self._lib.spiht_ctx_wait_event(self.handle, event.handle))

    # stage timing (HIP events on the context's stream)
    def set_timing(self, on):
        check(self._lib.spiht_ctx_set_timing(self.handle, 1 if on else 0))

    def reset_timing(self):
        check(self._lib.spiht_ctx_reset_timing(self.handle))

    def timing(self):
        out = {}
        for s in range(self._lib.spiht_ctx_num_stages()):
            ms, n = C.c_double(), C.c_uint64()
            check(self._lib.spiht_ctx_get_timing(self.handle, s, C.byref(ms), C.byref(n)))
            out[self._lib.spiht_ctx_stage_name(s).decode()] = (ms.value, n.value)
        return out

    # device memory helpers
    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self._lib.spiht_dev_alloc(self.handle, int(nbytes), C.byref(p)))
        return p.value

    def free(self, ptr):
        check(self._lib.spiht_dev_free(self.handle, C.c_void_p(ptr)))

    def upload(self, dptr, arr):
        check(self._lib.spiht_dev_upload(self.handle, C.c_void_p(dptr), C.c_void_p(arr.ctypes.data), arr.nbytes))

    def download(self, arr, dptr):
        check(self._lib.spiht_dev_download(self.handle, C.c_void_p(arr.ctypes.data), C.c_void_p(dptr), arr.nbytes))

    def memset(self, dptr, value, nbytes):
        check(self._lib.spiht_dev_memset(self.handle, C.c_void_p(dptr), int(value), int(nbytes)))


class Event:
    """spiht_event: a point in one context's queue that other contexts can wait for (device-side)."""

    def __init__(self, ctx):
        self._lib = lib()
        h = C.c_void_p()
        check(self._lib.spiht_event_create(ctx.handle, C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.spiht_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_ctxs = {}
_ctx_lock = threading.Lock()


def _host_free(ptr):
    try:
        if _lib is not None:
            _lib.spiht_host_free(C.c_void_p(ptr))
    except Exception:  # interpreter shutdown
        pass


def result_array(shape, dtype):
    """An uninitialised array for a call to fill and RETURN (the reference's decode calls return new arrays,
    spiht_wrapper.py:192-216, lib.rs:35-42): backed by page-locked memory from the library's pool (spiht_host_alloc), so
    the device -> host copy is one DMA at the link's speed; the buffer goes back to the pool when the array -- and every
    view of it -- is gone.  Small arrays, or when no such memory is to be had: an ordinary numpy array."""
    import weakref
    import numpy as np
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < (1 << 20):
        return np.empty(shape, dtype)
    p = C.c_void_p()
    if lib().spiht_host_alloc(nbytes, C.byref(p)) != OK or not p.value:
        return np.empty(shape, dtype)
    buf = (C.c_char * nbytes).from_address(p.value)
    weakref.finalize(buf, _host_free, p.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("SPIHT_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _ctx_lock:
        c = _ctxs.get(device)
        if c is None or c.handle is None:
            c = Context(device)
            if os.environ.get("SPIHT_DECODER_WAVES"):  # test runs of the whole suite on the 8-wavefront decoder
                c.set_decoder_waves(int(os.environ["SPIHT_DECODER_WAVES"]))
            _ctxs[device] = c
        return c
