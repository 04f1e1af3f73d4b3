"""Batched, device-resident image codec: the form the throughput metric is measured on.

The reference codes one image per call (spiht_wrapper.encode_image / decode_image).  Here B images of one
geometry are transformed, quantised and coded in one queue of HIP kernels; pixels, coefficient arrays and
bitstreams stay in HBM between the stages.  Device buffers may come from this module (hipMalloc through the
C ABI) or from anyone else (e.g. a torch tensor's data_ptr()) -- the C ABI takes plain pointers.
"""
import ctypes as C

import numpy as np

from . import _lib, color_models
from .spiht_wrapper import EncodingResult, SpihtSettings, _geometry, _mults_arg, _wavelet_mode_ids


class DeviceArray:
    """A hipMalloc'd buffer with a shape and dtype (no arithmetic: storage only)."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = ctx.alloc(max(self.nbytes, 4))

    def upload(self, arr, offset_bytes=0):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert offset_bytes + arr.nbytes <= self.nbytes
        self.ctx.upload(self.ptr + offset_bytes, arr)

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            self.ctx.download(out, self.ptr)
        return out

    def zero(self):
        self.ctx.memset(self.ptr, 0, self.nbytes)

    def free(self):
        if self.ptr:
            self.ctx.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BatchCodec:
    """encode/decode B images [B,c,H,W] (float64) at a fixed bit budget.

    settings / level / max_bits have the meaning of spiht_wrapper.encode_image, settings.color_model included: pixels
    go in and come out as RGB, the change to and from the coded colour model happens inside level 1 of the transforms
    (color_models.fused; 3-channel float64 images)."""

    def __init__(self, c, H, W, settings=None, level=None, max_bits=None, ctx=None, pixel_dtype=np.float64):
        # pixel_dtype float32: the encoder side runs PyWavelets' single-precision arithmetic (what the reference does
        # with float32 / float16 pixels); decoded images are float64 either way, as in the reference
        self.pixel_dtype = np.dtype(np.float32 if np.dtype(pixel_dtype) in (np.float32, np.float16) else np.float64)
        self.settings = settings if settings is not None else SpihtSettings()
        self.c, self.H, self.W, self.level = int(c), int(H), int(W), level
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.L = _lib.lib()
        self.wid, self.mid = _wavelet_mode_ids(self.settings)
        self.geom = _geometry(H, W, self.wid, level, self.mid)
        self.max_bits = 99999999999999999 if max_bits is None else int(max_bits)
        self.mults, self._mults_p = _mults_arg(self.settings.per_channel_quant_scales, self.c)
        bound = C.c_uint64()
        _lib.check(self.L.spiht_encode_bound(self.c, self.geom["enc_h"], self.geom["enc_w"], self.geom["ll_h"],
                                             self.geom["ll_w"], 0x3FFFFFFF, self.max_bits, C.byref(bound)))
        self.slot_stride = max(int(bound.value), 4)
        self._lv = -1 if level is None else int(level)
        if self.settings.color_model not in (None, "RGB"):
            if self.settings.color_model not in color_models.SUPPORTED_MODELS:
                raise ValueError(f'{self.settings.color_model} is not a supported color model. '
                                 f'Supported models are {color_models.SUPPORTED_MODELS}')
            if self.c != 3:
                raise ValueError("colour conversion needs 3 channels")
            if self.pixel_dtype != np.float64:
                raise ValueError("colour conversion on the device takes float64 pixels")

    def _color(self):
        return color_models.fused(self.ctx, self.settings.color_model)

    # ---- raw device-pointer API (ints) -------------------------------------------------------
    def encode_device(self, d_img, B, d_out, d_nbits, d_max_n, d_coeffs=None):
        fn = self.L.spiht_encode_image_batch_f32 if self.pixel_dtype == np.float32 else self.L.spiht_encode_image_batch_f64
        with self._color():
            _lib.check(fn(
                self.ctx.handle, C.c_void_p(d_img), int(B), self.c, self.H, self.W, self.wid, self.mid, self._lv,
                float(self.settings.quantization_scale), self._mults_p, self.max_bits, C.c_void_p(d_out),
                self.slot_stride, C.c_void_p(d_nbits), C.c_void_p(d_max_n), C.c_void_p(d_coeffs) if d_coeffs else None))

    def decode_device(self, d_data, d_nbytes, d_max_n, B, d_img_out, d_rec=None, slot_stride=None):
        with self._color():
            _lib.check(self.L.spiht_decode_image_batch_f64(
                self.ctx.handle, C.c_void_p(d_data), self.slot_stride if slot_stride is None else int(slot_stride),
                C.c_void_p(d_nbytes), C.c_void_p(d_max_n), int(B), self.c, self.H, self.W, self.wid, self.mid, self._lv,
                float(self.settings.quantization_scale), self._mults_p, C.c_void_p(d_img_out),
                C.c_void_p(d_rec) if d_rec else None))

    def nbits_to_nbytes(self, d_nbits, B, d_nbytes):
        _lib.check(self.L.spiht_nbits_to_nbytes(self.ctx.handle, C.c_void_p(d_nbits), int(B), C.c_void_p(d_nbytes)))

    # ---- host convenience ---------------------------------------------------------------------
    def encode(self, images):
        """images: float array [B,c,H,W] -> list of EncodingResult"""
        images = np.ascontiguousarray(images, dtype=self.pixel_dtype)
        B = images.shape[0]
        assert images.shape[1:] == (self.c, self.H, self.W)
        ctx = self.ctx
        d_img = DeviceArray(ctx, images.shape, self.pixel_dtype)
        d_out = DeviceArray(ctx, (B, self.slot_stride), np.uint8)
        d_nbits = DeviceArray(ctx, (B,), np.uint64)
        d_maxn = DeviceArray(ctx, (B,), np.uint8)
        try:
            d_img.upload(images)
            self.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)
            ctx.synchronize()
            out, nbits, maxn = d_out.download(), d_nbits.download(), d_maxn.download()
        finally:
            for d in (d_img, d_out, d_nbits, d_maxn):
                d.free()
        return [EncodingResult(out[b, :(int(nbits[b]) + 7) // 8].tobytes(), self.H, self.W, self.c, int(maxn[b]),
                               self.level) for b in range(B)]

    def decode(self, results):
        """list of EncodingResult (same geometry) -> float64 [B,c,H',W']"""
        B = len(results)
        stride = max(4, (max(len(r.encoded_bytes) for r in results) + 3) & ~3)
        data = np.zeros((B, stride), dtype=np.uint8)
        for b, r in enumerate(results):
            data[b, :len(r.encoded_bytes)] = np.frombuffer(r.encoded_bytes, np.uint8)
        nbytes = np.array([len(r.encoded_bytes) for r in results], dtype=np.uint64)
        maxn = np.array([r.max_n for r in results], dtype=np.uint8)
        ctx = self.ctx
        d_data = DeviceArray(ctx, data.shape, np.uint8)
        d_nbytes = DeviceArray(ctx, (B,), np.uint64)
        d_maxn = DeviceArray(ctx, (B,), np.uint8)
        d_img = DeviceArray(ctx, (B, self.c, self.geom["rec_h"], self.geom["rec_w"]), np.float64)
        try:
            d_data.upload(data)
            d_nbytes.upload(nbytes)
            d_maxn.upload(maxn)
            self.decode_device(d_data.ptr, d_nbytes.ptr, d_maxn.ptr, B, d_img.ptr, slot_stride=stride)
            ctx.synchronize()
            return d_img.download()
        finally:
            for d in (d_data, d_nbytes, d_maxn, d_img):
                d.free()

    def decode_prefixes(self, result, byte_lengths, one_walk=True):
        """Progressive decoding (the pattern of the reference's make_gif.py:46-61): the pictures of the
        prefixes `result.encoded_bytes[:k]` for every k in byte_lengths -> float64 [K,c,H',W'] (in the order given).
        one_walk (default): the stream is walked ONCE, to the longest prefix, and every tree node replays its operations
        into the K coefficient arrays (spiht_decode_budgets_dev_i32); then one batched inverse transform.
        one_walk=False: K streams in one batch, every prefix decoded by its own workgroup (K walks on K CUs)."""
        lens = [int(k) for k in byte_lengths]
        if not one_walk or not lens:
            pre = [EncodingResult(result.encoded_bytes[:k], result.h, result.w, result.c, result.max_n, result.level) for k in lens]
            return self.decode(pre)
        g = self.geom
        K = len(lens)
        order = np.argsort(np.asarray(lens), kind="stable")
        total = len(result.encoded_bytes)
        bud = np.ascontiguousarray([8 * min(lens[i], total) for i in order], dtype=np.uint64)
        data = np.frombuffer(result.encoded_bytes[:max(min(k, total) for k in lens)], dtype=np.uint8)
        d_rec = DeviceArray(self.ctx, (K, self.c, g["enc_h"], g["enc_w"]), np.int32)
        d_img = DeviceArray(self.ctx, (K, self.c, g["rec_h"], g["rec_w"]), np.float64)
        try:
            _lib.check(self.L.spiht_decode_budgets_dev_i32(
                self.ctx.handle, C.c_void_p(data.ctypes.data if data.size else 0), data.size, int(result.max_n), self.c,
                g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], C.c_void_p(bud.ctypes.data), K, C.c_void_p(d_rec.ptr)))
            with self._color():
                _lib.check(self.L.spiht_dequant_idwt_batch_f64(
                    self.ctx.handle, C.c_void_p(d_rec.ptr), K, self.c, self.H, self.W, self.wid, self.mid, self._lv,
                    float(self.settings.quantization_scale), self._mults_p, C.c_void_p(d_img.ptr)))
            self.ctx.synchronize()
            out = d_img.download()
        finally:
            d_rec.free()
            d_img.free()
        inv = np.empty(K, dtype=np.int64)
        inv[order] = np.arange(K)
        return out[inv]


class Pipeline:
    """The pipelined round trip as the C ABI offers it (include/spiht_hip.h: spiht_pipeline_*, csrc/pipeline.cpp): consecutive
    batches software-pipelined over three contexts -- the HBM-bound passes (transform + pyramid of step i+1, inverse transform
    of step i-1) on one, the list coding of step i on the two others in turn --, queued by the library itself: what a caller
    in any host language gets, and what bench.py times.  `codec` gives geometry and settings; its context runs the HBM-bound
    passes unless own_context is set."""

    def __init__(self, codec, B, own_context=False):
        """own_context: the HBM-bound passes on a context of the pipeline's own instead of the codec's (one more HIP stream;
        a process has few hardware queues for them)"""
        self.codec, self.B = codec, int(B)
        self.L = codec.L
        h = C.c_void_p()
        args = (codec.ctx.device, self.B, codec.c, codec.H, codec.W, codec.wid, codec.mid, codec._lv,
                float(codec.settings.quantization_scale), codec._mults_p, 0 if codec.max_bits >= 2 ** 63 else codec.max_bits, C.byref(h))
        if own_context:
            _lib.check(self.L.spiht_pipeline_create(*args))
        else:
            _lib.check(self.L.spiht_pipeline_create_on(codec.ctx.handle, *args))
        self.handle = h
        ss = C.c_uint64()
        _lib.check(self.L.spiht_pipeline_info(self.handle, C.byref(ss), None, None))
        assert ss.value == codec.slot_stride or codec.max_bits >= 2 ** 63, (ss.value, codec.slot_stride)
        self.slot_stride = int(ss.value)
        cm = codec.settings.color_model
        if cm not in (None, "RGB"):
            Af, Mf, pf = color_models._params("RGB", cm)
            Ai, Mi, pi = color_models._params(cm, "RGB")
            vp = C.c_void_p
            _lib.check(self.L.spiht_pipeline_set_color3(self.handle, vp(Af.ctypes.data), vp(Mf.ctypes.data), pf, vp(Ai.ctypes.data),
                                                        vp(Mi.ctypes.data), pi))

    def contexts(self):
        """[H, L0, L1] as borrowed Context objects (stage timing)"""
        hs = [C.c_void_p() for _ in range(3)]
        _lib.check(self.L.spiht_pipeline_contexts(self.handle, *[C.byref(x) for x in hs]))
        return [_lib.Context.borrowed(x, self.codec.ctx.device) for x in hs]

    def submit(self, d_img, d_out, d_nbits, d_max_n, d_img_out, comm=None, gathered=None, rank=0):
        """queue one step (device pointers as ints).  comm + gathered = (d_all_slots, d_all_nbits, d_all_max_n): the streams are
        all-gathered between encoder and decoder and the decoder reads this rank's rows of the gathered buffers"""
        vp = C.c_void_p
        if comm is None:
            _lib.check(self.L.spiht_pipeline_submit(self.handle, vp(d_img), vp(d_out), vp(d_nbits), vp(d_max_n), vp(d_img_out)))
        else:
            _lib.check(self.L.spiht_pipeline_submit_gather(self.handle, vp(d_img), vp(d_out), vp(d_nbits), vp(d_max_n), vp(d_img_out),
                                                           comm.handle, vp(gathered[0]), vp(gathered[1]), vp(gathered[2]), int(rank)))

    def flush(self):
        _lib.check(self.L.spiht_pipeline_flush(self.handle))

    def synchronize(self):
        _lib.check(self.L.spiht_pipeline_synchronize(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.L.spiht_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
