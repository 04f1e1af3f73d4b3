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


class OverlappedCodec:
    """Round trips a sequence of batches with the two kinds of work on different contexts.

    The transform / pyramid / inverse-transform passes are HBM-bound; the list coder is latency-bound and leaves the
    HBM idle.  Context H runs the former, contexts L0 / L1 (alternating by batch) the latter, ordered with events only
    (the host never blocks), so while a batch is list-coded, H already transforms the next one and inverse-transforms
    the previous one (`pair="inverse"`, the default):

        H:  A(i)               [X(i-1) done] I(i-1)   A(i+1)                 [X(i) done] I(i) ...
        L:  [A(i), I(i-2) done] U(i-2) E(i) X(i)              [A(i+1), I(i-1) done] U(i-1) E(i+1) X(i+1) ...

    A = DWT + quantise + pyramid, E / X = encoder / decoder list kernels, I = dequantise + inverse DWT, U = put the
    zeros back into the coefficient array X(i-2) scattered into (spiht_unscatter_lists_batch_i32: through the
    decoder's lists, which is why each of the two arrays has its own list-coding context) -- a full zero-fill per
    batch would add 6.6 GB of writes to the HBM-bound side.  E(i+1) is ordered after X(i), so list kernels never run
    beside one another; HBM-bound kernels never overlap one another either (one in-order stream).  Results are
    bit-identical to BatchCodec's fused calls (same kernels).  Coefficient arrays, pyramid and decoder output are
    double-buffered.

    Resident decoder workgroups (96 VGPRs a wavefront) leave an HBM-bound kernel two or three instead of six or seven
    workgroups per CU: whichever transform shares the GPU with the decoder loses.  The inverse transform therefore runs
    as persistent workgroups that fetch a tile ahead (k_idwt_level_pf), and the list-coding contexts use the
    8-wavefront build of the decoder (`decoder_waves=8`: a longer walk, a lighter neighbour): 17.6-18.0 ms per step at
    256 x 1080p with 12 wavefronts 18.3-18.7.  Variants measured there, none better than the default:
      pair="forward"    X(i) waits for I(i-1), so it runs beside A(i+1) and I(i-1) meets only the encoder kernel
      split_inverse     the coarse levels of I(i) (level .. 2) behind X(i) on the list-coding stream (the encoder
                        kernel of the next batch then waits for room on CUs full of transform workgroups)
      l_priority        high stream priority for the list-coding contexts: no change
      e_first           encoder kernel queued before the unscatter: worse
    and one that is the default since the inverse transform got faster:
      u_early           the unscatter of a batch right behind its inverse transform, on that batch's list-coding
                        context (it then runs beside the next forward transform, which loses 2 % to its scattered
                        writes; queued in front of the next encoder kernel it met the coarse inverse levels, and a
                        0.02 ms launch of those took 0.47 ms): 17.5-17.7 against 17.7-18.5 ms
    `between` (optional callable) runs between E(i) and X(i) with that batch's L context: the hook for the stream
    gather of a multi-GPU job; `dec_src` makes the decoder read the gathered buffers."""

    def __init__(self, codec, B, ctx_l=None, split_inverse=False, pair="inverse", l_priority=0, e_first=False, u_early=True,
                 decoder_waves=8, l1_flags=True, coarse_first=False):
        self.codec, self.B = codec, int(B)
        if pair not in ("forward", "inverse"):
            raise ValueError("pair must be 'forward' or 'inverse'")
        self.pair = pair
        self.H = codec.ctx
        self.Ls = [ctx_l if ctx_l is not None else _lib.Context(self.H.device, l_priority),
                   _lib.Context(self.H.device, l_priority)]
        # decoder workgroups of 8 instead of 12 wavefronts: a longer walk (11.4 instead of 8.9 ms per 256 1080p streams
        # in this schedule), but the transforms beside it lose less, and they are the longer queue
        for cx in self.Ls:
            cx.set_decoder_waves(decoder_waves)
            cx.set_option("wide_encode", 0)  # (one workgroup per image whatever the batch size: see csrc/pipeline.cpp)
        self.e_first = bool(e_first)  # experiment: encoder kernel queued before the unscatter
        self.u_early = bool(u_early)  # the unscatter of a batch right behind its inverse transform (off the next list-coding chain)
        self._unscattered = [True, True]
        self.L = self.Ls[0]
        g = codec.geom
        n = codec.c * g["enc_h"] * g["enc_w"]
        self.n = n
        mk = lambda shape, dt: [DeviceArray(self.H, shape, dt) for _ in range(2)]  # noqa: E731
        self.coeffs, self.rec = mk((B, n), np.int32), mk((B, n), np.int32)
        self.dmsb, self.lmsb = mk((B, n), np.uint8), mk((B, n), np.uint8)
        self.maxabs = mk((B,), np.uint32)
        ah, aw = C.c_int64(), C.c_int64()
        _lib.check(codec.L.spiht_idwt_approx_shape(codec.H, codec.W, codec.wid, codec._lv, C.byref(ah), C.byref(aw)))
        self.split = ah.value > 0 and split_inverse  # two levels or more: coarse levels on the list-coding stream
        # coarse_first: the coarse levels of I(i-1) between the transform and the pyramid of A(i) on H -- there the list-coding
        # streams are idle (X(i-1) has ended, E(i) waits for the pyramid), so they run alone instead of beside the encoder
        self.cf = ah.value > 0 and coarse_first and not self.split and pair == "inverse"
        self.approx = mk((B, codec.c, ah.value, aw.value), np.float64) if (self.split or self.cf) else [None, None]
        self._coarse_done = [False, False]
        # occupancy words of the inverse transform's level-1 tiles: decoder -> inverse transform (include/spiht_hip.h)
        nw = C.c_uint64()
        _lib.check(codec.L.spiht_l1_flags_words(codec.c, codec.H, codec.W, codec.wid, codec.mid, codec._lv, C.byref(nw)))
        self.flags = mk((B, nw.value), np.uint32) if nw.value and l1_flags else [None, None]
        # the coefficient arrays are this object's own and only the forward transform writes them: their zero padding is
        # written once per array, not once per step (beside a list decoder that launch of thin strips took 0.74 ms)
        self.H.set_option("pads_persist", 1)
        for r in self.rec:  # zero once; from then on U keeps them zero
            self.H.memset(r.ptr, 0, r.nbytes)
        self.H.synchronize()
        self.ev_a = [_lib.Event(self.H) for _ in range(2)]    # A(i) done
        self.ev_d = [_lib.Event(self.Ls[s]) for s in range(2)]  # X(i) done
        self.ev_i = [_lib.Event(self.H) for _ in range(2)]    # I(i) done
        self.used = [False, False]
        self.i = 0
        self._pending = None  # (slot, d_img_out) of the batch whose inverse transform has not been queued yet

    def contexts(self):
        return [self.H] + self.Ls

    def close(self):
        """give the arrays back (the promise about their padding ends with them)"""
        if getattr(self, "coeffs", None):
            try:
                self.synchronize()
            finally:
                for d in self.coeffs + self.rec + self.dmsb + self.lmsb + self.maxabs + [f for f in self.flags if f] + \
                        [a for a in self.approx if a]:
                    d.free()  # (spiht_dev_free also drops what the context remembers about the array)
                self.coeffs = None
                self.H.set_option("pads_persist", 0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _idwt(self, s, d_img_out):
        cd = self.codec
        self.H.wait_event(self.ev_d[s])
        with cd._color():  # (the colour setting of H is put on around each transform call: H may serve other codecs too)
            if self.cf:
                self._coarse(s)
                self._coarse_done[s] = False
                _lib.check(cd.L.spiht_idwt_level1_flags_batch_f64(
                    self.H.handle, C.c_void_p(self.rec[s].ptr), C.c_void_p(self.approx[s].ptr),
                    C.c_void_p(self.flags[s].ptr if self.flags[s] else None), self.B, cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv,
                    float(cd.settings.quantization_scale), cd._mults_p, C.c_void_p(d_img_out)))
            elif self.split:
                _lib.check(cd.L.spiht_idwt_level1_batch_f64(
                    self.H.handle, C.c_void_p(self.rec[s].ptr), C.c_void_p(self.approx[s].ptr), self.B, cd.c, cd.H, cd.W,
                    cd.wid, cd.mid, cd._lv, float(cd.settings.quantization_scale), cd._mults_p, C.c_void_p(d_img_out)))
            else:
                _lib.check(cd.L.spiht_dequant_idwt_flags_batch_f64(
                    self.H.handle, C.c_void_p(self.rec[s].ptr), C.c_void_p(self.flags[s].ptr if self.flags[s] else None), self.B,
                    cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv, float(cd.settings.quantization_scale), cd._mults_p,
                    C.c_void_p(d_img_out)))
        self.H.record(self.ev_i[s])

    def _coarse(self, s):
        """coarse_first: levels level .. 2 of batch s's inverse transform on H (once)"""
        cd = self.codec
        if self._coarse_done[s]:
            return
        self.H.wait_event(self.ev_d[s])
        _lib.check(cd.L.spiht_idwt_coarse_batch_f64(
            self.H.handle, C.c_void_p(self.rec[s].ptr), self.B, cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv,
            float(cd.settings.quantization_scale), cd._mults_p, C.c_void_p(self.approx[s].ptr)))
        self._coarse_done[s] = True

    def submit(self, d_img, d_out, d_nbits, d_max_n, d_nbytes, d_img_out, between=None, dec_src=None):
        """queue the round trip of one batch (device pointers as in BatchCodec.encode_device / decode_device).
        dec_src = (d_slots, d_nbits, d_max_n): what the decoder reads instead of the encoder's own outputs -- e.g. this
        rank's rows of the buffers `between` gathered the streams into."""
        cd, B, g = self.codec, self.B, self.codec.geom
        s = self.i & 1
        Lc = self.Ls[s]
        vp = C.c_void_p
        q = float(cd.settings.quantization_scale)
        # H: front half of the encoder
        with cd._color():
            if self.cf:
                _lib.check(cd.L.spiht_dwt_pyramid_batch_f64(
                    self.H.handle, vp(d_img), B, cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv, q, cd._mults_p,
                    vp(self.coeffs[s].ptr), None, None, vp(self.maxabs[s].ptr)))
                if self._pending is not None:
                    self._coarse(self._pending[0])
                _lib.check(cd.L.spiht_pyramid_batch_i32(
                    self.H.handle, vp(self.coeffs[s].ptr), B, cd.c, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"],
                    vp(self.dmsb[s].ptr), vp(self.lmsb[s].ptr), None))
            else:
                _lib.check(cd.L.spiht_dwt_pyramid_batch_f64(
                    self.H.handle, vp(d_img), B, cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv, q, cd._mults_p,
                    vp(self.coeffs[s].ptr), vp(self.dmsb[s].ptr), vp(self.lmsb[s].ptr), vp(self.maxabs[s].ptr)))
        self.H.record(self.ev_a[s])
        # L: zeros back into the array batch i-2 was decoded into, once its inverse transform (queued on H by the
        # previous submit) has read it ...
        # (queued behind A(i): right after I(i-2) the forward DWT of this batch starts on H, and the scattered writes
        # would take HBM bandwidth from it; here they run beside the next inverse transform instead)
        if self.used[s ^ 1]:
            Lc.wait_event(self.ev_d[s ^ 1])
        Lc.wait_event(self.ev_a[s])
        def unscatter():
            if self.used[s] and not self._unscattered[s]:
                Lc.wait_event(self.ev_i[s])
                _lib.check(cd.L.spiht_unscatter_lists_batch_i32(Lc.handle, vp(self.rec[s].ptr), B, cd.c, g["enc_h"], g["enc_w"]))
                self._unscattered[s] = True
        if not self.e_first:
            unscatter()
        # ... and list coding, after the previous batch's decoder on the other context
        _lib.check(cd.L.spiht_encode_lists_batch_i32(
            Lc.handle, vp(self.coeffs[s].ptr), vp(self.dmsb[s].ptr), vp(self.lmsb[s].ptr), vp(self.maxabs[s].ptr), B,
            cd.c, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], cd.max_bits, vp(d_out), cd.slot_stride, vp(d_nbits),
            vp(d_max_n)))
        if self.e_first:
            unscatter()
        if between is not None:
            between(Lc)
        if self.pair == "forward" and self._pending is not None:
            # the previous batch's inverse transform goes on H now (behind A(i)), and this batch's decoder waits for it
            self._idwt(*self._pending)
            Lc.wait_event(self.ev_i[self._pending[0]])
            self._pending = None
        x_out, x_nbits, x_max_n = dec_src if dec_src is not None else (d_out, d_nbits, d_max_n)
        _lib.check(cd.L.spiht_nbits_to_nbytes(Lc.handle, vp(x_nbits), B, vp(d_nbytes)))
        _lib.check(cd.L.spiht_decode_lists_flags_batch_i32(
            Lc.handle, vp(x_out), cd.slot_stride, vp(d_nbytes), vp(x_max_n), B, cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv,
            vp(self.rec[s].ptr), vp(self.flags[s].ptr if self.flags[s] else None)))
        if self.split:  # the coarse levels of this batch's inverse transform, behind its decoder
            _lib.check(cd.L.spiht_idwt_coarse_batch_f64(
                Lc.handle, vp(self.rec[s].ptr), B, cd.c, cd.H, cd.W, cd.wid, cd.mid, cd._lv, q, cd._mults_p,
                vp(self.approx[s].ptr)))
        Lc.record(self.ev_d[s])
        self.used[s] = True
        self._unscattered[s] = False
        # H: back half of the previous batch's decoder (pair="inverse": beside this batch's decoder)
        if self._pending is not None:
            self._idwt(*self._pending)
            if self.u_early:  # ... and the zeros back into its array as soon as that has read it, on ITS list-coding context
                sp = self._pending[0]
                Lp = self.Ls[sp]
                Lp.wait_event(self.ev_i[sp])
                _lib.check(cd.L.spiht_unscatter_lists_batch_i32(Lp.handle, vp(self.rec[sp].ptr), B, cd.c, g["enc_h"], g["enc_w"]))
                self._unscattered[sp] = True
        self._pending = (s, d_img_out)
        self.i += 1

    def flush(self):
        """queue the inverse transform of the last submitted batch (call before synchronising)"""
        if self._pending is not None:
            self._idwt(*self._pending)
            self._pending = None

    def synchronize(self):
        self.flush()
        for Lc in self.Ls:
            Lc.synchronize()
        self.H.synchronize()


class Pipeline:
    """The pipelined round trip as the C ABI offers it (include/spiht_hip.h: spiht_pipeline_*, csrc/pipeline.cpp): the schedule
    of OverlappedCodec's defaults, queued by the library itself on three contexts it owns -- what a caller in any host
    language gets, and what bench.py times.  `codec` gives geometry and settings (its context is not used)."""

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
