#!/usr/bin/env python3
"""bench.py -- throughput of the SPIHT image hot path on MI355X.

Metric (BASELINE.json): Mpixels/s encode+decode at fixed bpp (pixels = H*W per image, not x channels).
Workload: 1920x1080 RGB float64 images, bior2.2 / reflect / level 7 / q=50, 0.5 bpp (max_bits = 1 036 800):
BASELINE config 2's image, --batch of them per GPU (default 256 = config 4's per-GPU shard, weak scaling).
One step = every image of the batch goes pixels -> DWT -> quantise -> SPIHT stream -> SPIHT decode ->
dequantise -> inverse DWT -> pixels, all resident in HBM (inputs are uploaded before the timed region).
With N > 1 each rank codes its own shard and the streams are all-gathered (RCCL) between encode and decode.

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline      achieved vs peak HBM bandwidth of the dominant HBM-bound kernel (forward DWT, level 1),
                timed with HIP events on the library's own stream inside the timed region
  cpu_baseline  the CPU oracle (a port of the reference algorithm; the Rust reference cannot be built here)
                timed on a bounded sample of the same workload, one core
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, C_IMG = 1080, 1920, 3
LEVEL, BPP = 7, 0.5
WAVELET, MODE, QSCALE = "bior2.2", "reflect", 50.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def synth_image(seed, c, h, w):
    """SURVEY.md 8(d) pixel-domain generator (mimics spiht/utils.py imload: uint8/255 as float64)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((c, h, w))
    b = np.cumsum(np.cumsum(g, axis=1), axis=2)
    mn = b.min(axis=(1, 2), keepdims=True)
    mx = b.max(axis=(1, 2), keepdims=True)
    b = (b - mn) / (mx - mn)
    b = b + 0.02 * rng.standard_normal((c, h, w))
    return np.round(np.clip(b, 0, 1) * 255).astype(np.uint8) / 255


def _cpu_worker(job):
    """one oracle round trip in a worker process (cpu_baseline_all_cores); returns pixels coded"""
    seed, reps = job
    from oracle import oracle as O
    img = synth_image(seed, C_IMG, H, W)
    mb = int(H * W * BPP)
    for _ in range(reps):
        data, mn, _g = O.encode_image(img, WAVELET, MODE, LEVEL, QSCALE, None, mb)
        O.decode_image(data, mn, C_IMG, H, W, WAVELET, LEVEL, QSCALE, None)
    return reps * H * W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic images cycled through the batch")
    ap.add_argument("--cpu-sample", type=int, default=24,
                    help="images the CPU baseline codes, cycling through the distinct ones (0 = skip); 24 = about 11 s of one core")
    ap.add_argument("--cpu-cores", type=int, default=16,
                    help="processes of the all-cores CPU figure (one image each; 16 = the CPU share of a one-GPU box; <= 1: skip)")
    ap.add_argument("--pixels", choices=["float64", "float32"], default="float64",
                    help="pixel dtype.  float64 (default) is what the reference's loader produces and what the metric is quoted "
                         "on; float32 runs the single-precision forward transform PyWavelets would run on such pixels (half the "
                         "DWT read traffic); the decode side is float64 either way, as in the reference")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default): steps are software-pipelined -- the HBM-bound halves (DWT + pyramid of step i+1, "
                         "inverse DWT of step i-1) run on one context while step i is list-coded on another "
                         "(spiht_amd/batch.py:OverlappedCodec); all K steps complete inside the timed region.  Measured: "
                         "19.4-19.6 vs 22.6-22.9 ms/step.  The forward DWT starts when the decoder of the previous step "
                         "has finished, so it still has the GPU to itself (same time as in the serial schedule); the "
                         "inverse DWT and the decoder share it (x1.9 and x1.25 their own time).  0: every step runs its "
                         "stages back to back on one stream, each kernel with the whole GPU")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (library contexts) the batch is split over.  Measured on MI355X/ROCm 7.2: chunks on "
                         "separate streams did not overlap (2 streams = same time, 4 and 8 slower), so the default is 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    dist = torch = None
    # under torchrun (RANK set) the distributed path is taken even with one rank, so it can be rehearsed on one GPU
    if N > 1 or world > 1 or ("RANK" in os.environ and os.environ.get("SPIHT_BENCH_FORCE_DIST", "1") == "1"):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()

    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec, DeviceArray
    from spiht_amd.spiht_wrapper import SpihtSettings

    ctx = _lib.default_context(local_rank)
    B = args.batch
    max_bits = int(H * W * BPP)  # demonstrate.py:50
    K = max(1, min(args.streams, B))
    ctxs = [ctx] + [_lib.Context(local_rank) for _ in range(K - 1)]
    pix = np.dtype(args.pixels)
    codecs = [BatchCodec(C_IMG, H, W, SpihtSettings(WAVELET, QSCALE, MODE), LEVEL, max_bits, ctx=cx, pixel_dtype=pix) for cx in ctxs]
    codec = codecs[0]
    g = codec.geom
    slot = codec.slot_stride
    bounds = [(k * B // K, (k + 1) * B // K) for k in range(K)]  # chunk k of the batch runs on stream k

    # ---- synthetic inputs, resident in HBM before the timed region ----
    nd = max(1, min(args.distinct, B))
    base = [synth_image(1000 + rank * nd + i, C_IMG, H, W) for i in range(nd)]
    d_img = DeviceArray(ctx, (B, C_IMG, H, W), pix)
    per = C_IMG * H * W * pix.itemsize
    for b in range(B):
        d_img.upload(base[b % nd], offset_bytes=b * per)
    d_rec_img = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
    d_nbytes = DeviceArray(ctx, (B,), np.uint64)
    if dist is not None:
        # buffers the collective touches are torch tensors (RCCL needs them); the codec only sees their pointers
        out_t = torch.zeros((B, slot), dtype=torch.uint8, device="cuda")
        nbits_t = torch.zeros((B,), dtype=torch.int64, device="cuda")
        maxn_t = torch.zeros((B,), dtype=torch.uint8, device="cuda")
        gathered = torch.zeros((world * B, slot), dtype=torch.uint8, device="cuda")
        g_nbits = torch.zeros((world * B,), dtype=torch.int64, device="cuda")
        g_maxn = torch.zeros((world * B,), dtype=torch.uint8, device="cuda")
        out_ptr, nbits_ptr, maxn_ptr = out_t.data_ptr(), nbits_t.data_ptr(), maxn_t.data_ptr()
    else:
        d_out = DeviceArray(ctx, (B, slot), np.uint8)
        d_nbits = DeviceArray(ctx, (B,), np.uint64)
        d_maxn = DeviceArray(ctx, (B,), np.uint8)
        out_ptr, nbits_ptr, maxn_ptr = d_out.ptr, d_nbits.ptr, d_maxn.ptr

    img_b = C_IMG * H * W * pix.itemsize
    rec_b = C_IMG * g["rec_h"] * g["rec_w"] * 8

    def enc_chunk(k):
        a, b = bounds[k]
        codecs[k].encode_device(d_img.ptr + a * img_b, b - a, out_ptr + a * slot, nbits_ptr + a * 8, maxn_ptr + a)

    def dec_chunk(k):
        a, b = bounds[k]
        codecs[k].nbits_to_nbytes(nbits_ptr + a * 8, b - a, d_nbytes.ptr + a * 8)
        codecs[k].decode_device(out_ptr + a * slot, d_nbytes.ptr + a * 8, maxn_ptr + a, b - a, d_rec_img.ptr + a * rec_b)

    pipe = None
    if args.pipeline and K == 1 and pix == np.float64:
        # The HBM-bound halves (DWT+pyramid of step i+1, zero-fill, inverse DWT of step i-1) run on context H while
        # context L list-codes step i; ordered by events, the host never blocks (spiht_amd/batch.py:OverlappedCodec).
        from spiht_amd.batch import OverlappedCodec
        pipe = OverlappedCodec(codec, B)
        ctxs.extend(pipe.Ls)
        gather_hook = None
        if dist is not None:
            # the stream gather rides on the batch's list-coding stream between the encoder's and the
            # decoder's list kernels
            l_streams = {cx.handle.value: torch.cuda.ExternalStream(cx.stream_ptr(), device=torch.device("cuda", local_rank))
                         for cx in pipe.Ls}

            def gather_hook(ctx_l):
                with torch.cuda.stream(l_streams[ctx_l.handle.value]):
                    dist.all_gather_into_tensor(gathered, out_t)
                    dist.all_gather_into_tensor(g_nbits, nbits_t)
                    dist.all_gather_into_tensor(g_maxn, maxn_t)

    def step():
        if pipe is not None:
            pipe.submit(d_img.ptr, out_ptr, nbits_ptr, maxn_ptr, d_nbytes.ptr, d_rec_img.ptr, between=gather_hook)
            return
        # every call below only queues kernels on the chunk's own stream
        if dist is None:
            for k in range(K):
                enc_chunk(k)
                dec_chunk(k)
        else:
            for k in range(K):
                enc_chunk(k)
            # the one exchange of the path: fixed-size stream slots + bit counts + start planes,
            # rank-major (spiht_amd/dist.py: rank r owns rows [r*B, (r+1)*B))
            for cx in ctxs:
                cx.synchronize()
            dist.all_gather_into_tensor(gathered, out_t)
            dist.all_gather_into_tensor(g_nbits, nbits_t)
            dist.all_gather_into_tensor(g_maxn, maxn_t)
            torch.cuda.synchronize()
            for k in range(K):
                dec_chunk(k)

    def sync_all():
        if pipe is not None:
            pipe.flush()  # the inverse transform of the last step (inside the timed region)
        for cx in ctxs:
            cx.synchronize()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    for cx in ctxs:
        cx.reset_timing()
        cx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    stages = {}
    for cx in ctxs:
        cx.set_timing(False)
        for name, (ms, n) in cx.timing().items():
            o = stages.get(name, (0.0, 0))
            stages[name] = (o[0] + ms, o[1] + n)

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- correctness of what was timed (outside the timed region) ----
    nbits = np.empty(B, np.uint64)
    maxn = np.empty(B, np.uint8)
    ctx.download(nbits, nbits_ptr)
    ctx.download(maxn, maxn_ptr)
    gather_ok = None
    if dist is not None:
        # the gathered rows of this rank must be its own slots
        gather_ok = bool(torch.equal(gathered[rank * B:(rank + 1) * B], out_t) and
                         torch.equal(g_nbits[rank * B:(rank + 1) * B], nbits_t))
    rec0 = np.empty((C_IMG, g["rec_h"], g["rec_w"]), np.float64)
    ctx.download(rec0, d_rec_img.ptr)
    mae = float(np.abs(rec0[:, :H, :W] - base[0]).mean())

    result = None
    if rank == 0:
        total_images = world * B * args.steps
        mpix = total_images * H * W / dt / 1e6
        h1, w1 = (H + 6 - 1) // 2, (W + 6 - 1) // 2
        # algorithmic bytes of one forward-DWT level-1 launch over the whole batch:
        # read the float64 image once, write LL as float64 and the three detail bands as int32
        per_launch = bounds[0][1] - bounds[0][0]  # images one launch covers (chunk size)
        dwt_bytes = per_launch * C_IMG * (H * W * pix.itemsize + h1 * w1 * pix.itemsize + 3 * h1 * w1 * 4)
        ms_l1, n_l1 = stages.get("dwt_level1", (0.0, 0))
        avg_ms = ms_l1 / n_l1 if n_l1 else float("nan")
        achieved = dwt_bytes / (avg_ms * 1e-3) / 1e9 if n_l1 else float("nan")
        # HBM traffic of that kernel from PMC counters (tools/collect_traffic.py; separate rocprofv3 --pmc passes)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "dwt_l1_traffic.json")
        if os.path.exists(tpath) and pix == np.float64:
            try:
                traffic = round(json.load(open(tpath))["hbm_bytes_per_image"] * per_launch)
            except Exception:
                traffic = None
        # latency of ONE image (BASELINE config 2 as written: "single 1920x1080 RGB"), HBM-resident, same kernels
        t_enc, t_dec = [], []
        for _ in range(5):
            ctx.synchronize()
            t1 = time.perf_counter()
            codec.encode_device(d_img.ptr, 1, out_ptr, nbits_ptr, maxn_ptr)
            ctx.synchronize()
            t2 = time.perf_counter()
            codec.nbits_to_nbytes(nbits_ptr, 1, d_nbytes.ptr)
            codec.decode_device(out_ptr, d_nbytes.ptr, maxn_ptr, 1, d_rec_img.ptr)
            ctx.synchronize()
            t3 = time.perf_counter()
            t_enc.append((t2 - t1) * 1e3)
            t_dec.append((t3 - t2) * 1e3)
        single = {"encode_ms": round(sorted(t_enc)[2], 3), "decode_ms": round(sorted(t_dec)[2], 3)}

        # the same kernels with the GPU to themselves (one serial round trip of the batch after the timed region): in
        # the pipelined schedule the numbers above are those of kernels that share the GPU with the list coder
        alone = None
        if pipe is not None:
            enc_chunk(0)
            dec_chunk(0)
            ctx.synchronize()
            ctx.reset_timing()
            ctx.set_timing(True)
            for _ in range(3):
                enc_chunk(0)
                dec_chunk(0)
            ctx.synchronize()
            ctx.set_timing(False)
            tm = ctx.timing()
            a_ms = {k2: v2[0] / v2[1] for k2, v2 in tm.items() if v2[1] and k2 in ("dwt_level1", "idwt_level1")}
            if "dwt_level1" in a_ms:
                alone = {"dwt_level1_ms": round(a_ms["dwt_level1"], 4),
                         "dwt_level1_frac": round(dwt_bytes / (a_ms["dwt_level1"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "idwt_level1_ms": round(a_ms.get("idwt_level1", float("nan")), 4)}

        # the other HBM-bound passes north_star names, same definition (algorithmic bytes / stage time per launch group)
        def _gbs(stage, nbytes):
            ms, n = stages.get(stage, (0.0, 0))
            if not n or ms <= 0:
                return None
            per_group = ms / (args.steps * K)  # one launch group per chunk per step
            return {"achieved_GBps": round(nbytes / (per_group * 1e-3) / 1e9, 1),
                    "frac": round(nbytes / (per_group * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms_per_group": round(per_group, 4)}
        n_coef = C_IMG * g["enc_h"] * g["enc_w"]
        n_par = C_IMG * (g["enc_h"] // 2) * (g["enc_w"] // 2)
        other = {
            # inverse level 1: read 3 int32 bands + float64 LL, write the float64 image
            "idwt_level1": _gbs("idwt_level1", per_launch * C_IMG * (H * W * 8 + h1 * w1 * 8 + 3 * h1 * w1 * 4)),
            # significance pyramid: read 4 B per coefficient, write 1 B per parent (D) + 1 B per grand-parent (L)
            "pyramid": _gbs("pyramid", per_launch * (4 * n_coef + n_par + n_par // 4)),
        }
        if alone is not None and other.get("idwt_level1") is not None and alone["idwt_level1_ms"] == alone["idwt_level1_ms"]:
            # in the pipelined schedule the inverse transform shares the GPU with the list decoder (by design)
            nb_i = per_launch * C_IMG * (H * W * 8 + h1 * w1 * 8 + 3 * h1 * w1 * 4)
            other["idwt_level1"]["alone_ms"] = alone["idwt_level1_ms"]
            other["idwt_level1"]["alone_frac"] = round(nb_i / (alone["idwt_level1_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # PMC traffic of those two passes (tools/collect_traffic.py), per image, beside the algorithmic bytes
        opath = os.path.join(ROOT, "profiles", "hbm_traffic_other.json")
        if os.path.exists(opath) and pix == np.float64:
            try:
                ot = json.load(open(opath))
                for key, src in (("idwt_level1", "idwt_level1"), ("pyramid", "pyramid_rounds")):
                    if other.get(key) is not None:
                        other[key]["traffic_bytes_per_image"] = round(ot[src]["hbm_bytes_per_image"])
                        other[key]["algorithmic_bytes_per_image"] = ot[src]["algorithmic_bytes_per_image"]
            except Exception:
                pass
        result = {
            "metric": "Mpixels/sec encode+decode at fixed bpp; bitstream-exact vs Rust ref",
            "value": round(mpix, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f64" if pix == np.float64 else "f32 forward / f64 inverse") + " DWT / int32 bit-plane coding",
            "data": "synthetic",
            "config": {"workload": "cfg2 image (1920x1080 RGB, bior2.2 reflect level 7, q=50, 0.5 bpp) x %d per GPU "
                                   "(cfg4 shard), encode+decode, HBM-resident" % B,
                       "images_per_gpu": B, "streams": K, "images_per_launch": per_launch,
                       "schedule": ("steps software-pipelined: HBM-bound passes of steps i+1 / i-1 on one stream while step i is "
                                    "list-coded on another" if pipe is not None else "stages back to back"), "max_bits": max_bits, "images_per_s": round(total_images / dt, 2),
                       "coeff_array": [C_IMG, g["enc_h"], g["enc_w"]], "ll": [g["ll_h"], g["ll_w"]]},
            "roofline": {"bound": "hbm", "kernel": ("k_dwt_level<6>" if pix == np.float64 else "k_dwt_level_f32<6>") +
                         " (forward DWT level 1, fused quantise)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dwt_bytes, "avg_launch_ms": round(avg_ms, 4),
                         "kernel_alone": alone},
            "stages_ms_per_step_summed_over_streams": {k: round(v[0] / args.steps, 3) for k, v in stages.items() if v[1]},
            "roofline_other_hbm_passes": other,
            "single_image_latency": single,
            "check": {"nbits_all_equal_budget": bool((nbits == max_bits).all()), "max_n": int(maxn[0]),
                      "mean_abs_err_image0": round(mae, 5), "gather_rows_match": gather_ok},
        }

        # ---- CPU baseline: the oracle (port of the reference algorithm) on a bounded sample, one core ----
        result["cpu_baseline"] = None
        if args.cpu_sample > 0 and world == 1:  # the CPU leg runs at N = 1 only
            from oracle import oracle as O
            ns = args.cpu_sample
            tc = time.perf_counter()
            streams = []
            for i in range(ns):
                data, mn, _ = O.encode_image(base[i % nd].astype(pix), WAVELET, MODE, LEVEL, QSCALE, None, max_bits)
                streams.append((data, mn))
            t_enc = time.perf_counter() - tc
            tc = time.perf_counter()
            for i in range(ns):
                O.decode_image(streams[i][0], streams[i][1], C_IMG, H, W, WAVELET, LEVEL, QSCALE, None)
            t_dec = time.perf_counter() - tc
            result["cpu_baseline"] = {
                "value": round(ns * H * W / (t_enc + t_dec) / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                "kind": "port",
                "sample": "%d of the same 1080p images, encode %.2fs + decode %.2fs, oracle/liboracle.so (gcc -O3), "
                          "host has %d cores" % (ns, t_enc, t_dec, os.cpu_count() or 0)}
            # the same port, one independent image per core (the reference is single-threaded per image; SURVEY.md 8d ii)
            if args.cpu_cores > 1:
                import concurrent.futures as cf
                import multiprocessing as mp
                ncores = min(args.cpu_cores, os.cpu_count() or 1)
                jobs = [(1000 + i % nd, 2) for i in range(ncores)]
                with cf.ProcessPoolExecutor(max_workers=ncores, mp_context=mp.get_context("spawn")) as ex:
                    list(ex.map(_cpu_worker, [(1000, 1)] * ncores))  # start-up (imports, library load) outside the timing
                    tc = time.perf_counter()
                    npx = sum(ex.map(_cpu_worker, jobs))
                    t_all = time.perf_counter() - tc
                result["cpu_baseline_all_cores"] = {
                    "value": round(npx / t_all / 1e6, 3), "unit": "Mpixels/s", "cores": ncores, "kind": "port",
                    "sample": "%d processes x 2 round trips of a 1080p image in %.2fs" % (ncores, t_all)}
            # bit-exactness of the timed GPU output against the oracle, image 0 of rank 0
            out0 = np.empty(slot, np.uint8)
            ctx.download(out0, out_ptr)
            gpu_stream = out0[:(int(nbits[0]) + 7) // 8].tobytes()
            result["check"]["stream_bit_exact_vs_oracle"] = bool(gpu_stream == streams[0][0] and int(maxn[0]) == streams[0][1])
            ref_img = O.decode_image(streams[0][0], streams[0][1], C_IMG, H, W, WAVELET, LEVEL, QSCALE, None)
            result["check"]["decoded_image_bit_exact_vs_oracle"] = bool(np.array_equal(ref_img, rec0))
        print(json.dumps(result))
        sys.stdout.flush()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
