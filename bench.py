#!/usr/bin/env python3
"""bench.py -- throughput of the SPIHT image hot path on MI355X.

Metric (BASELINE.json): Mpixels/s encode+decode at fixed bpp (pixels = H*W per image, not x channels).
Workload: 1920x1080 RGB float64 images, bior2.2 / reflect / level 7 / q=50, 0.5 bpp (max_bits = 1 036 800):
BASELINE config 2's image, --batch of them per GPU (default 256 = config 4's per-GPU shard, weak scaling), every
image of the job distinct: image i of the job is synth_image(seed = 1000 + i), rank r owns images
[r*batch, (r+1)*batch).  One step = every image of the batch goes pixels -> DWT -> quantise -> SPIHT stream ->
(gather) -> SPIHT decode -> dequantise -> inverse DWT -> pixels, all resident in HBM (inputs are uploaded before the
timed region).

N > 1: one process per GPU.  `python bench.py --gpus N` from a bare shell starts the N rank processes itself (before
anything touches a GPU); under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
(python -m torch.distributed.run ...) each process is one rank.  Every rank codes its own shard; the streams are
all-gathered between encode and decode by the library itself (spiht_gather_streams: RCCL on the list-coding stream)
and each rank decodes its rows of the GATHERED buffer.  No torch anywhere.

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline      achieved vs peak HBM bandwidth of the dominant HBM-bound kernel (forward DWT, level 1),
                timed with HIP events on the library's own stream inside the timed region
  cpu_baseline  the CPU oracle (a port of the reference algorithm; the Rust reference cannot be built here)
                timed on a bounded sample of the same workload, one core
"""
import argparse
import json
import os
import subprocess
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
CACHE_DIR = os.environ.get("SPIHT_BENCH_CACHE", "/tmp/spiht_bench_cache")


def synth_image(seed, c, h, w):
    """SURVEY.md 8(d) pixel-domain generator (mimics spiht/utils.py imload: uint8/255 as float64)."""
    return synth_u8(seed, c, h, w) / 255


def synth_u8(seed, c, h, w):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((c, h, w))
    b = np.cumsum(np.cumsum(g, axis=1), axis=2)
    mn = b.min(axis=(1, 2), keepdims=True)
    mx = b.max(axis=(1, 2), keepdims=True)
    b = (b - mn) / (mx - mn)
    b = b + 0.02 * rng.standard_normal((c, h, w))
    return np.round(np.clip(b, 0, 1) * 255).astype(np.uint8)


def _cache_path(seed):
    return os.path.join(CACHE_DIR, "img_%d_%dx%dx%d.u8.npy" % (seed, C_IMG, H, W))


def _make_cached(seed):
    """worker: synthesise one image into the cache (uint8; the pixels are uint8/255, exactly what synth_image returns)"""
    p = _cache_path(seed)
    if not os.path.exists(p):
        os.makedirs(CACHE_DIR, exist_ok=True)
        tmp = "%s.%d.tmp.npy" % (p, os.getpid())
        np.save(tmp, synth_u8(seed, C_IMG, H, W))
        os.replace(tmp, p)
    return seed


def load_image(seed):
    p = _cache_path(seed)
    if os.path.exists(p):
        try:
            return np.load(p) / 255
        except Exception:
            pass
    return synth_image(seed, C_IMG, H, W)


def _nproc():
    """CPUs this process may use: its affinity mask, cut by the cgroup's CPU quota where there is one (what `nproc` and a
    container's share say) -- the all-cores CPU leg runs one independent image per such CPU."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def digest(a):
    """order-independent-free checksum of an array's bytes: (wrapping sum, xor) of its 64-bit words"""
    b = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    pad = (-b.size) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    v = b.view(np.uint64)
    return int(v.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(v)) if v.size else 0


def _cpu_worker(job):
    """oracle round trips of a list of seeds in a worker process -> (pixels coded, {seed: (stream digest, nbytes, max_n,
    decoded-image digest)}).  Serves both the all-cores CPU figure and the parity check of every distinct image."""
    seeds, reps = job
    from oracle import oracle as O
    mb = int(H * W * BPP)
    out = {}
    for seed in seeds:
        img = load_image(seed)
        for _ in range(reps):
            data, mn, _g = O.encode_image(img, WAVELET, MODE, LEVEL, QSCALE, None, mb)
            rec = O.decode_image(data, mn, C_IMG, H, W, WAVELET, LEVEL, QSCALE, None)
        out[seed] = (digest(np.frombuffer(data, np.uint8)), len(data), mn, digest(rec))
    return len(seeds) * reps * H * W, out


def spawn_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start the N rank processes (this process never touches a GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    token = "%d.%d" % (os.getpid(), time.time_ns())  # tells this job's ranks from another job's on nearby ports (dist.py)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SPIHT_JOB_TOKEN=token)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--distinct", type=int, default=0,
                    help="distinct synthetic images per GPU, cycled through the batch (0 = default: every image of the batch "
                         "is distinct, seed 1000 + global image index)")
    ap.add_argument("--cpu-sample", type=int, default=24,
                    help="images the one-core CPU baseline codes (0 = skip both CPU legs); 24 = about 11 s of one core")
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="processes of the all-cores CPU leg (SURVEY.md 8d: one independent image per core), which codes EVERY "
                         "distinct image of the batch once and is also the parity check of the timed GPU output.  0 (default): "
                         "every CPU this process may run on (nproc: sched_getaffinity), at most one per image; 1: skip")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes that synthesise the inputs (0 = cores / ranks, <= 16)")
    ap.add_argument("--pixels", choices=["float64", "float32"], default="float64",
                    help="pixel dtype.  float64 (default) is what the reference's loader produces and what the metric is quoted "
                         "on; float32 runs the single-precision forward transform PyWavelets would run on such pixels (half the "
                         "DWT read traffic); the decode side is float64 either way, as in the reference")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default): steps are software-pipelined, queued by the library's own spiht_pipeline_submit (csrc/pipeline.cpp) "
                         "-- the HBM-bound halves (DWT + pyramid of step i+1, inverse DWT of step i-1) run on one context while step i is "
                         "list-coded on another; all K steps complete inside the timed region.  "
                         "0: every step runs its stages back to back on one stream, each kernel with the whole GPU")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams (library contexts) the batch is split over.  Measured on MI355X/ROCm 7.2: chunks on "
                         "separate streams did not overlap (2 streams = same time, 4 and 8 slower), so the default is 1")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SPIHT_BENCH_DEVICE"):  # rehearsal of N ranks on a box with fewer GPUs: every rank on this device
        local_rank = int(os.environ["SPIHT_BENCH_DEVICE"])
    # under a launcher (RANK set) the distributed path is taken even with one rank, so it can be rehearsed on one GPU
    use_comm = world > 1 or ("RANK" in os.environ and os.environ.get("SPIHT_BENCH_FORCE_DIST", "1") == "1")

    B = args.batch
    nd = B if args.distinct <= 0 else max(1, min(args.distinct, B))
    seeds = [1000 + rank * B + i for i in range(nd)]  # image i of the job has seed 1000 + i
    pix = np.dtype(args.pixels)

    # ---- synthesise the inputs into the cache (before anything touches the GPU: worker processes) ----
    missing = [s for s in seeds if not os.path.exists(_cache_path(s))]
    if missing:
        import concurrent.futures as cf
        import multiprocessing as mp
        nw = args.gen_workers or max(1, min(16, (os.cpu_count() or 1) // max(1, world)))
        t_gen = time.perf_counter()
        with cf.ProcessPoolExecutor(max_workers=min(nw, len(missing)), mp_context=mp.get_context("spawn")) as ex:
            for k, _ in enumerate(ex.map(_make_cached, missing)):
                if rank == 0 and (k + 1) % 64 == 0:
                    print("[bench] synthesised %d / %d images (%.0f s)" % (k + 1, len(missing), time.perf_counter() - t_gen),
                          file=sys.stderr, flush=True)

    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec, DeviceArray
    from spiht_amd.spiht_wrapper import SpihtSettings

    ctx = _lib.default_context(local_rank)
    comm = group = None
    comm_error = None
    if use_comm:
        # the ranks' host channel (barrier, max of the times, the RCCL id) and the library's RCCL communicator.  Should RCCL
        # not come up on some node, the job still runs -- every rank codes and decodes its own shard, which is all the
        # metric needs -- and the line says that the gather did not take place.
        from spiht_amd.dist import Comm, HostGroup
        group = HostGroup(rank, world)
        try:
            if os.environ.get("SPIHT_BENCH_NO_RCCL"):  # rehearsal of the fallback
                raise RuntimeError("RCCL disabled by SPIHT_BENCH_NO_RCCL")
            comm = Comm(ctx, rank, world, group=group)
        except Exception as e:  # noqa: BLE001
            comm_error = repr(e)
            print("[bench] rank %d: RCCL communicator not available: %s" % (rank, comm_error), file=sys.stderr, flush=True)
        ok_all = group.max(0.0 if comm is not None else 1.0) == 0.0
        if not ok_all and comm is not None:  # some other rank failed: nobody gathers
            comm.close()
            comm, comm_error = None, comm_error or "another rank could not join the communicator"

    max_bits = int(H * W * BPP)  # demonstrate.py:50
    K = max(1, min(args.streams, B))
    ctxs = [ctx] + [_lib.Context(local_rank) for _ in range(K - 1)]
    settings = SpihtSettings(WAVELET, QSCALE, MODE)
    codecs = [BatchCodec(C_IMG, H, W, settings, LEVEL, max_bits, ctx=cx, pixel_dtype=pix) for cx in ctxs]
    codec = codecs[0]
    g = codec.geom
    slot = codec.slot_stride
    bounds = [(k * B // K, (k + 1) * B // K) for k in range(K)]  # chunk k of the batch runs on stream k

    # ---- inputs resident in HBM before the timed region ----
    d_img = DeviceArray(ctx, (B, C_IMG, H, W), pix)
    per = C_IMG * H * W * pix.itemsize
    for b in range(B):
        if b < nd:
            d_img.upload(load_image(seeds[b]).astype(pix, copy=False), offset_bytes=b * per)
        else:  # --distinct < batch: device-side copies of the distinct ones
            _lib.check(_lib.lib().spiht_dev_copy(ctx.handle, d_img.ptr + b * per, d_img.ptr + (b % nd) * per, per))
    base0 = load_image(seeds[0])
    d_rec_img = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
    d_nbytes = DeviceArray(ctx, (B,), np.uint64)
    d_out = DeviceArray(ctx, (B, slot), np.uint8)
    d_nbits = DeviceArray(ctx, (B,), np.uint64)
    d_maxn = DeviceArray(ctx, (B,), np.uint8)
    out_ptr, nbits_ptr, maxn_ptr = d_out.ptr, d_nbits.ptr, d_maxn.ptr
    # what the decoder reads: the encoder's own outputs, or this rank's rows of the gathered buffers
    dec_out, dec_nbits, dec_maxn = out_ptr, nbits_ptr, maxn_ptr
    if comm is not None:
        g_out = DeviceArray(ctx, (world * B, slot), np.uint8)
        g_nbits = DeviceArray(ctx, (world * B,), np.uint64)
        g_maxn = DeviceArray(ctx, (world * B,), np.uint8)
        dec_out, dec_nbits, dec_maxn = g_out.ptr + rank * B * slot, g_nbits.ptr + rank * B * 8, g_maxn.ptr + rank * B

        def gather(ctx_l):
            # the one exchange of the path: fixed-size stream slots + bit counts + start planes,
            # rank-major (spiht_amd/dist.py: rank r owns rows [r*B, (r+1)*B)); queued on the list-coding stream
            comm.gather_streams(ctx_l, out_ptr, nbits_ptr, maxn_ptr, B, slot, g_out.ptr, g_nbits.ptr, g_maxn.ptr)

    img_b = C_IMG * H * W * pix.itemsize
    rec_b = C_IMG * g["rec_h"] * g["rec_w"] * 8

    def enc_chunk(k):
        a, b = bounds[k]
        codecs[k].encode_device(d_img.ptr + a * img_b, b - a, out_ptr + a * slot, nbits_ptr + a * 8, maxn_ptr + a)

    def dec_chunk(k, src=None):
        a, b = bounds[k]
        so, sn, sm = src if src is not None else (dec_out, dec_nbits, dec_maxn)
        codecs[k].nbits_to_nbytes(sn + a * 8, b - a, d_nbytes.ptr + a * 8)
        codecs[k].decode_device(so + a * slot, d_nbytes.ptr + a * 8, sm + a, b - a, d_rec_img.ptr + a * rec_b)

    cpipe = None
    if args.pipeline and K == 1 and pix == np.float64:
        # the pipelined schedule queued by the library itself (include/spiht_hip.h: spiht_pipeline_*, csrc/pipeline.cpp): what a
        # caller in any host language gets; two list-coding contexts of its own, the HBM-bound passes on `ctx`
        from spiht_amd.batch import Pipeline
        cpipe = Pipeline(codec, B)
        ctxs.extend(cpipe.contexts()[1:])   # (its H context is `ctx`)

    def step():
        if cpipe is not None:
            cpipe.submit(d_img.ptr, out_ptr, nbits_ptr, maxn_ptr, d_rec_img.ptr, comm=comm,
                         gathered=(g_out.ptr, g_nbits.ptr, g_maxn.ptr) if comm is not None else None, rank=rank)
            return
        # every call below only queues work on the chunk's own stream
        for k in range(K):
            enc_chunk(k)
        if comm is not None:
            for cx in ctxs[1:K]:
                ctxs[0].wait_on(cx)
            gather(ctxs[0])
            for cx in ctxs[1:K]:
                cx.wait_on(ctxs[0])
        for k in range(K):
            dec_chunk(k)

    def sync_all():
        if cpipe is not None:
            cpipe.flush()  # the inverse transform of the last step (inside the timed region)
        for cx in ctxs:
            cx.synchronize()
        if group is not None:
            group.barrier()

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
    dt_rank = [dt]
    gather_ms_rank = [stages.get("gather", (0.0, 0))[0] / max(1, args.steps)]
    if group is not None:
        dt_rank = group.allgather(dt)
        gather_ms_rank = group.allgather(gather_ms_rank[0])
        dt = max(dt_rank)

    # ---- correctness of what was timed (outside the timed region) ----
    nbits = d_nbits.download()
    maxn = d_maxn.download()
    streams_gpu = d_out.download()
    # digests of every distinct image's stream and decoded image, as the GPU left them after the last timed step
    gpu_dig = {}
    rec1 = np.empty((C_IMG, g["rec_h"], g["rec_w"]), np.float64)
    for b in range(nd):
        nby = (int(nbits[b]) + 7) // 8
        ctx.download(rec1, d_rec_img.ptr + b * rec_b)
        gpu_dig[seeds[b]] = (digest(streams_gpu[b, :nby]), nby, int(maxn[b]), digest(rec1))
        if b == 0:
            mae = float(np.abs(rec1[:, :H, :W] - base0).mean())
    copies_ok = True
    for b in range(nd, B):  # cycled copies must equal their originals
        copies_ok = copies_ok and int(nbits[b]) == int(nbits[b % nd]) and np.array_equal(streams_gpu[b], streams_gpu[b % nd])

    gather_ok = foreign_checked = None
    if comm is not None:
        # own rows of the gathered buffers are the encoder's outputs ...
        ga, gn, gm = g_out.download(), g_nbits.download(), g_maxn.download()
        gather_ok = bool(np.array_equal(ga[rank * B:(rank + 1) * B], streams_gpu) and
                         np.array_equal(gn[rank * B:(rank + 1) * B], nbits) and np.array_equal(gm[rank * B:(rank + 1) * B], maxn))
        # ... and rows of OTHER ranks are what this GPU makes of the same images (re-encoded here, a seeded sample)
        foreign_checked = 0
        if world > 1:
            rng = np.random.default_rng(77 + rank)
            others = [r for r in range(world) if r != rank]
            picks = [(int(rng.choice(others)), int(rng.integers(0, B))) for _ in range(4)]
            d_f = DeviceArray(ctx, (C_IMG, H, W), pix)
            for r2, i2 in picks:
                seed2 = 1000 + r2 * B + (i2 % nd)
                d_f.upload(load_image(seed2).astype(pix, copy=False))
                codec.encode_device(d_f.ptr, 1, out_ptr, nbits_ptr, maxn_ptr)
                ctx.synchronize()
                s1 = np.empty(slot, np.uint8)
                nb1, mn1 = np.empty(1, np.uint64), np.empty(1, np.uint8)
                ctx.download(s1, out_ptr)
                ctx.download(nb1, nbits_ptr)
                ctx.download(mn1, maxn_ptr)
                row = r2 * B + i2
                gather_ok = gather_ok and bool(np.array_equal(ga[row], s1) and gn[row] == nb1[0] and gm[row] == mn1[0])
                foreign_checked += 1
        gather_ok = group.max(0.0 if gather_ok else 1.0) == 0.0  # all ranks

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
        # HBM traffic of that kernel: a committed PMC measurement (tools/collect_traffic.py: separate rocprofv3 --pmc
        # passes, FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes), per image, scaled to this launch
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "dwt_l1_traffic.json")
        if os.path.exists(tpath) and pix == np.float64:
            try:
                tj = json.load(open(tpath))
                traffic = round(tj["hbm_bytes_per_image"] * per_launch)
                traffic_src = "profiles/dwt_l1_traffic.json: PMC counters of a separate rocprofv3 run (%s images per launch), " \
                              "per image x %d; not measured in this run" % (tj.get("images_per_launch", "?"), per_launch)
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
        # the drop-in call itself, host array -> bytes and bytes -> host array (the reference's timing pattern,
        # encode_decode.py:55-72): includes the 49.8 MB pixel upload / download over PCIe; never part of `value`
        import spiht_amd
        h_enc, h_dec = [], []
        img0 = base0.astype(pix, copy=False)
        for it in range(6):
            t1 = time.perf_counter()
            er = spiht_amd.encode_image(img0, settings, LEVEL, max_bits)
            t2 = time.perf_counter()
            spiht_amd.decode_image(er, settings)
            t3 = time.perf_counter()
            if it:  # the first call sizes the context's buffers
                h_enc.append((t2 - t1) * 1e3)
                h_dec.append((t3 - t2) * 1e3)
        single["host_api_encode_ms"] = round(sorted(h_enc)[2], 3)
        single["host_api_decode_ms"] = round(sorted(h_dec)[2], 3)
        single["host_api_stream_equals_batch"] = bool(digest(np.frombuffer(er.encoded_bytes, np.uint8)) == gpu_dig[seeds[0]][0])

        # the same kernels with the GPU to themselves (one serial round trip of the batch after the timed region): in
        # the pipelined schedule the numbers above are those of kernels that share the GPU with the list coder
        alone = None
        if cpipe is not None:
            own = (out_ptr, nbits_ptr, maxn_ptr)
            enc_chunk(0)
            dec_chunk(0, own)
            ctx.synchronize()
            ctx.reset_timing()
            ctx.set_timing(True)
            for _ in range(3):
                enc_chunk(0)
                dec_chunk(0, own)
            ctx.synchronize()
            ctx.set_timing(False)
            tm = ctx.timing()
            a_ms = {k2: v2[0] / v2[1] for k2, v2 in tm.items() if v2[1] and k2 in ("dwt_level1", "idwt_level1", "pyramid")}
            if "dwt_level1" in a_ms:
                # (the pipeline leaves nothing on `ctx` between its calls: these are the library's single-call settings)
                alone = {"context_options": {k2: ctx.get_option(k2) for k2 in ("idwt_groups", "pads_persist", "l1_flags")},
                         "dwt_level1_ms": round(a_ms["dwt_level1"], 4),
                         "dwt_level1_frac": round(dwt_bytes / (a_ms["dwt_level1"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "idwt_level1_ms": round(a_ms.get("idwt_level1", float("nan")), 4),
                         "pyramid_ms": round(a_ms.get("pyramid", float("nan")), 4)}

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
        nb_i = per_launch * C_IMG * (H * W * 8 + h1 * w1 * 8 + 3 * h1 * w1 * 4)
        nb_p = per_launch * (4 * n_coef + n_par + n_par // 4)
        other = {
            # inverse level 1: read 3 int32 bands + float64 LL, write the float64 image
            "idwt_level1": _gbs("idwt_level1", nb_i),
            # significance pyramid: read 4 B per coefficient, write 1 B per parent (D) + 1 B per grand-parent (L)
            "pyramid": _gbs("pyramid", nb_p),
        }
        if alone is not None:
            # in the pipelined schedule the inverse transform shares the GPU with the list decoder (by design)
            for key, nb, ak in (("idwt_level1", nb_i, "idwt_level1_ms"), ("pyramid", nb_p, "pyramid_ms")):
                if other.get(key) is not None and alone[ak] == alone[ak]:
                    other[key]["alone_ms"] = alone[ak]
                    other[key]["alone_frac"] = round(nb / (alone[ak] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # PMC traffic of those two passes (tools/collect_traffic.py), per image, beside the algorithmic bytes
        opath = os.path.join(ROOT, "profiles", "hbm_traffic_other.json")
        if os.path.exists(opath) and pix == np.float64:
            try:
                ot = json.load(open(opath))
                for key, src in (("idwt_level1", "idwt_level1"), ("pyramid", "pyramid_rounds")):
                    if other.get(key) is not None:
                        other[key]["traffic_bytes_per_image"] = round(ot[src]["hbm_bytes_per_image"])
                        other[key]["algorithmic_bytes_per_image"] = ot[src]["algorithmic_bytes_per_image"]
                        other[key]["traffic_source"] = "profiles/hbm_traffic_other.json (committed PMC measurement, separate run, " \
                                                       "%s images per launch)" % ot.get("images_per_launch", "?")
                occ = ot.get("idwt_level1_with_occupancy_words", {}).get("%g bpp" % BPP)
                if occ and other.get("idwt_level1") is not None:
                    # what this run's inverse level 1 moves: the decoder's occupancy words are on, empty tiles' detail bands
                    # unread -- so its rate is quoted on the bytes it MOVES (PMC, this bit rate), not on the dense 81 MB / image
                    o1 = other["idwt_level1"]
                    o1["with_occupancy_words"] = {k2: (round(v2) if k2.endswith("per_image") else v2) for k2, v2 in occ.items()}
                    moved = per_launch * occ["hbm_bytes_per_image"]
                    o1["dense_bytes_frac"] = o1["frac"]
                    o1["achieved_GBps"] = round(moved / (o1["ms_per_group"] * 1e-3) / 1e9, 1)
                    o1["frac"] = round(o1["achieved_GBps"] / HBM_PEAK_GBS, 4)
                    if "alone_ms" in o1:
                        o1["dense_bytes_alone_frac"] = o1["alone_frac"]
                        o1["alone_frac"] = round(moved / (o1["alone_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            except Exception:
                pass
        # every HBM pass of a step counted once: the transform levels in both directions + the pyramid
        hh, ww, step_bytes = [H], [W], 0
        for _l in range(g["level"]):
            hh.append((hh[-1] + 6 - 1) // 2)
            ww.append((ww[-1] + 6 - 1) // 2)
        for _l in range(1, g["level"] + 1):
            ll_b = 4 if _l == g["level"] else 8  # the coarsest approximation lives in the int32 array
            step_bytes += 2 * C_IMG * (hh[_l - 1] * ww[_l - 1] * 8 + hh[_l] * ww[_l] * (ll_b + 12))
        step_bytes = B * (step_bytes + 4 * n_coef + n_par + n_par // 4)
        whole = {"algorithmic_bytes_per_step": step_bytes, "achieved_GBps": round(step_bytes / (dt / args.steps) / 1e9, 1),
                 "frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                 "what": "every HBM pass of the step once (7 forward + 7 inverse levels, pyramid) / ms_per_step; the list coder's own "
                         "traffic (a few MB per image) not counted"}
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
                       "images_per_gpu": B, "distinct_images_per_gpu": nd, "seeds": "1000 + global image index",
                       "streams": K, "images_per_launch": per_launch,
                       "decoder_waves": 8 if cpipe is not None else 12,
                       "schedule": ("steps software-pipelined: HBM-bound passes of steps i+1 / i-1 on one stream while step i is "
                                    "list-coded on another, queued by the library (spiht_pipeline_submit)"
                                    if cpipe is not None else "stages back to back"),
                       "max_bits": max_bits, "images_per_s": round(total_images / dt, 2),
                       "coeff_array": [C_IMG, g["enc_h"], g["enc_w"]], "ll": [g["ll_h"], g["ll_w"]],
                       "per_rank_ms_per_step": [round(v / args.steps * 1e3, 3) for v in dt_rank],
                       "gather": (dict(comm.info(), library=_lib.lib().spiht_rccl_library().decode(),
                                       stream_ms_per_step_per_rank=[round(v, 4) for v in gather_ms_rank],
                                       bytes_per_rank_per_step=B * (slot + 9),
                                       where="spiht_gather_streams (ncclAllGather on the list-coding stream, timed with an "
                                             "event pair on that stream); every rank decodes its rows of the gathered buffer")
                                  if comm is not None else
                                  ({"failed": comm_error, "library": _lib.lib().spiht_rccl_library().decode(),
                                    "note": "RCCL did not come up: every rank coded and decoded its own "
                                    "shard, no stream gather took place"} if comm_error else None))},
            "roofline": {"bound": "hbm", "kernel": ("k_dwt_level<6>" if pix == np.float64 else "k_dwt_level_f32<6>") +
                         " (forward DWT level 1, fused quantise)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dwt_bytes, "avg_launch_ms": round(avg_ms, 4),
                         "kernel_alone": alone, "whole_step": whole},
            "stages_ms_per_step_summed_over_streams": {k: round(v[0] / args.steps, 3) for k, v in stages.items() if v[1]},
            "roofline_other_hbm_passes": other,
            "single_image_latency": single,
            "check": {"nbits_all_equal_budget": bool((nbits == max_bits).all()), "max_n_values": sorted(set(int(v) for v in maxn)),
                      "mean_abs_err_image0": round(mae, 5), "cycled_copies_equal": bool(copies_ok),
                      "gather_rows_match": gather_ok, "gather_foreign_rows_checked_per_rank": foreign_checked},
        }

        # ---- CPU legs: the oracle (port of the reference algorithm) ----
        result["cpu_baseline"] = None
        if args.cpu_sample > 0 and world == 1:  # the CPU legs run at N = 1 only
            from oracle import oracle as O
            # (1) one core, a bounded sample: the reported baseline
            ns = args.cpu_sample
            imgs = [load_image(seeds[i % nd]).astype(pix, copy=False) for i in range(min(ns, nd))]
            tc = time.perf_counter()
            enc = [O.encode_image(imgs[i % len(imgs)], WAVELET, MODE, LEVEL, QSCALE, None, max_bits)[:2] for i in range(ns)]
            t_enc = time.perf_counter() - tc
            tc = time.perf_counter()
            for i in range(ns):
                O.decode_image(enc[i][0], enc[i][1], C_IMG, H, W, WAVELET, LEVEL, QSCALE, None)
            t_dec = time.perf_counter() - tc
            del imgs
            result["cpu_baseline"] = {
                "value": round(ns * H * W / (t_enc + t_dec) / 1e6, 3), "unit": "Mpixels/s", "cores": 1,
                "kind": "port",
                "sample": "%d of the batch's 1080p images, encode %.2fs + decode %.2fs, oracle/liboracle.so (gcc -O3), "
                          "host has %d cores" % (ns, t_enc, t_dec, os.cpu_count() or 0)}
            # (2) one independent image per core (the reference is single-threaded per image; SURVEY.md 8d ii): every
            # distinct image of the batch once -- which is also the parity check of ALL of them
            checked = 0
            all_stream_ok = all_img_ok = True
            if args.cpu_cores != 1 and pix == np.float64:
                import concurrent.futures as cf
                import multiprocessing as mp
                nproc = _nproc()
                ncores = max(2, min(args.cpu_cores if args.cpu_cores > 0 else nproc, nproc, nd, 512))
                jobs = [(seeds[k::ncores], 1) for k in range(ncores) if seeds[k::ncores]]
                with cf.ProcessPoolExecutor(max_workers=ncores, mp_context=mp.get_context("spawn")) as ex:
                    list(ex.map(_cpu_worker, [([seeds[0]], 1)] * ncores))  # start-up (imports, library load) outside the timing
                    tc = time.perf_counter()
                    res = list(ex.map(_cpu_worker, jobs))
                    t_all = time.perf_counter() - tc
                npx = sum(r[0] for r in res)
                result["cpu_baseline_all_cores"] = {
                    "value": round(npx / t_all / 1e6, 3), "unit": "Mpixels/s", "cores": ncores, "kind": "port",
                    "sample": "%d processes (nproc %d, host %d), the batch's %d distinct 1080p images once each, one image per "
                              "process at a time, %.2fs" % (ncores, nproc, os.cpu_count() or 0, nd, t_all)}
                for _, dd in res:
                    for seed, (sd, nby, mn, idg) in dd.items():
                        gd = gpu_dig[seed]
                        all_stream_ok = all_stream_ok and gd[0] == sd and gd[1] == nby and gd[2] == mn
                        all_img_ok = all_img_ok and gd[3] == idg
                        checked += 1
            else:  # no all-cores leg: image 0 only
                data0, mn0 = enc[0]
                gd = gpu_dig[seeds[0]]
                all_stream_ok = gd[0] == digest(np.frombuffer(data0, np.uint8)) and gd[2] == mn0
                all_img_ok = gd[3] == digest(O.decode_image(data0, mn0, C_IMG, H, W, WAVELET, LEVEL, QSCALE, None))
                checked = 1
            result["check"]["images_checked_vs_oracle"] = checked
            result["check"]["stream_bit_exact_vs_oracle"] = bool(all_stream_ok)
            result["check"]["decoded_image_bit_exact_vs_oracle"] = bool(all_img_ok)
        print(json.dumps(result))
        sys.stdout.flush()

    if group is not None:
        group.barrier()
    if comm is not None:
        comm.close()
    if group is not None:
        group.close()


if __name__ == "__main__":
    main()
