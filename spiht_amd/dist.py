"""Multi-GPU layout helpers (host logic only; no torch import here).

The path shards across images and not within one: rank r of `world` owns a contiguous block of the
batch, codes it on its own GPU, and the only exchange is a gather of fixed-capacity stream slots
(`slot_stride` bytes per image) plus the per-image bit counts and start planes.  The collective itself is issued
by the caller (`torch.distributed.all_gather_into_tensor` over RCCL in bench.py, gloo in the CPU tests); these
helpers define who owns what and how the gathered buffers are read back.
"""
import numpy as np


def partition(total, world, rank):
    """Contiguous block [start, stop) of `total` images owned by `rank`: the first total % world ranks get one
    extra image (equal blocks when world divides total, as the all-gather of equal slots needs)."""
    if world < 1 or not 0 <= rank < world or total < 0:
        raise ValueError("bad partition arguments")
    q, r = divmod(total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def padded_count(total, world):
    """Images per rank once every rank is padded to the same count (all_gather needs equal shapes)."""
    return (total + world - 1) // world


def owner_of(index, total, world):
    """Rank that owns global image `index` under partition()."""
    q, r = divmod(total, world)
    edge = r * (q + 1)
    return index // (q + 1) if index < edge else r + (index - edge) // max(q, 1)


def pack_slots(streams, slot_stride, count=None):
    """List of (bytes, max_n) -> (uint8 [count, slot_stride], uint64 nbytes [count], uint8 max_n [count]); rows past
    len(streams) are padding (nbytes 0)."""
    count = len(streams) if count is None else count
    if slot_stride % 4:
        raise ValueError("slot_stride must be a multiple of 4")
    slots = np.zeros((count, slot_stride), dtype=np.uint8)
    nbytes = np.zeros(count, dtype=np.uint64)
    maxn = np.zeros(count, dtype=np.uint8)
    for i, (data, mn) in enumerate(streams):
        if len(data) > slot_stride:
            raise ValueError("stream of %d bytes does not fit a %d-byte slot" % (len(data), slot_stride))
        slots[i, :len(data)] = np.frombuffer(data, np.uint8)
        nbytes[i] = len(data)
        maxn[i] = mn
    return slots, nbytes, maxn


def unpack_gathered(slots, nbytes, maxn, total, world):
    """Inverse of the gather: slots [world*per, slot_stride] etc. (rank-major, each rank padded to `per` rows) ->
    list of (bytes, max_n) for the `total` real images in global order."""
    per = padded_count(total, world)
    slots = np.asarray(slots).reshape(world * per, -1)
    nbytes = np.asarray(nbytes).reshape(world * per)
    maxn = np.asarray(maxn).reshape(world * per)
    out = []
    for r in range(world):
        a, b = partition(total, world, r)
        for k in range(b - a):
            row = r * per + k
            out.append((slots[row, :int(nbytes[row])].tobytes(), int(maxn[row])))
    return out
