"""CPU tests of the multi-GPU path: world_size-2 gloo processes shard a batch image-per-rank, code their shard
(with the CPU oracle standing in for the GPU -- this tests the sharding/gather logic, not the kernels), all-gather
the fixed-size stream slots and check that every rank reassembles exactly the streams a single process makes."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import synth_image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties():
    from spiht_amd.dist import owner_of, padded_count, partition
    for total in [0, 1, 5, 8, 13, 256, 2048]:
        for world in [1, 2, 3, 8]:
            seen = []
            for r in range(world):
                a, b = partition(total, world, r)
                assert 0 <= a <= b <= total
                assert b - a <= padded_count(total, world)
                seen.extend(range(a, b))
                for i in range(a, b):
                    assert owner_of(i, total, world) == r
            assert seen == list(range(total))
    assert partition(2048, 8, 3) == (768, 1024)  # BASELINE config 4: 256 images per GPU
    with pytest.raises(ValueError):
        partition(4, 2, 2)


def test_pack_unpack_roundtrip():
    from spiht_amd.dist import pack_slots, padded_count, partition, unpack_gathered
    rng = np.random.default_rng(0)
    total, world, slot = 7, 3, 64
    streams = [(rng.integers(0, 256, int(rng.integers(0, 60)), dtype=np.uint8).tobytes(), int(rng.integers(0, 14)))
               for _ in range(total)]
    per = padded_count(total, world)
    parts = []
    for r in range(world):
        a, b = partition(total, world, r)
        parts.append(pack_slots(streams[a:b], slot, per))
    slots = np.concatenate([p[0] for p in parts])
    nbytes = np.concatenate([p[1] for p in parts])
    maxn = np.concatenate([p[2] for p in parts])
    assert unpack_gathered(slots, nbytes, maxn, total, world) == streams
    with pytest.raises(ValueError):
        pack_slots([(b"x" * 70, 1)], slot)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        from oracle import oracle as O
        from spiht_amd.dist import pack_slots, padded_count, partition, unpack_gathered
        dist.init_process_group("gloo", rank=rank, world_size=world)
        c, H, W, level, max_bits = 3, 40, 56, 2, 2500
        slot = ((max_bits + 7) // 8 + 3) & ~3
        a, b = partition(total, world, rank)
        mine = []
        for i in range(a, b):  # image i of the batch has seed 1000 + i
            data, mn, _ = O.encode_image(synth_image(1000 + i, c, H, W), "bior2.2", "reflect", level, 50.0, None, max_bits)
            mine.append((data, mn))
        per = padded_count(total, world)
        slots, nbytes, maxn = pack_slots(mine, slot, per)
        g_slots = torch.zeros((world * per, slot), dtype=torch.uint8)
        g_nb = torch.zeros(world * per, dtype=torch.int64)
        g_mn = torch.zeros(world * per, dtype=torch.uint8)
        dist.all_gather_into_tensor(g_slots, torch.from_numpy(slots))
        dist.all_gather_into_tensor(g_nb, torch.from_numpy(nbytes.astype(np.int64)))
        dist.all_gather_into_tensor(g_mn, torch.from_numpy(maxn))
        got = unpack_gathered(g_slots.numpy(), g_nb.numpy().astype(np.uint64), g_mn.numpy(), total, world)
        # every rank decodes the image it does NOT own from the gathered slots and checks it against a local encode
        other = (b % total) if total else 0
        ref, mn, _ = O.encode_image(synth_image(1000 + other, c, H, W), "bior2.2", "reflect", level, 50.0, None, max_bits)
        ok = got[other] == (ref, mn) and len(got) == total
        dec = O.decode_image(got[other][0], got[other][1], c, H, W, "bior2.2", level, 50.0, None)
        ok = ok and dec.shape == (c, H, W)
        dist.barrier()
        dist.destroy_process_group()
        import hashlib
        q.put((rank, ok, [hashlib.sha1(s[0] + bytes([s[1]])).hexdigest() for s in got]))
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("total", [4, 5])
def test_two_rank_gloo_gather(total, oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    res.sort()
    assert all(r[1] is True for r in res), res
    assert res[0][2] == res[1][2]  # both ranks hold the same gathered streams


def _id_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        from spiht_amd.dist import exchange_id
        payload = bytes(range(128)) if rank == 0 else None
        got = exchange_id(rank, world, payload, addr="127.0.0.1", port=port, timeout=60)
        q.put((rank, got == bytes(range(128))))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


def test_id_exchange_two_and_three_ranks():
    """The host channel that carries rank 0's RCCL id to the other ranks (spiht_amd/dist.py:exchange_id): a TCP exchange
    on MASTER_ADDR, ports MASTER_PORT+1..; late starters, a busy first port and world sizes 2 and 3."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    for world, order in ((2, (1, 0)), (3, (2, 0, 1))):
        port = _free_port()
        blocker = socket.socket()      # a foreign service on the first candidate port
        try:
            blocker.bind(("127.0.0.1", port + 1))
            blocker.listen(1)
        except OSError:
            blocker = None
        q = ctx.Queue()
        procs = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in order]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
        if blocker is not None:
            blocker.close()
        assert res == [(r, True) for r in range(world)], res


def _group_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        from spiht_amd.dist import HostGroup
        g = HostGroup(rank, world, addr="127.0.0.1", port=port, timeout=60)
        g.barrier()
        m = g.max(rank * 1.5)
        b = g.bcast(bytes(range(128)) if rank == 0 else None)
        m2 = g.max(-float(rank))
        ag = g.allgather(10.0 + rank)
        g.barrier()
        g.close()
        q.put((rank, m, m2, b == bytes(range(128)) and ag == [10.0 + r for r in range(world)]))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


def test_host_group_barrier_max_bcast():
    """The job's host channel (spiht_amd/dist.py:HostGroup: barrier around the timed region, maximum of the ranks' times,
    rank 0's RCCL id to everyone) with four ranks that start in scrambled order."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    world, port = 4, _free_port()
    q = ctx.Queue()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, q)) for r in (3, 1, 0, 2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, 4.5, 0.0, True) for r in range(world)], res


def test_gathered_rows_of_every_rank_world_8():
    """spiht_pipeline_submit_gather hands the decoder of rank r its rows of the GATHERED arrays: spiht_gather_row_offsets is
    that arithmetic, on its own (no device).  World 8 on fake buffers laid out as the all-gather leaves them (rank-major,
    spiht_amd/dist.py): every rank's offsets select exactly the slots / bit counts / start planes that rank contributed,
    and out-of-range arguments are refused."""
    import ctypes as C
    from spiht_amd import _lib
    L = _lib.lib()
    world, B, slot = 8, 256, 129600
    rng = np.random.default_rng(5)
    own = [(rng.integers(0, 256, (B, slot), dtype=np.uint8), rng.integers(0, 1 << 40, B).astype(np.uint64),
            rng.integers(0, 31, B).astype(np.uint8)) for _ in range(2)]  # (two distinct shards are enough to tell rows apart)
    shard = lambda r: own[r & 1]  # noqa: E731
    all_slots = np.concatenate([shard(r)[0] for r in range(world)]).reshape(-1)
    all_nbits = np.concatenate([shard(r)[1] for r in range(world)])
    all_maxn = np.concatenate([shard(r)[2] for r in range(world)])
    for r in range(world):
        o = [C.c_uint64() for _ in range(3)]
        assert L.spiht_gather_row_offsets(r, world, B, slot, *[C.byref(v) for v in o]) == 0
        os_, on, om = (int(v.value) for v in o)
        assert (os_, on, om) == (r * B * slot, r * B * 8, r * B)
        assert np.array_equal(all_slots[os_:os_ + B * slot].reshape(B, slot), shard(r)[0])
        assert np.array_equal(all_nbits.view(np.uint8)[on:on + 8 * B].view(np.uint64), shard(r)[1])
        assert np.array_equal(all_maxn[om:om + B], shard(r)[2])
    o = [C.c_uint64() for _ in range(3)]
    for bad in ((8, 8), (-1, 8), (0, 0)):
        assert L.spiht_gather_row_offsets(bad[0], bad[1], B, slot, *[C.byref(v) for v in o]) != 0
    assert L.spiht_gather_row_offsets(1, 2, 1 << 40, 1 << 40, *[C.byref(v) for v in o]) != 0  # overflow


def _job_worker(job, rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["SPIHT_JOB_TOKEN"] = "job%d" % job  # what bench.py's spawner sets; torchrun: TORCHELASTIC_RUN_ID
        from spiht_amd.dist import HostGroup
        g = HostGroup(rank, world, addr="127.0.0.1", port=port, timeout=60)
        b = g.bcast(bytes([job]) * 128 if rank == 0 else None)
        m = g.max(float(10 * job + rank))
        g.barrier()
        g.close()
        q.put((job, rank, b == bytes([job]) * 128, m))
    except Exception as e:  # pragma: no cover
        q.put((job, rank, repr(e), None))


def test_two_jobs_with_overlapping_port_ranges_do_not_mix():
    """Two jobs of the same world size on one host whose MASTER_PORTs are one apart share 15 of their 16 candidate ports:
    a peer must join ITS job's rank 0 (the hello carries a job token; a foreign rank 0 is skipped), or it would be handed
    the other job's RCCL id.  Peers start before their rank 0, and job 1's rank 0 listens first -- on the very port job
    0's peers probe first."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    order = [(0, 1, port), (1, 1, port - 1), (1, 0, port - 1), (0, 0, port)]
    # job 1 uses MASTER_PORT = port - 1: its rank 0 binds port (= job 0's MASTER_PORT + 0 ... first candidate of job 1)
    procs = []
    for job, rank, mp_port in order:
        p = ctx.Process(target=_job_worker, args=(job, rank, 2, mp_port, q))
        p.start()
        procs.append(p)
        time.sleep(0.3)
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, 0, True, 1.0), (0, 1, True, 1.0), (1, 0, True, 11.0), (1, 1, True, 11.0)], res


def test_candidate_ports_stay_in_range():
    from spiht_amd.dist import _ports
    assert _ports(29500) == list(range(29501, 29517))
    assert _ports(65530) == [65531, 65532, 65533, 65534, 65535]
    assert _ports(65535) == []
