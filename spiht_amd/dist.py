"""Multi-GPU side of the path (no torch anywhere).

The path shards across images and not within one: rank r of `world` owns a contiguous block of the
batch, codes it on its own GPU, and the only exchange is a gather of fixed-capacity stream slots
(`slot_stride` bytes per image) plus the per-image bit counts and start planes.  The collective is the library's
own (`spiht_gather_streams`: ncclAllGather on RCCL, queued on the context's stream); `Comm` binds it.  The job's
RCCL id travels from rank 0 to the others over a small TCP exchange on MASTER_ADDR (`exchange_id`) -- the launcher
contract is torchrun's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), nothing of torch is
imported.  The helpers below `Comm` define who owns what and how gathered buffers are read back.
"""
import ctypes as C
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b"SPIHTID2"
_PORT_SPAN = 16  # candidate ports MASTER_PORT+1 .. MASTER_PORT+_PORT_SPAN (MASTER_PORT itself is the launcher's store)


def _job_token(addr, base):
    """8 bytes every rank of ONE job computes alike and another job on the same host does not: two jobs with nearby
    MASTER_PORTs (29500 / 29501) have overlapping candidate ranges, and a peer must not take the other job's rank 0 for
    its own -- it would receive a foreign RCCL id and hang in ncclCommInitRank.  From the launcher's rendezvous point,
    its run id where there is one (torchrun: TORCHELASTIC_RUN_ID) and SPIHT_JOB_TOKEN (set by bench.py's own spawner)."""
    import hashlib
    key = "%s:%d|%s|%s" % (addr, base, os.environ.get("TORCHELASTIC_RUN_ID", ""), os.environ.get("SPIHT_JOB_TOKEN", ""))
    return hashlib.sha1(key.encode()).digest()[:8]


def _ports(base):
    """the candidate ports, inside the valid range whatever MASTER_PORT is"""
    return [p for p in range(base + 1, base + _PORT_SPAN + 1) if 0 < p <= 65535]


def _listen(addr, base, backlog, what):
    """rank 0's listening socket: the first free candidate port, on MASTER_ADDR's interface (rank 0 runs there by the
    launcher's contract) -- on every interface only if that name is not an address of this host"""
    import errno
    hosts = [addr, ""]
    for p in _ports(base):
        for host in list(hosts):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((host, p))
                s.listen(backlog)
                return s
            except socket.gaierror:   # MASTER_ADDR does not resolve here
                s.close()
                hosts = [""]
            except OSError as e:
                s.close()
                if host and e.errno == errno.EADDRNOTAVAIL:  # ... or is not an address of this host
                    hosts = [""]
                    continue
                break  # port taken: the next one
    raise RuntimeError("no free port in %d..%d for %s" % (base + 1, base + _PORT_SPAN, what))


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return buf


def exchange_id(rank, world, payload=None, addr=None, port=None, timeout=300.0):
    """Rank 0's `payload` (bytes) to every rank; returns it.  Rank 0 listens on the first free port of
    MASTER_PORT+1.. and serves world-1 peers; a peer tries the candidate ports in turn until one answers with the
    handshake of THIS job (magic + world size), so a foreign service on a candidate port is skipped."""
    if world == 1:
        return payload
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
    token = _job_token(addr, base)
    hello = _MAGIC + token + struct.pack("<II", world, 0)
    deadline = time.time() + timeout
    if rank == 0:
        srv = _listen(addr, base, world, "the id exchange")
        served = 0
        srv.settimeout(1.0)
        try:
            while served < world - 1:
                if time.time() > deadline:
                    raise TimeoutError("id exchange: %d of %d peers arrived" % (served, world - 1))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(5.0)
                    try:
                        if _recv_exact(conn, len(hello)) != hello:
                            continue  # another job's peer (or anything else): not served
                        conn.sendall(_MAGIC + token + struct.pack("<I", len(payload)) + payload)
                        served += 1
                    except (OSError, ConnectionError):
                        continue
        finally:
            srv.close()
        return payload
    while True:
        for p in _ports(base):
            try:
                with socket.create_connection((addr, p), timeout=2.0) as c:
                    c.settimeout(5.0)
                    c.sendall(hello)
                    head = _recv_exact(c, len(_MAGIC) + len(token) + 4)
                    if head[:len(_MAGIC) + len(token)] != _MAGIC + token:
                        continue  # another job's rank 0
                    (n,) = struct.unpack("<I", head[len(_MAGIC) + len(token):])
                    return _recv_exact(c, n)
            except (OSError, ConnectionError):
                continue
        if time.time() > deadline:
            raise TimeoutError("id exchange: rank 0 not reachable on %s:%d..%d" % (addr, base + 1, base + _PORT_SPAN))
        time.sleep(0.2)


class HostGroup:
    """The job's ranks as a star of TCP connections to rank 0 (same ports as `exchange_id`): the host-side channel for what
    must work even where RCCL does not come up -- handing out the RCCL id, the barrier around the timed region, the
    maximum of the ranks' times.  Payloads are a few bytes; no data of the path travels here."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.peers, self.sock = {}, None
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        token = _job_token(addr, base)
        hello = _MAGIC + token + struct.pack("<II", self.world, 1)
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = _listen(addr, base, self.world, "the job's host channel")
            srv.settimeout(1.0)
            try:
                while len(self.peers) < self.world - 1:
                    if time.time() > deadline:
                        raise TimeoutError("host channel: %d of %d peers arrived" % (len(self.peers), self.world - 1))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    try:
                        conn.settimeout(5.0)
                        head = _recv_exact(conn, len(hello) + 4)
                        (r,) = struct.unpack("<I", head[len(hello):])
                        if head[:len(hello)] != hello or not 0 < r < self.world or r in self.peers:
                            conn.close()  # (a peer of another job included: its token differs)
                            continue
                        conn.sendall(_MAGIC + token)
                        conn.settimeout(timeout)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self.peers[r] = conn
                    except (OSError, ConnectionError):
                        conn.close()
            finally:
                srv.close()
            return
        while self.sock is None:
            for p in _ports(base):
                try:
                    c = socket.create_connection((addr, p), timeout=2.0)
                    c.settimeout(5.0)
                    c.sendall(hello + struct.pack("<I", self.rank))
                    if _recv_exact(c, len(_MAGIC) + len(token)) != _MAGIC + token:
                        c.close()
                        continue
                    c.settimeout(timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.sock = c
                    break
                except (OSError, ConnectionError):
                    continue
            if self.sock is None:
                if time.time() > deadline:
                    raise TimeoutError("host channel: rank 0 not reachable on %s:%d..%d" % (addr, base + 1, base + _PORT_SPAN))
                time.sleep(0.2)

    def _exchange(self, mine, combine):
        """every rank contributes `mine` (bytes); rank 0 combines the list (rank order) into one reply for all"""
        if self.world == 1:
            return combine([mine])
        if self.rank == 0:
            parts = [mine] + [None] * (self.world - 1)
            for r, c in self.peers.items():
                (n,) = struct.unpack("<I", _recv_exact(c, 4))
                parts[r] = _recv_exact(c, n)
            out = combine(parts)
            for c in self.peers.values():
                c.sendall(struct.pack("<I", len(out)) + out)
            return out
        self.sock.sendall(struct.pack("<I", len(mine)) + mine)
        (n,) = struct.unpack("<I", _recv_exact(self.sock, 4))
        return _recv_exact(self.sock, n)

    def barrier(self):
        self._exchange(b"", lambda parts: b"")

    def max(self, value):
        out = self._exchange(struct.pack("<d", float(value)), lambda parts: struct.pack("<d", max(struct.unpack("<d", p)[0] for p in parts)))
        return struct.unpack("<d", out)[0]

    def allgather(self, value):
        """every rank's double, in rank order, on every rank"""
        out = self._exchange(struct.pack("<d", float(value)), lambda parts: b"".join(parts))
        return [v[0] for v in struct.iter_unpack("<d", out)]

    def bcast(self, payload=None):
        """rank 0's payload to every rank"""
        return self._exchange(payload if self.rank == 0 else b"", lambda parts: parts[0])

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


class Comm:
    """The job's RCCL communicator (spiht_comm of include/spiht_hip.h) on one context's GPU."""

    def __init__(self, ctx, rank=None, world=None, group=None):
        """group: a HostGroup to carry rank 0's id to the others (else a one-shot exchange_id)"""
        from . import _lib
        self._lib = _lib.lib()
        self._check = _lib.check
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        ident = None
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            try:
                self._check(self._lib.spiht_comm_unique_id(buf))
                ident = bytes(buf)
            except Exception:
                if group is None:
                    raise
                ident = b""  # the others must not wait for an id that will not come
        ident = group.bcast(ident) if group is not None else exchange_id(self.rank, self.world, ident)
        if len(ident) != 128:
            raise RuntimeError("rank 0 could not create the RCCL id")
        h = C.c_void_p()
        self._check(self._lib.spiht_comm_create(ctx.handle, (C.c_uint8 * 128).from_buffer_copy(ident), self.world,
                                                self.rank, C.byref(h)))
        self.handle = h

    def info(self):
        w, r, v = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.spiht_comm_info(self.handle, C.byref(w), C.byref(r), C.byref(v)))
        return dict(world=w.value, rank=r.value, rccl_version=v.value)

    def gather_streams(self, ctx, d_slots, d_nbits, d_max_n, B, slot_stride, d_all_slots, d_all_nbits, d_all_max_n):
        """queue the all-gather on ctx's stream (device pointers as ints); does not block"""
        vp = C.c_void_p
        self._check(self._lib.spiht_gather_streams(ctx.handle, self.handle, vp(d_slots), vp(d_nbits), vp(d_max_n), int(B),
                                                   int(slot_stride), vp(d_all_slots), vp(d_all_nbits), vp(d_all_max_n)))

    def barrier(self, ctx):
        self._check(self._lib.spiht_comm_barrier(ctx.handle, self.handle))

    def max_over_ranks(self, ctx, value):
        v = C.c_double(float(value))
        self._check(self._lib.spiht_comm_allreduce_max_f64(ctx.handle, self.handle, C.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "handle", None):
            self._lib.spiht_comm_destroy(self.handle)
            self.handle = None


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
