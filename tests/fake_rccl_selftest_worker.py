"""Worker of tests/test_fake_rccl.py: one rank of a world-N communicator of tests/fake_rccl/libfake_rccl.so in its
host-memory mode (KK_FAKE_RCCL_HOSTMEM=1: "device" pointers are NumPy buffers).  TEST INFRASTRUCTURE."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

rank, world, rdv = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3])
lib = C.CDLL(str(Path(__file__).parent / "fake_rccl" / "libfake_rccl.so"))


class UID(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


vp = C.c_void_p
lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, UID, C.c_int]
lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
lib.ncclReduceScatter.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclCommDestroy.argtypes = [vp]
F64, I64, SUM, MAX, MIN = 8, 4, 0, 2, 3

uid = UID()
if rank == 0:
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    (rdv / "id.tmp").write_bytes(bytes(uid.internal).ljust(128, b"\0"))
    (rdv / "id.tmp").rename(rdv / "id")
else:
    t0 = time.time()
    while not (rdv / "id").exists():
        assert time.time() - t0 < 60
        time.sleep(0.01)
    uid.internal = (rdv / "id").read_bytes().rstrip(b"\0")
comm = vp()
assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
p = lambda a: a.ctypes.data_as(vp)


def data(r, n, seed=0):
    return np.random.default_rng([seed, r]).standard_normal(n)


# all-reduce: sum (in place, chunked: 3 000 000 doubles = 24 MB > the 16 MB slot), max, min, int64 sum
for n in (1, 7, 201, 3_000_000):
    x = data(rank, n)
    assert lib.ncclAllReduce(p(x), p(x), n, F64, SUM, comm, None) == 0
    ref = data(0, n)
    for r in range(1, world):
        ref = ref + data(r, n)
    assert np.array_equal(x, ref), ("allreduce sum", n)       # rank order 0..world-1: bitwise
x = data(rank, 33); y = np.empty(33)
assert lib.ncclAllReduce(p(x), p(y), 33, F64, MAX, comm, None) == 0
assert np.array_equal(y, np.max([data(r, 33) for r in range(world)], axis=0))
assert lib.ncclAllReduce(p(x), p(y), 33, F64, MIN, comm, None) == 0
assert np.array_equal(y, np.min([data(r, 33) for r in range(world)], axis=0))
xi = np.arange(5, dtype=np.int64) * (rank + 1)
assert lib.ncclAllReduce(p(xi), p(xi), 5, I64, SUM, comm, None) == 0
assert np.array_equal(xi, np.arange(5) * sum(r + 1 for r in range(world)))
# all-gather (chunked as well) and reduce-scatter
for n in (3, 2_500_000):
    x = data(rank, n, 1); out = np.empty(n * world)
    assert lib.ncclAllGather(p(x), p(out), n, F64, comm, None) == 0
    assert np.array_equal(out, np.concatenate([data(r, n, 1) for r in range(world)]))
for n in (5, 1_500_000):
    x = data(rank, n * world, 2); out = np.empty(n)
    assert lib.ncclReduceScatter(p(x), p(out), n, F64, SUM, comm, None) == 0
    ref = data(0, n * world, 2)[rank * n:(rank + 1) * n].copy()
    for r in range(1, world):
        ref = ref + data(r, n * world, 2)[rank * n:(rank + 1) * n]
    assert np.array_equal(out, ref), ("reducescatter", n)
# grouped point-to-point: every rank sends 3 messages of different sizes (one of them larger than the 1 MB mailbox) to
# every other rank and receives theirs, all in ONE group -- mutually dependent, several messages per peer
sizes = (4, 300_000, 17)
send = {(q, k): data(rank * 100 + q, sizes[k], 3 + k) for q in range(world) if q != rank for k in range(3)}
recv = {(q, k): np.empty(sizes[k]) for q in range(world) if q != rank for k in range(3)}
assert lib.ncclGroupStart() == 0
for k in range(3):
    for q in range(world):
        if q == rank:
            continue
        assert lib.ncclSend(p(send[(q, k)]), sizes[k], F64, q, comm, None) == 0
        assert lib.ncclRecv(p(recv[(q, k)]), sizes[k], F64, q, comm, None) == 0
assert lib.ncclGroupEnd() == 0
for (q, k), buf in recv.items():
    assert np.array_equal(buf, data(q * 100 + rank, sizes[k], 3 + k)), ("p2p", q, k)
# ungrouped pair: even ranks send first, odd ranks receive first (NCCL's blocking order outside groups)
if world >= 2 and rank < 2:
    a, b = data(rank, 9, 9), np.empty(9)
    if rank == 0:
        assert lib.ncclSend(p(a), 9, F64, 1, comm, None) == 0 and lib.ncclRecv(p(b), 9, F64, 1, comm, None) == 0
    else:
        assert lib.ncclRecv(p(b), 9, F64, 0, comm, None) == 0 and lib.ncclSend(p(a), 9, F64, 0, comm, None) == 0
    assert np.array_equal(b, data(1 - rank, 9, 9))
assert lib.ncclCommDestroy(comm) == 0
print(f"rank {rank} OK", flush=True)
