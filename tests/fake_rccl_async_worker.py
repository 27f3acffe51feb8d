"""Worker of tests/test_fake_rccl.py::test_fake_rccl_async_mode_really_is_asynchronous: one rank of a world-2 communicator
of the stand-in on cuda:0 with KK_FAKE_RCCL_ASYNC=1 and a long artificial delay.  The all-reduce call must return long
before the result exists (the receive buffer still holds its old contents when read through another stream), and the result
must be there -- bit-identical on both ranks -- once the collective's stream has been synchronised.  TEST INFRASTRUCTURE."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

rank, world, rdv = int(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3])
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
lib = C.CDLL(str(Path(__file__).parent / "fake_rccl" / "libfake_rccl.so"))
vp = C.c_void_p


class UID(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, UID, C.c_int]
lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclCommDestroy.argtypes = [vp]
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipStreamSynchronize.argtypes = [vp]
H2D, D2H = 1, 2
ck = lambda e: (_ for _ in ()).throw(RuntimeError(f"hip error {e}")) if e else None

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
ck(hip.hipSetDevice(0))
comm = vp()
assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
stream = vp()
ck(hip.hipStreamCreateWithFlags(C.byref(stream), 1))      # hipStreamNonBlocking: the null stream does not wait for it
n = 1000
x = np.random.default_rng([7, rank]).standard_normal(n)
old = np.full(n, -7.0)
dx, dy = vp(), vp()
ck(hip.hipMalloc(C.byref(dx), n * 8)); ck(hip.hipMalloc(C.byref(dy), n * 8))
ck(hip.hipMemcpy(dx, x.ctypes.data_as(vp), n * 8, H2D)); ck(hip.hipMemcpy(dy, old.ctypes.data_as(vp), n * 8, H2D))
t0 = time.time()
assert lib.ncclAllReduce(dx, dy, n, 8, 0, comm, stream) == 0
t_call = time.time() - t0
peek = np.empty(n)
ck(hip.hipMemcpy(peek.ctypes.data_as(vp), dy, n * 8, D2H))  # through the null stream: NOT ordered behind the collective
assert t_call < 0.15, f"the call took {t_call:.3f} s: it waited for the exchange"
assert np.array_equal(peek, old), "the result was there before the collective's stream was synchronised"
ck(hip.hipStreamSynchronize(stream))
got = np.empty(n)
ck(hip.hipMemcpy(got.ctypes.data_as(vp), dy, n * 8, D2H))
ref = np.random.default_rng([7, 0]).standard_normal(n)
for r in range(1, world):
    ref = ref + np.random.default_rng([7, r]).standard_normal(n)
assert np.array_equal(got, ref)
# a grouped exchange the same way
sb, rb = vp(), vp()
ck(hip.hipMalloc(C.byref(sb), n * 8)); ck(hip.hipMalloc(C.byref(rb), n * 8))
ck(hip.hipMemcpy(sb, x.ctypes.data_as(vp), n * 8, H2D)); ck(hip.hipMemcpy(rb, old.ctypes.data_as(vp), n * 8, H2D))
peer = 1 - rank
assert lib.ncclGroupStart() == 0
assert lib.ncclSend(sb, n, 8, peer, comm, stream) == 0 and lib.ncclRecv(rb, n, 8, peer, comm, stream) == 0
assert lib.ncclGroupEnd() == 0
ck(hip.hipMemcpy(peek.ctypes.data_as(vp), rb, n * 8, D2H))
assert np.array_equal(peek, old)
ck(hip.hipStreamSynchronize(stream))
ck(hip.hipMemcpy(got.ctypes.data_as(vp), rb, n * 8, D2H))
assert np.array_equal(got, np.random.default_rng([7, peer]).standard_normal(n))
assert lib.ncclCommDestroy(comm) == 0
print(f"rank {rank} OK", flush=True)
