"""Self-test of the test-only RCCL stand-in (tests/fake_rccl/fake_rccl.cpp) in its host-memory mode: 2 and 3 processes,
all-reduce (sum / max / min, chunked), all-gather, reduce-scatter, grouped and ungrouped send / recv, and the bounded
wait when a peer is missing.  Runs without a GPU; the GPU tests that put libkrylov_hip's native multi-rank path on top
of this shim are tests/test_gpu_world2.py."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
LIB = HERE / "fake_rccl" / "libfake_rccl.so"


def _build():
    subprocess.run(["make", "-s", "-C", str(HERE / "fake_rccl")], check=True)


@pytest.mark.parametrize("world", [2, 3])
def test_fake_rccl_collectives_and_p2p(tmp_path, world):
    _build()
    env = dict(os.environ, KK_FAKE_RCCL_HOSTMEM="1", KK_FAKE_RCCL_TIMEOUT="60", KK_FAKE_RCCL_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(HERE / "fake_rccl_selftest_worker.py"), str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=180)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} OK" in o, o
    assert not list(tmp_path.glob("kkfake_rccl_*")), "the rendezvous file must be unlinked once every rank is attached"


def test_fake_rccl_missing_peer_times_out(tmp_path):
    """a communicator whose second rank never shows up fails after the bounded wait instead of hanging the box"""
    _build()
    code = f"""
import ctypes as C, sys
lib = C.CDLL({str(LIB)!r})
class UID(C.Structure): _fields_ = [("internal", C.c_char * 128)]
lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
uid = UID(); assert lib.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
sys.exit(0 if lib.ncclCommInitRank(C.byref(comm), 2, uid, 0) == 2 else 1)
"""
    env = dict(os.environ, KK_FAKE_RCCL_HOSTMEM="1", KK_FAKE_RCCL_TIMEOUT="1", KK_FAKE_RCCL_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "only 1 of 2 ranks attached" in r.stderr


@pytest.mark.gpu
def test_fake_rccl_async_mode_really_is_asynchronous(tmp_path):
    """KK_FAKE_RCCL_ASYNC=1 on cuda:0, two ranks, 0.4 s artificial delay: the call returns at once, the receive buffer still
    holds its old contents when read through another stream, the bit-identical result is there after the stream sync"""
    _build()
    env = dict(os.environ, KK_FAKE_RCCL_ASYNC="1", KK_FAKE_RCCL_DELAY_US="400000", KK_FAKE_RCCL_TIMEOUT="60", KK_FAKE_RCCL_DIR=str(tmp_path))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, str(HERE / "fake_rccl_async_worker.py"), str(r), "2", str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=180)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} OK" in o, o
