"""`python bench.py --gpus N` from a plain shell must become its own launcher (the round-1 version exited unless it had
been started by torch.distributed.run).  CPU check at world size 2: bench.self_launch (torch.distributed.run on 127.0.0.1),
the gloo rendezvous, the row partition, the ghost exchange and the two all-reduces of the sharded Lanczos sweep run with
the NumPy checker backend of the test-suite through tests/bench_checker.py (bench.py itself carries no stand-in engine);
the line it prints is marked data = "checker" (never a measurement)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "KK_BENCH_SPAWNED")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "bench_checker.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_shell_launch_world2_checker():
    out2 = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--ny", "8", "--orth", "mgs2"])
    assert out2["n_gpus"] == 2 and out2["data"] == "checker" and out2["scaling"] == "weak"
    assert np.isfinite(out2["last_alpha"]) and out2["last_beta"] > 0
    # the same launcher contract the driver uses (pre-launched ranks) gives the same numbers
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "tests" / "bench_checker.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--ny", "8", "--orth", "mgs2"], capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    pre = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert pre["n_gpus"] == 2 and abs(pre["last_alpha"] - out2["last_alpha"]) <= 1e-12 * abs(pre["last_alpha"])


def test_world1_checker_and_mismatch_guard():
    out1 = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--ny", "8"])
    assert out1["n_gpus"] == 1 and out1["data"] == "checker"
    # WORLD_SIZE that contradicts --gpus is refused (the driver's launch line always agrees)
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    for script in (ROOT / "bench.py", ROOT / "tests" / "bench_checker.py"):     # bench.py's own guard (before it touches a device) and its twin
        p = subprocess.run([sys.executable, str(script), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
        assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_py_carries_no_stand_in_engine():
    src = (ROOT / "bench.py").read_text()
    assert "CheckerBackend" not in src and "--backend" not in src and "dist_checker_backend" not in src
