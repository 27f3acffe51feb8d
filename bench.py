#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: Lanczos iterations/s (+ achieved HBM GB/s) for
eigsolve(Lanczos) on the 10M-row 5-point Laplacian (SparseMatrixCSC), krylovdim = 100.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU.  Launched either by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
  (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or from a plain shell -- then bench.py re-executes
  itself under torch.distributed.run on 127.0.0.1.  The data path has no torch in it: libkrylov_hip owns the RCCL
  communicator (kk_comm_init) and issues its two all-reduces + one ghost exchange per iteration itself; torch.distributed
  (gloo) only carries the 128-byte communicator id and the barriers around the timed region.

One *step* = one full Krylov sweep of the hot path: `initialize` + 99 `expand!` calls
(basis size m = 2..100), i.e. 99 Lanczos iterations (1 iteration = 1 expand! = 1 operator
application, the reference's `numops` unit).  value = 99*K / elapsed  [iterations/s], inputs
resident in HBM before the timed region.  N > 1, `--scaling weak` (default): every rank owns 10M rows of a
(4000 x 2500*N)-grid Laplacian, basis row-sharded, 2 RCCL all-reduces + 1 halo exchange per
iteration; value is the whole-job aggregate: every rank processes its 10M-row shard of each
iteration, so value = N * (job iterations / s) in units of 10M-row Lanczos iterations per second
(identical to plain iterations/s at N = 1; "job_iterations_per_second" is also reported).
`--scaling strong`: the ONE 10M-row problem split over the N ranks, value = job iterations/s.  The mode not chosen runs
as a secondary leg ("other_scaling_leg").

Extra objects: "roofline" (dominant kernel, HIP events recorded on the kernels' stream inside
the timed region), "cpu_baseline" (the C twin of the oracle timed on the host cores, rank 0,
N = 1 only: the FULL 10M-row sweep from the same start vector) and "parity" (alpha / beta trajectories and Ritz
values of CPU runs against GPU sweeps for three start vectors, north_star's 1e-10 bar at the headline size), plus the
OTHER execution order of MGS2 as a secondary leg: the headline runs the library default (`mgs_mode` auto: the reference's
sequential order, src/orthonormal.jl:414-439, through the persistent kernel at N = 1 -> leg "mgs2_lowsync"; the
low-synchronisation form on a sharded context -> leg "mgs2_strict"), and "without_event_bracketing" (the same sweeps with
no HIP event in the timed region).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))

NX, NY = 4000, 2500          # configs[1]: 10M-row 5-point Laplacian
KRYLOVDIM = 100
HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)


def laplacian_rows(nx: int, ny_total: int, y0: int, y1: int) -> sp.csr_matrix:
    """Rows [y0*nx, y1*nx) of the 5-point Dirichlet Laplacian on an nx x ny_total grid
    (diag 4, off-diag -1), global column indices.  SURVEY.md 8(d) cfg 2."""
    n_glob = nx * ny_total
    r = np.arange(y0 * nx, y1 * nx, dtype=np.int64)
    ix = r % nx
    cols = [r, r - 1, r + 1, r - nx, r + nx]
    vals = [np.full(r.size, 4.0)] + [np.full(r.size, -1.0)] * 4
    ok = [np.ones(r.size, bool), ix > 0, ix < nx - 1, r - nx >= 0, r + nx < n_glob]
    rows = np.concatenate([(r - y0 * nx)[m] for m in ok])
    cc = np.concatenate([c[m] for c, m in zip(cols, ok)])
    vv = np.concatenate([v[m] for v, m in zip(vals, ok)])
    return sp.csr_matrix((vv, (rows, cc)), shape=(r.size, n_glob))


def algorithmic_bytes_sweep(n_rows: int, krylovdim: int) -> float:
    """(176 + 16 m) N bytes per expand at basis size m, m = 2..krylovdim."""
    return float(sum((176 + 16 * m) * n_rows for m in range(2, krylovdim + 1)))


def usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a container
    can see 256 hardware threads in os.cpu_count() and own 8 of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, min(n, 64))


def _load_cpu_ref():
    lib_path = ROOT / "oracle" / "libcpu_ref.so"
    if not lib_path.exists():
        return None
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    lib = C.CDLL(str(lib_path))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    lib.kkref_lanczos.argtypes = [C.c_int64, ip, ip, dp, dp, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp,
                                  C.POINTER(C.c_int), dp]
    lib.kkref_lanczos.restype = C.c_int
    return lib


def _run_cpu_ref(lib, ny: int, x0: np.ndarray, orth_code: int, cores: int):
    """initialize + 99 expand! of oracle/cpu_ref.c on the NX x ny grid from the start vector x0 -> (seconds, alphas, betas)"""
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    steps = KRYLOVDIM - 1
    n = NX * ny
    A = laplacian_rows(NX, ny, 0, ny).tocsc()
    A.sort_indices()
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1   # Julia SparseMatrixCSC{Float64,Int64}
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nz = np.ascontiguousarray(A.data)
    x0 = np.ascontiguousarray(x0[:n], dtype=np.float64)
    al, be = np.zeros(steps + 1), np.zeros(steps + 1)
    passes = C.c_int()
    t0 = time.perf_counter()
    rc = lib.kkref_lanczos(n, colptr.ctypes.data_as(ip), rowval.ctypes.data_as(ip), nz.ctypes.data_as(dp),
                           x0.ctypes.data_as(dp), steps, orth_code, 0.0, cores, al.ctypes.data_as(dp),
                           be.ctypes.data_as(dp), C.byref(passes), None)
    dt = time.perf_counter() - t0
    return (dt, al, be) if rc == 0 else (None, None, None)


def cpu_baseline(orth_code: int, x0_full: np.ndarray, ny_full: int, max_full_seconds: float = 60.0):
    """Time oracle/libcpu_ref.so (C twin of the oracle = the reference's un-fused CPU path) on the host cores.
    A 200 000-row calibration run predicts the cost of the FULL workload (the path is linear in N); if that fits
    `max_full_seconds` the full NX x ny_full sweep is run from the GPU run's own start vector -- the measured rate is
    then un-scaled and its (alpha, beta) are the reference trajectory of the "parity" block.  Otherwise (slow host) a
    bounded sample is timed and scaled, and parity stays unmeasured.  Returns (baseline dict, alphas, betas)."""
    lib = _load_cpu_ref()
    if lib is None:
        return None, None, None
    cores = usable_cores()
    steps = KRYLOVDIM - 1
    cal_ny = 50
    dt, _, _ = _run_cpu_ref(lib, cal_ny, x0_full, orth_code, cores)   # 200 000 rows (basis of 100 vectors = 160 MB, out of cache)
    if dt is None:
        return None, None, None
    if cores > 8 and dt > 5.0:           # oversubscribed container (visible cores != usable cores): retry narrow
        dt8, _, _ = _run_cpu_ref(lib, cal_ny, x0_full, orth_code, 8)
        if dt8 is not None and dt8 < dt:
            cores, dt = 8, dt8
    predicted_full = dt * ny_full / cal_ny
    n_full = NX * ny_full
    if predicted_full <= max_full_seconds:
        dtf, al, be = _run_cpu_ref(lib, ny_full, x0_full, orth_code, cores)
        if dtf is not None:
            return ({
                "value": round(steps / dtf, 4), "unit": "it/s", "cores": cores, "kind": "port",
                "sample": f"the full workload: {steps}-expand sweep (initialize included) on the {NX}x{ny_full} grid = {n_full} rows "
                          f"from the GPU run's start vector, {dtf:.2f} s on {cores} threads, no scaling; oracle/cpu_ref.c: un-fused "
                          "BLAS-1 passes (OpenMP) + serial Int64 CSC SpMV as the reference issues them",
                "hbm_equiv_GBps": round(algorithmic_bytes_sweep(n_full, KRYLOVDIM) / dtf / 1e9, 2),
            }, al, be)
    ny = int(min(ny_full, max(cal_ny, cal_ny * 20.0 / max(dt, 1e-3))))
    dts, _, _ = _run_cpu_ref(lib, ny, x0_full, orth_code, cores)
    if dts is None:
        return None, None, None
    n = NX * ny
    scale = n / float(n_full)
    return ({
        "value": round(steps / dts * scale, 4), "unit": "it/s", "cores": cores, "kind": "port",
        "sample": f"full {steps}-expand sweep on a {NX}x{ny} grid = {n} rows ({dts:.2f} s on {cores} threads; the full workload was "
                  f"predicted at {predicted_full:.0f} s > {max_full_seconds:.0f} s); rate scaled by {scale:g} (the path is linear in N)",
        "hbm_equiv_GBps": round(algorithmic_bytes_sweep(n, KRYLOVDIM) / dts / 1e9, 2),
    }, None, None)


def parity_block(al_g, be_g, al_c, be_c):
    """north_star: results match the reference CPU path within 1e-10 relative -- (alpha, beta) trajectories of the 99
    expand! calls and the Ritz values of the 100 x 100 tridiagonal, GPU sweep vs oracle/cpu_ref.c, same start vector."""
    al_g, be_g, al_c, be_c = (np.asarray(v, dtype=np.float64) for v in (al_g, be_g, al_c, be_c))
    tri = lambda a, b: np.linalg.eigvalsh(np.diag(a) + np.diag(b[:-1], 1) + np.diag(b[:-1], -1))
    th_g, th_c = tri(al_g, be_g), tri(al_c, be_c)
    out = {
        "alpha_relerr": float(np.max(np.abs(al_g - al_c) / np.abs(al_c))),
        "beta_relerr": float(np.max(np.abs(be_g - be_c) / np.abs(be_c))),
        "ritz_relerr": float(np.max(np.abs(th_g - th_c) / np.abs(th_c))),
        "ritz_abserr_over_norm": float(np.max(np.abs(th_g - th_c)) / np.max(np.abs(th_c))),
        "tol": 1e-10, "n_steps": int(len(al_g)),
        "reference": "oracle/cpu_ref.c (restatement of src/factorizations/lanczos.jl:180-376), full-size run",
    }
    out["ok"] = bool(max(out["alpha_relerr"], out["beta_relerr"], out["ritz_relerr"]) <= out["tol"])
    return out


def kernel_source_sha() -> str:
    """hash of every device / host source of the library: PMC traffic figures are only reported for the sources they were
    measured on (tools/profile_gpu.sh stamps them with this)"""
    import hashlib
    h = hashlib.sha256()
    src = ROOT / "krylovkit.jl_amd" / "csrc"
    for f in sorted(src.glob("*.hip")) + sorted(src.glob("*.h")) + [src.parent / "Makefile"]:   # (the Makefile carries per-file code-generation flags)
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def stamped_traffic(fname: str):
    """profiles/<fname> (written by tools/profile_gpu.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) if it
    was measured on the current sources -> (dict, note); (None, note) otherwise"""
    tf = ROOT / "profiles" / fname
    if not tf.exists():
        return None, f"profiles/{fname} absent: no counter pass on these sources yet"
    try:
        tj = json.loads(tf.read_text())
    except Exception:
        return None, f"profiles/{fname} unreadable"
    if tj.get("source_sha") != kernel_source_sha():
        return None, f"profiles/{fname} was measured on other kernel sources (hash mismatch): stale, not reported"
    return tj, f"PMC passes of {tj.get('stamped_by', 'tools/profile_gpu.sh')} on the same kernel sources (sha {tj.get('source_sha')})"


def read_ceiling():
    """the box's own attainable HBM read rate, measured now (tools/hbm_peak.hip --json, ~1 s in a child process before anything
    of the bench is resident): best non-temporal and best plain 16-byte read stream over 2 GiB.  None when the tool is not
    built (build() of __graft_entry__.py compiles it)."""
    import subprocess
    exe = ROOT / "tools" / "bin" / "hbm_peak"
    if not exe.exists():
        return None
    try:
        r = subprocess.run([str(exe), "--json"], capture_output=True, text=True, timeout=60)
        ln = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(ln[-1]) if r.returncode == 0 and ln else None
    except Exception:
        return None


def read_gather_ceiling(table_doubles: int = 393216):
    """the box's own RANDOM-GATHER rate for an L2-resident table with a value stream (tools/gather_rate.hip --json, the column tile of the
    tiled SELL format as the table): the ceiling of config 4's apply, which issues one L2 request per stored entry (DESIGN section 3,
    profiles/r03_gather_rate_ceiling.json).  None when the tool is not built."""
    import subprocess
    exe = ROOT / "tools" / "bin" / "gather_rate"
    if not exe.exists():
        return None
    try:
        r = subprocess.run([str(exe), "--json", str(table_doubles)], capture_output=True, text=True, timeout=120)
        ln = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(ln[-1]) if r.returncode == 0 and ln else None
    except Exception:
        return None


def gather_roofline(leg: dict, nnz: int, applies_per_sweep: int, ceiling):
    """config 4: next to the HBM fraction (which the gather-bound apply can never move) the fraction of the box's gather ceiling"""
    roof = leg.get("roofline") or {}
    k = (leg.get("kernels") or {}).get("k_spmv_sell")
    if not roof or not k or not k.get("ms_per_sweep"):
        return
    ach = nnz * applies_per_sweep / (k["ms_per_sweep"] * 1e-3) / 1e9
    roof["achieved_Ggathers_per_s"] = round(ach, 1)
    if ceiling and ceiling.get("Ggathers_per_s"):
        roof["gather_ceiling_Ggathers_per_s"] = ceiling["Ggathers_per_s"]
        roof["frac_of_gather_ceiling"] = round(ach / ceiling["Ggathers_per_s"], 4)
        roof["gather_ceiling"] = {k_: ceiling[k_] for k_ in ("table_bytes", "in_flight", "grid", "value_stream") if k_ in ceiling}
    roof["bound_is"] = "gather rate (one L2 request per stored entry), not HBM: `frac` cannot approach 1; `frac_of_gather_ceiling` is the figure to read"


def physical_roofline(kernel: str, seconds: float, launches: int, traffic_bytes, model_bytes, alg_bytes, min_bytes=None, note=None):
    """roofline object of one kernel class over a timed region: `achieved` = bytes the kernel really moved (counter traffic
    when a stamped PMC pass exists, the byte model of DESIGN.md otherwise) / its event-timed duration; `frac` = achieved /
    8 TB/s.  The SURVEY 8(d) contract figure (ALGORITHMIC bytes, which a kernel that keeps data on chip undercuts) is
    carried as `algorithmic_equiv_*`, the least any kernel of that shape could move as `min_bytes` / `frac_of_min`."""
    phys = traffic_bytes if traffic_bytes is not None else model_bytes
    out = {"kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "launches": int(launches),
           "avg_launch_ms": round(seconds / max(launches, 1) * 1e3, 5)}
    if phys is not None:
        out["achieved"] = round(phys / seconds / 1e9, 1)
        out["frac"] = round(phys / seconds / 1e9 / HBM_PEAK_GBPS, 4)
    else:
        out["achieved"] = out["frac"] = None
    out["traffic"] = None if traffic_bytes is None else round(traffic_bytes / max(launches, 1))
    out["traffic_source"] = "pmc" if traffic_bytes is not None else ("model" if model_bytes is not None else None)
    if note:
        out["traffic_note"] = note
    if model_bytes is not None:
        out["hbm_model_bytes_per_launch"] = round(model_bytes / max(launches, 1))
        out["hbm_model_frac"] = round(model_bytes / seconds / 1e9 / HBM_PEAK_GBPS, 4)
    if alg_bytes is not None:
        out["algorithmic_bytes_per_launch"] = round(alg_bytes / max(launches, 1))
        out["algorithmic_equiv_GBps"] = round(alg_bytes / seconds / 1e9, 1)
        out["algorithmic_equiv_frac"] = round(alg_bytes / seconds / 1e9 / HBM_PEAK_GBPS, 4)
    if min_bytes is not None:
        out["min_bytes_per_launch"] = round(min_bytes / max(launches, 1))
        out["frac_of_min"] = round(min_bytes / seconds / 1e9 / HBM_PEAK_GBPS, 4)
    return out


def convdiff_rows(nx: int, ny: int, px: float = 0.5, py: float = 0.25) -> sp.csr_matrix:
    """SURVEY.md 8(d) cfg 3: 2-D convection-diffusion, 5-point central differences, cell Peclet numbers px / py (nonsymmetric)"""
    n = nx * ny
    r = np.arange(n, dtype=np.int64)
    ix = r % nx
    cols = [r, r - 1, r + 1, r - nx, r + nx]
    vals = [np.full(n, 4.0), np.full(n, -(1 + px)), np.full(n, -(1 - px)), np.full(n, -(1 + py)), np.full(n, -(1 - py))]
    ok = [np.ones(n, bool), ix > 0, ix < nx - 1, r - nx >= 0, r + nx < n]
    rows = np.concatenate([r[m] for m in ok]); cc = np.concatenate([c[m] for c, m in zip(cols, ok)])
    vv = np.concatenate([v[m] for v, m in zip(vals, ok)])
    return sp.csr_matrix((vv, (rows, cc)), shape=(n, n))


LEG_ROUNDS = 3
LEG_CLASSES = ("k_project", "k_unproject", "k_unproj_proj", "k_mgs_persist", "k_mgs_panel", "k_mgs_step", "k_spmv_ell", "k_spmv_dia", "k_spmv_sell",
               "k_spmv_csr", "k_scal", "k_dot", "k_axpby", "k_block_gram", "k_block_update", "k_spmm_dia", "k_spmm_ell")


def run_leg(ctx, name: str, sweep, units_per_sweep: int, K: int, model_bytes_per_sweep: dict, alg_bytes_per_sweep: float, meta: dict):
    """One secondary configuration inside the default line: a warm-up sweep, K event-profiled timed sweeps, the rate, and a
    roofline object for the kernel class that took the most time (physical bytes: stamped PMC traffic of this configuration
    when there is one, the byte model otherwise)."""
    sweep()
    ctx.sync()
    rounds = []
    for _round in range(LEG_ROUNDS):         # the rate: K sweeps with no event anywhere, LEG_ROUNDS times over; the MEDIAN round is reported, the
        t0 = time.perf_counter()             # fastest and the slowest next to it (host-synchronisation-heavy steps -- GKL: four host round trips
        for _ in range(K):                   # per expand! -- jitter with whatever else the box is doing; rounds 3-4 reported the best of two)
            fact = sweep()
        ctx.sync()
        rounds.append(time.perf_counter() - t0)
    dt = sorted(rounds)[len(rounds) // 2]
    ctx.prof_reset(); ctx.prof_enable(1)     # the same K sweeps once more, every kernel class bracketed by HIP events
    for _ in range(K):
        sweep()
    ctx.sync()
    ctx.prof_enable(0)
    classes = {}
    for c in LEG_CLASSES:
        ms, n = ctx.prof_get(c)
        if n:
            classes[c] = (ms, n)
    tj, note = stamped_traffic("traffic_configs.json")
    tcfg = (tj or {}).get("configs", {}).get(name, {}) if tj else {}
    table = {}
    for c, (ms, n) in sorted(classes.items(), key=lambda kv: -kv[1][0]):
        tb = tcfg.get(c)          # bytes per sweep from the counters
        mb = model_bytes_per_sweep.get(c)
        phys = tb if tb is not None else mb
        table[c] = {"ms_per_sweep": round(ms / K, 4), "launches_per_sweep": round(n / K, 2),
                    "bytes_per_sweep": None if phys is None else round(phys), "bytes_source": "pmc" if tb is not None else ("model" if mb is not None else None),
                    "frac": None if phys is None else round(phys * K / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    dom = max(classes, key=lambda c: classes[c][0]) if classes else None
    roof = None
    if dom:
        ms, n = classes[dom]
        tb, mb = tcfg.get(dom), model_bytes_per_sweep.get(dom)
        roof = physical_roofline(dom, ms * 1e-3, n, None if tb is None else tb * K, None if mb is None else mb * K, None, note=note)
    kernel_ms = sum(v[0] for v in classes.values()) / K
    out = {"value": round(units_per_sweep * K / dt, 2), "ms_per_step": round(dt / K * 1e3, 3), "steps": K,
           "rate_is": f"median of {LEG_ROUNDS} rounds of {K} sweeps", "value_min": round(units_per_sweep * K / max(rounds), 2),
           "value_max": round(units_per_sweep * K / min(rounds), 2),
           "kernel_ms_per_sweep": round(kernel_ms, 3),
           "algorithmic_equiv_frac": round(alg_bytes_per_sweep * K / dt / 1e9 / HBM_PEAK_GBPS, 4),
           "roofline": roof, "kernels": table}
    out.update(meta)
    return out, fact


def self_launch(args, script=None) -> int:
    """`python bench.py --gpus N` from a plain shell: run N ranks under torch.distributed.run on this node.  (`script`: the
    CPU test of this launcher, tests/bench_checker.py, re-launches itself through the same code.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(script or __file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, KK_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ workloads
def gkl_rows(m_total: int, n: int, per: int, r0: int, r1: int):
    """Rows [r0, r1) of the config-4 map: 5M x 1M sparse random, 20 entries per row (duplicates summed).  Generated in
    blocks of 625 000 rows with per-block seeds so that the global matrix does not depend on the number of ranks."""
    blk = 625_000 if m_total >= 625_000 else m_total
    parts = []
    for b0 in range(r0 // blk * blk, r1, blk):
        b1 = min(b0 + blk, m_total)
        rng = np.random.default_rng([5, b0 // blk])
        cols = rng.integers(0, n, size=(b1 - b0) * per, dtype=np.int32)
        vals = rng.standard_normal((b1 - b0) * per)
        A = sp.csr_matrix((vals, cols, np.arange(0, (b1 - b0) * per + 1, per, dtype=np.int64)), shape=(b1 - b0, n))
        A.sum_duplicates()
        lo, hi = max(r0, b0) - b0, min(r1, b1) - b0
        parts.append(A[lo:hi])
    return sp.vstack(parts, format="csr") if len(parts) > 1 else parts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--orth", default=os.environ.get("KK_BENCH_ORTH", "mgs2"), choices=["cgs2", "mgs2"])
    ap.add_argument("--mgs-mode", default="auto", choices=["auto", "lowsync", "strict"],
                    help="how the MGS family runs: auto (library default: the reference's sequential order, src/orthonormal.jl:414-439, "
                         "through the persistent kernel where that is faster, the low-synchronisation form otherwise), or one of the two forced")
    ap.add_argument("--config", default="lanczos", choices=["lanczos", "gkl", "block"],
                    help="lanczos = BASELINE.json configs[1] (the judged line); gkl / block = configs[3] / configs[4], row-sharded")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="config lanczos, N > 1: weak = --ny grid lines (10M rows) PER GPU, the north_star reading; strong = the ONE 10M-row "
                         "problem of BASELINE.json's metric split over the N GPUs.  The other mode runs as a secondary leg and is reported "
                         "next to the headline value (N = 1: both coincide)")
    ap.add_argument("--no-other-scaling-leg", action="store_true")
    ap.add_argument("--parity-seeds", type=int, default=3, help="start vectors of the full-size parity block (each costs one CPU sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict-leg", action="store_true")
    ap.add_argument("--no-sharded-leg", action="store_true",
                    help="config lanczos, N = 1: skip the `sharded_world1` leg (the same sweep on a row-sharded context of world size 1 with real RCCL "
                         "collectives issued: the like-for-like single-GPU base of the N > 1 lines)")
    ap.add_argument("--no-configs", action="store_true",
                    help="config lanczos, N = 1: skip the `configs` block (BASELINE.json configs[2..4] at full size + the general-format leg of configs[1])")
    ap.add_argument("--config-steps", type=int, default=3, help="timed sweeps per entry of the `configs` block")
    ap.add_argument("--only-leg", default=None, choices=["lanczos_ell", "gmres", "block", "gkl"],
                    help="run nothing but the sweeps of ONE entry of the `configs` block (tools/profile_gpu.sh: counter passes per configuration)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="kk_ctx_set_option before the workload is built (A/B runs; every override is reported in the line as `options_overridden`)")
    ap.add_argument("--ny", type=int, default=NY, help="grid rows per GPU (default 2500 -> 10M rows per GPU)")
    ap.add_argument("--deadline", type=float, default=900.0, help="multi-rank runs only: abort if the whole run takes longer (seconds)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        if os.environ.get("KK_BENCH_SPAWNED"):
            raise SystemExit("bench.py: spawned without WORLD_SIZE")
        raise SystemExit(self_launch(args))     # plain `python bench.py --gpus N`: become the launcher
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        # A multi-rank run that stops making progress (a rank died, a collective never completes) must not sit on the GPU
        # box until an outer limit kills it: every rank gives itself a deadline and leaves with a message instead.
        import threading

        def _deadline():
            sys.stderr.write(f"bench.py: rank {rank} of {world} did not finish within {args.deadline:.0f} s -- aborting the run\n")
            sys.stderr.flush()
            os._exit(3)

        wd = threading.Timer(args.deadline, _deadline)
        wd.daemon = True
        wd.start()

    ceiling = read_ceiling() if (world == 1 and not os.environ.get("KK_BENCH_FORCE_DIST") and not args.only_leg) else None
    gather_ceiling = read_gather_ceiling() if (world == 1 and not os.environ.get("KK_BENCH_FORCE_DIST") and
                                               (args.config == "gkl" or args.only_leg == "gkl" or (args.config == "lanczos" and not args.no_configs and not args.only_leg and args.ny == NY))) else None

    import krylovkit_hip as kk
    from krylovkit_hip import dist as kd

    orth = kk.Orthogonalizer(args.orth)
    K, W = args.steps, args.warmup
    force_dist = bool(os.environ.get("KK_BENCH_FORCE_DIST"))   # exercise the sharded path (RCCL issued) on 1 GPU
    use_dist = world > 1 or force_dist
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist          # control plane only (gloo): communicator id + barriers
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # one rank per GPU; if the launcher narrowed the visible devices per rank (HIP_VISIBLE_DEVICES), LOCAL_RANK wraps
    ctx = kk.Context(local_rank % max(kk.device_count(), 1)) if use_dist else kk.default_context()
    comm = None
    if use_dist:
        comm = kd.NativeComm.from_torch_distributed(ctx, force_collectives=force_dist) if world > 1 else \
            kd.NativeComm.single(ctx, force_collectives=True)
    MODE = {"strict": 0, "lowsync": 1, "auto": 2}
    ctx.set_option("mgs_mode", MODE[args.mgs_mode])
    overridden = {}
    for kv in args.opt:
        k_, v_ = kv.split("=", 1)
        ctx.set_option(k_, float(v_))
        overridden[k_] = float(v_)
    sync = ctx.sync
    barrier = dist.barrier if world > 1 else (lambda: None)

    # ------------------------------------------------------------ workload set-up (inputs resident in HBM afterwards)
    def build_lanczos(mode):
        """operator + slab + start vector of config 2 for one scaling mode -> dict(sweep, n_local, n_global, ...)"""
        if mode == "strong" and world > 1:       # the 10M-row problem itself, rows split along grid lines
            ny_tot = args.ny
            pt = kd.Partition.even(NX * ny_tot, world, rank, align=NX)
            y0, y1 = pt.lo // NX, pt.hi // NX
        else:                                     # 10M rows per GPU
            ny_tot = args.ny * world
            y0, y1 = rank * args.ny, (rank + 1) * args.ny
            pt = kd.Partition.even(NX * ny_tot, world, rank, align=NX)
        nl = NX * (y1 - y0)
        A = laplacian_rows(NX, ny_tot, y0, y1)
        if use_dist:
            op_ = kd.NativeShardedOperator(A, pt, ctx, symmetric=True)    # kk_csr_create_sharded: ghost plan negotiated inside
        else:
            op_ = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)   # handed over as Julia's SparseMatrixCSC
        del A
        V_ = kk.DeviceBasis(nl, KRYLOVDIM + 2, ctx)
        x0_ = kk.DeviceBasis(nl, 1, ctx)
        x0_[0].rand_(3 + rank)                                              # x0 = rand!(similar(A, T, n)), resident in HBM
        it_ = kk.LanczosIterator(op_, x0_[0], orth, capacity=KRYLOVDIM + 2)

        def sweep_():
            fact = kk.initialize(it_, V_)
            for _ in range(KRYLOVDIM - 1):
                fact = kk.expand_(it_, fact)
            return fact

        return dict(sweep=sweep_, n_local=nl, n_global=NX * ny_tot, ny_tot=ny_tot, x0=x0_, keep=(op_, V_, it_))

    def build_gkl(ny_arg):
        """config 4: operator + iterator of the 5M x 1M sparse random map (rows of A and of U sharded over the ranks)"""
        m_tot, n_tot, per, Kg = 5_000_000, 1_000_000, 20, 30
        if ny_arg != NY:                            # reduced size for quick checks: --ny = rows / 2000
            m_tot, n_tot = ny_arg * 2000, ny_arg * 400
        r0, r1 = rank * m_tot // world, (rank + 1) * m_tot // world
        A = gkl_rows(m_tot, n_tot, per, r0, r1)
        nnz_loc = A.nnz
        op = kd.NativeShardedRectOperator(A, n_tot, ctx) if use_dist else kk.SparseOperator(A, ctx)
        del A
        u0 = np.random.default_rng([6, rank]).random(r1 - r0)
        it = kk.GKLIterator(op, u0, orth, capacity=Kg + 2)
        its = Kg - 1

        def sweep_():
            fact = kk.initialize(it)
            for _ in range(its):
                fact = kk.expand_(it, fact)
            return fact

        nnz_tot = nnz_loc * world
        spmv1 = 12 * nnz_tot + 4 * (m_tot + n_tot + 2) + 8 * (m_tot + n_tot) * 2     # one direction, CSR accounting of SURVEY 8(d)
        alg = float(sum(2 * spmv1 + (16 + 24 + 16 * (k - 1) + 24 + 16) * n_tot + (16 + 24 + 16 * k + 24 + 16) * m_tot for k in range(2, Kg + 1)))
        return dict(sweep=sweep_, sweep_its=its, n_local=r1 - r0, alg_sweep=alg, op=op, spmv_alg_bytes=spmv1, nnz=nnz_tot, shape=(m_tot, n_tot),
                    workload=f"svdsolve(GKL) expand! sweep: {m_tot}x{n_tot} sparse random, {per} nnz/row, krylovdim={Kg}, 1 step = initialize + {its} expand!",
                    keep=(op, it))

    def build_block(ny_arg):
        """config 5: BlockLanczos bs = 16 on the 10M-row Laplacian (rows split over the ranks)"""
        bs_, Kb_ = 16, 100
        ny_tot = ny_arg                             # 10M rows in total, split over the ranks (strong scaling)
        assert ny_tot % world == 0
        nyl = ny_tot // world
        nl = NX * nyl
        A = laplacian_rows(NX, ny_tot, rank * nyl, (rank + 1) * nyl)
        if use_dist:
            part = kd.Partition.even(NX * ny_tot, world, rank, align=NX)
            op = kd.NativeShardedOperator(A, part, ctx, symmetric=True)
        else:
            op = kk.SparseOperator(A, ctx, symmetric=True)
        del A
        S = kk.DeviceBasis(nl, Kb_ + 3 * bs_, ctx)
        it = kk.BlockLanczosIterator(op, [None] * bs_, Kb_ + bs_)
        area_b = it.maxdim + bs_

        def sweep_():
            for j in range(bs_):
                S[area_b + j].rand_(100 + j + 1000 * rank)
            it.x0 = [S[area_b + j] for j in range(bs_)]
            f = it.initialize(S)
            while len(f) < Kb_:
                f = it.expand(f)
            return f

        def expands_only_():
            """seconds of the 6 block expand! alone (initialize outside the bracket)"""
            for j in range(bs_):
                S[area_b + j].rand_(100 + j + 1000 * rank)
            it.x0 = [S[area_b + j] for j in range(bs_)]
            f = it.initialize(S)
            sync()
            t_ = time.perf_counter()
            while len(f) < Kb_:
                f = it.expand(f)
            sync()
            return time.perf_counter() - t_

        its = 6                                     # block steps after initialize: 16 -> 112 basis vectors
        alg = float(sum((1856 + 16 * k) * NX * ny_tot for k in range(2 * bs_, Kb_ + bs_ + 1, bs_)))
        return dict(sweep=sweep_, sweep_its=its, n_local=nl, alg_sweep=alg, bs=bs_, Kb=Kb_, expands_only=expands_only_,
                    workload=f"BlockLanczos eigsolve expand!: {NX}x{ny_tot} 5-point Laplacian, block size {bs_}, krylovdim={Kb_}, 1 step = initialize + {its} block expand!",
                    keep=(op, S, it))

    def leg_spec(name):
        """workload of one entry of the `configs` block -> dict(sweep, units, model bytes per sweep and class, algorithmic bytes
        per sweep, meta[, kmult, cleanup]); also what `--only-leg` runs under rocprofv3 for profiles/traffic_configs.json"""
        if name == "lanczos_ell":
            # configs[1] again with the operator applied the general way: ELL gather kernel on the SparseMatrixCSC's entries, no
            # stencil recognition (neither the value-free constant-coefficient form nor the stored diagonals)
            ctx.set_option("spmv_dia", 0)
            return dict(sweep=prob["sweep"], units=KRYLOVDIM - 1, cleanup=lambda: ctx.set_option("spmv_dia", 1),
                        model={"k_spmv_ell": 84.0 * prob["n_local"] * KRYLOVDIM, "k_scal": 16.0 * prob["n_local"] * 3,
                               "k_mgs_persist": float(sum((8 * m + 24) * prob["n_local"] for m in range(2, KRYLOVDIM + 1)))},
                        alg=algorithmic_bytes_sweep(prob["n_global"], KRYLOVDIM),
                        meta={"metric": "lanczos_iterations_per_second", "unit": "it/s",
                              "workload": "configs[1] with the library's stencil recognition OFF (option spmv_dia = 0): the general SparseMatrixCSC path, "
                                          "ELL gather SpMV (12 bytes per stored entry + vectors = 84 N per apply)"})
        if name == "gmres":
            # configs[2]: linsolve(GMRES) 2M-row convection-diffusion, krylovdim 60: one Arnoldi cycle m = 2..60 (the restart
            # kernels are measured by tools/restart_bench.py)
            nxg, nyg, Kg_ = 2000, 1000, 60
            Ng = nxg * nyg
            opg = kk.SparseOperator(convdiff_rows(nxg, nyg), ctx)
            Vg = kk.DeviceBasis(Ng, Kg_ + 2, ctx)
            x0g = kk.DeviceBasis(Ng, 1, ctx); x0g[0].rand_(4)
            itg = kk.ArnoldiIterator(opg, x0g[0], orth, capacity=Kg_ + 2)

            def sweep_g():
                f_ = kk.initialize(itg, Vg)
                for _ in range(Kg_ - 1):
                    f_ = kk.expand_(itg, f_)
                return f_

            ms_ = range(2, Kg_ + 1)
            return dict(sweep=sweep_g, units=Kg_ - 1, kmult=3, keep=(opg, Vg, x0g, itg),
                        model={"k_project": float(sum((8 * m + 16) * Ng for m in ms_)), "k_unproj_proj": float(sum((8 * m + 16) * Ng for m in ms_)),
                               "k_unproject": float(sum((8 * m + 16) * Ng for m in ms_)), "k_mgs_panel": float(sum((16 * m + 16) * Ng for m in ms_)),   # (two sweeps in one launch, every basis vector read once per sweep, w read and written once)
                               "k_spmv_dia": 24.0 * Ng * Kg_, "k_scal": 16.0 * Ng * Kg_},
                        alg=float(sum((144 + 32 * m) * Ng for m in ms_)),
                        meta={"metric": "arnoldi_iterations_per_second", "unit": "it/s",
                              "workload": f"linsolve(GMRES) expand! cycle: {nxg}x{nyg} convection-diffusion ({Ng} rows), krylovdim={Kg_}, orth {orth.name}, "
                                          f"1 step = initialize + {Kg_ - 1} expand!"})
        if name == "block":
            gb = build_block(NY)     # configs[4]: BlockLanczos bs = 16, 10M rows
            return dict(sweep=gb["sweep"], units=gb["sweep_its"], keep=gb,
                        model={"k_block_update": (sum(8 * kn + 256 for kn in range(2 * gb["bs"], gb["Kb"] + gb["bs"] + 1, gb["bs"])) + 2 * 256) * float(gb["n_local"])},   # (+ the two 16 -> 16 column back-substitutions that really run: below)
                        alg=gb["alg_sweep"], meta={"metric": "block_lanczos_steps_per_second", "unit": "block steps/s (bs=16, 10M rows)", "workload": gb["workload"]})
        if name == "gkl":
            gg = build_gkl(NY)       # configs[3]: svdsolve(GKL) 5M x 1M sparse random (one GPU holds the whole map)
            return dict(sweep=gg["sweep"], units=gg["sweep_its"], keep=gg,
                        model={"k_spmv_sell": float(gg["spmv_alg_bytes"]) * (gg["sweep_its"] * 2 + 2)}, alg=gg["alg_sweep"],
                        meta={"metric": "gkl_iterations_per_second", "unit": "it/s", "workload": gg["workload"], "operator": gg["op"].info(),
                              "bound_note": "the tiled SpMV is bound by the chip's random-gather rate (one L2 request per stored entry), not by HBM: "
                                            "profiles/r03_gather_rate_ceiling.json, DESIGN.md section 3"})
        raise SystemExit(f"unknown leg {name}")

    x0_handle = None
    other_leg_mode = None
    if args.config == "lanczos":
        sweep_its = KRYLOVDIM - 1
        prob = build_lanczos(args.scaling)
        sweep, n_local, x0_handle = prob["sweep"], prob["n_local"], prob["x0"]
        scaling = args.scaling if world > 1 else "weak"
        if world > 1 and not args.no_other_scaling_leg:
            other_leg_mode = "strong" if scaling == "weak" else "weak"
        # weak: each rank advances a 10M-row shard per iteration, value = N x job iterations/s in units of 10M-row iterations;
        # strong: the job IS one 10M-row problem, value = job iterations/s
        units_per_sweep = sweep_its * world if scaling == "weak" else sweep_its
        alg_sweep = algorithmic_bytes_sweep(prob["n_global"], KRYLOVDIM)
        metric = "lanczos_iterations_per_second"
        unit = "it/s (10M-row Lanczos iterations, summed over GPUs)" if scaling == "weak" else "it/s (Lanczos iterations of the one 10M-row problem)"
        workload = (f"eigsolve(Lanczos) expand! sweep: {NX}x{prob['ny_tot']} 5-point Laplacian ({prob['n_global']} rows, "
                    f"SparseMatrixCSC handed over via kk_csc_create), krylovdim={KRYLOVDIM}, 1 step = initialize + {sweep_its} expand! (m=2..{KRYLOVDIM})")
        parallelism = "single GPU" if not use_dist else \
            (f"basis row-sharded over {world} GPUs ({n_local} rows each, {scaling} scaling); per iteration libkrylov_hip issues 2 ncclAllReduce "
             "(2m+1 and 1 doubles) + 1 grouped ncclSend/Recv ghost exchange")
    elif args.config == "gkl":
        g = build_gkl(args.ny)
        sweep, sweep_its, n_local = g["sweep"], g["sweep_its"], g["n_local"]
        units_per_sweep = sweep_its                # the problem is fixed, ranks split its rows (strong scaling)
        alg_sweep = g["alg_sweep"]
        scaling = "strong"
        metric = "gkl_iterations_per_second"
        unit = "it/s (GKL expand! on the whole 5M x 1M map)"
        workload = g["workload"]
        parallelism = "single GPU" if not use_dist else \
            f"rows of A and of the U basis sharded over {world} GPUs, V basis sharded evenly; per iteration 1 ncclAllGather (v) + 1 ncclReduceScatter (A'u) + the all-reduces of the sweeps"
    else:  # block
        g = build_block(args.ny)
        sweep, sweep_its, n_local, bs, Kb = g["sweep"], g["sweep_its"], g["n_local"], g["bs"], g["Kb"]
        units_per_sweep = sweep_its
        alg_sweep = g["alg_sweep"]
        scaling = "strong"
        metric = "block_lanczos_steps_per_second"
        unit = "block steps/s (bs=16, 10M rows)"
        workload = g["workload"]
        parallelism = "single GPU" if not use_dist else f"rows sharded over {world} GPUs; Gram panels all-reduced (ncclAllReduce), ghost exchange per column of the block apply"

    if args.only_leg:
        assert args.config == "lanczos" and world == 1 and not use_dist
        spec = leg_spec(args.only_leg)
        nsw = 1 + max(1, args.config_steps)
        for _ in range(nsw):
            spec["sweep"]()
        sync()
        if spec.get("cleanup"):
            spec["cleanup"]()
        print(json.dumps({"only_leg": args.only_leg, "sweeps": nsw, "source_sha": kernel_source_sha()}), flush=True)
        return

    # warm-up sweeps; the last one is event-profiled per kernel class (breakdown only, untimed)
    ctx.prof_reset()
    for i in range(max(W, 1)):   # --warmup 0 still runs this one profiled sweep: the per-kernel breakdown (and which MGS form ran) comes from it
        ctx.prof_enable(1 if i == max(W, 1) - 1 else 0)
        sweep()
    barrier(); sync()
    ctx.prof_enable(0)
    breakdown, breakdown_launches = {}, {}
    for name in ("k_project", "k_unproject", "k_unproj_proj", "k_spmv_ell", "k_spmv_dia", "k_spmv_sell", "k_spmv_csr", "k_scal", "k_mgs_step", "k_dot",
                 "k_axpby", "k_block_gram", "k_block_update", "k_spmm_ell", "k_spmm_dia", "k_mgs_persist", "nccl_allreduce", "nccl_p2p", "nccl_gather"):
        ms, n = ctx.prof_get(name)
        if n:
            breakdown[name] = round(ms, 3)
            breakdown_launches[name] = int(n)
    # timed region: K sweeps; only the basis-streaming kernels (the dominant ones) carry HIP events
    ctx.prof_reset()
    ctx.prof_enable(0 if os.environ.get("KK_BENCH_NOPROF") else (1 if args.config == "block" else 2))   # block step: ms-scale kernels, every class
    stats0 = comm.stats() if comm else None
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(K):
        fact = sweep()
    barrier(); sync()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(0)
    stats1 = comm.stats() if comm else None
    ranks_agree = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must have seen bit-identical scalars (all-reduced inner products): a cheap end-to-end check of the
        # sharded path, printed with the collectives
        mine = (float(fact.normres),) if args.config == "block" else (float(fact.alphas[-1]), float(fact.betas[-1]))
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        ranks_agree = all(v == seen[0] for v in seen)

    def timed(sweep_fn, k):
        """K sweeps bracketed by barrier + sync on both sides, max over ranks"""
        barrier(); sync()
        t_ = time.perf_counter()
        for _ in range(k):
            f_ = sweep_fn()
        barrier(); sync()
        d_ = time.perf_counter() - t_
        if world > 1:
            tt = torch.tensor([d_], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d_ = float(tt.item())
        return d_, f_

    # The timed region above carries HIP-event pairs around the two basis-streaming kernels (the roofline object needs their
    # durations from THIS region).  The same K sweeps once more with no event anywhere show what the bracketing costs.
    unbracketed = None
    if args.config == "lanczos" and not os.environ.get("KK_BENCH_NOPROF"):
        d_, _ = timed(sweep, K)
        unbracketed = {"value": round(units_per_sweep * K / d_, 3), "ms_per_step": round(d_ / K * 1e3, 3),
                       "note": "same K sweeps, no HIP events recorded (the headline region brackets every launch of the basis-streaming kernels)"}

    # ---------------- the other scaling mode of config 2 as a secondary leg (N > 1)
    other_leg = None
    if other_leg_mode:
        prob2 = build_lanczos(other_leg_mode)
        prob2["sweep"]()
        d_, f2 = timed(prob2["sweep"], K)
        u2 = sweep_its * world if other_leg_mode == "weak" else sweep_its
        other_leg = {"scaling": other_leg_mode, "value": round(u2 * K / d_, 3), "ms_per_step": round(d_ / K * 1e3, 3),
                     "rows_per_gpu": prob2["n_local"], "rows_total": prob2["n_global"],
                     "job_iterations_per_second": round(sweep_its * K / d_, 3),
                     "algorithmic_equiv_frac_of_peak_per_gpu": round(algorithmic_bytes_sweep(prob2["n_global"], KRYLOVDIM) * K / d_ / 1e9 / (HBM_PEAK_GBPS * world), 4),
                     "unit": "it/s (10M-row Lanczos iterations, summed over GPUs)" if other_leg_mode == "weak" else "it/s (Lanczos iterations of the one 10M-row problem)"}
        del prob2

    # ---------------- roofline of the dominant kernel (HIP events on the kernels' stream)
    roofline = None
    if args.config == "lanczos":
        classes = {}
        for name in ("k_project", "k_unproject", "k_mgs_step", "k_mgs_persist"):
            ms, n = ctx.prof_get(name)
            if n:
                classes[name] = (ms, n)
        # the two basis-streaming kernels are within 1 % of each other: the ROOFLINE kernel is the one with more
        # algorithmic bytes (k_unproject: V once + w read and written), so that the object does not flip run to run
        dom = "k_mgs_persist" if "k_mgs_persist" in classes else \
            ("k_unproject" if "k_unproject" in classes else (max(classes, key=lambda k: classes[k][0]) if classes else None))
        if dom == "k_mgs_persist":
            ms, n = classes[dom]
            # one launch per expand at basis size m = 2..100: the whole MGS sweep of the reference (for q in V: s = <q, w>;
            # w -= s q, the pending "w -= alpha v" in front, |w| behind, w / |w| stored).  What the launch really moves: w read and
            # written once (it lives in registers in between), every basis vector once, plus the grid-rows of it that do not
            # fit on chip (LDS + spare registers) between the inner product and the update a second time:
            #   model(m) = (8 m + 24) N  (+ 8 m reread N when the second read misses the L2).  `frac` is PHYSICAL: counter traffic (stamped PMC pass) or this model
            # over the event-timed duration.  The contract figure of SURVEY 8(d), pass(m) = (16 m + 24) N -- every basis vector
            # read twice -- is what the projection-based kernels move and is carried as algorithmic_equiv_*; the least a strict
            # sweep with w on chip could move is (8 m + 16) N (min_bytes, frac_of_min).
            launches_per_sweep = KRYLOVDIM - 1
            pt = int(ctx.get_option("persist_threads"))
            ncu = int(ctx.get_option("num_cus"))
            ld_rows = (n_local + 511) // 512 * 512
            if (ld_rows // 512) % 2 == 0:
                ld_rows += 512
            rows = -(-ld_rows // (ncu * pt * 2))                                # grid-rows of double2 per thread
            lds_rows = min(rows, (160 * 1024 - 256) // (pt * 16)) if ctx.get_option("persist_lds") >= 1 else 0
            reg_rows = min(rows - lds_rows, 9) if (ctx.get_option("persist_lds") >= 2 and pt == 512) else 0
            reread = (rows - lds_rows - reg_rows) / rows
            scale_n = n / launches_per_sweep                                   # sweeps in the timed region
            alg = sum((16 * m + 24) * n_local for m in range(2, KRYLOVDIM + 1)) * scale_n
            # (round 4: the rows that are read a second time are the only ones the first read leaves cache-allocated -- 11 rows x 8 KB x
            # 32 CUs = 2.8 MB per XCD -- so that second read is an L2 hit and does not reach the fabric: the model counts every
            # basis vector once; the counters decide)
            model = sum((8 * m + 24) * n_local for m in range(2, KRYLOVDIM + 1)) * scale_n
            least = sum((8 * m + 16) * n_local for m in range(2, KRYLOVDIM + 1)) * scale_n
            tj, traffic_note = stamped_traffic("traffic.json")
            per_launch = (tj or {}).get(dom)
            roofline = physical_roofline(dom, ms * 1e-3, n, None if per_launch is None else per_launch * n, model, alg, least, traffic_note)
            cap_rows = int(ctx.get_option("persist_capacity_rows"))
            roofline["on_chip_parking"] = {"grid_rows_per_thread": int(rows), "parked_in_lds": int(lds_rows), "parked_in_registers": int(reg_rows),
                                           "second_read_fraction": round(reread, 4),
                                           "second_read_served_by": "the XCD's L2 (only these rows are cache-allocated by the first read; the parked rows are loaded non-temporally)"}
            roofline["persist"] = {"eligible": True, "rows": int(ld_rows), "capacity_rows": cap_rows, "rows_per_thread": int(2 * rows),
                                   "used_fraction_of_capacity": round(ld_rows / cap_rows, 4),
                                   "beyond_capacity": "a work vector longer than capacity_rows does not fit the register file of the chip: mgs_mode auto then runs the "
                                                      "low-synchronisation form (the `mgs2_lowsync` leg of this line); both sides of the limit are parity-tested "
                                                      "(tests/test_gpu_fullsize.py)"}
            roofline["scale_pass_folded"] = bool(ctx.get_option("fold_scale"))
            roofline["timed_region_kernel_ms"] = {k: round(v[0], 3) for k, v in classes.items()}
            roofline["one_sweep_kernel_ms_breakdown"] = {"bracketed": True, "note": "every kernel of ONE warm-up sweep between its own pair of HIP events: the brackets serialise the stream, so the sum exceeds ms_per_step", **breakdown}
        elif dom in ("k_project", "k_unproject"):
            ms, n = classes[dom]
            # one launch per expand at basis size m = 2..100: project moves (8m + 8) N bytes (V once + w; + 8 N with the Gram
            # row of the newest vector riding along), unproject (8m + 16) N (V once + w read/write): here the byte model IS the
            # algorithmic count of SURVEY 8(d) (their sum is pass(m) = (16m + 24) N) and the counters agree with it within 1 %
            extra = 8 if dom == "k_project" else 16
            scale_n = n / (KRYLOVDIM - 1)
            model = sum((8 * m + extra) * n_local for m in range(2, KRYLOVDIM + 1)) * scale_n
            tj, traffic_note = stamped_traffic("traffic.json")
            per_launch = (tj or {}).get(dom)
            roofline = physical_roofline(dom, ms * 1e-3, n, None if per_launch is None else per_launch * n, model, model, model, traffic_note)
            roofline["timed_region_kernel_ms"] = {k: round(v[0], 3) for k, v in classes.items()}
            roofline["one_sweep_kernel_ms_breakdown"] = {"bracketed": True, "note": "every kernel of ONE warm-up sweep between its own pair of HIP events: the brackets serialise the stream, so the sum exceeds ms_per_step", **breakdown}
        elif dom == "k_mgs_step":
            ms, n = classes[dom]
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(32.0 * n_local / (ms / n * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(32.0 * n_local / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                        "launches": int(n), "avg_launch_ms": round(ms / n, 5), "algorithmic_bytes_per_launch": 32 * n_local,
                        "one_sweep_kernel_ms_breakdown": {"bracketed": True, **breakdown}}

    if args.config == "block":
        # the two basis-streaming classes of the block step.  k_block_update_lds: W <- W - V P with the whole basis (kn = 32 ..
        # 112 columns after the push) streamed once and the 16-column residual block read and written: (8 kn + 256) N bytes per
        # launch, + TWO 16 -> 16 column launches per sweep that run in full (256 N each: the second CholQR2 round of the start block
        # in initialize and of the first residual block, which no commit has normalised yet -- 485 us each in profiles/r05_block_kernel_stats.csv;
        # rounds 3-4 counted one); the other second-round back-substitutions are skipped on the device (5 us each): in the launch count,
        # no bytes.  k_block_gram: counters only.
        tj, note = stamped_traffic("traffic_configs.json")
        tcfg = (tj or {}).get("configs", {}).get("block", {}) if tj else {}
        per_class = {}
        for cls in ("k_block_gram", "k_block_update", "k_spmm_dia", "k_spmm_ell"):
            ms, n = ctx.prof_get(cls)
            if n:
                per_class[cls] = (ms, n)
        model = {"k_block_update": (sum(8 * kn + 256 for kn in range(2 * bs, Kb + bs + 1, bs)) + 2 * 256) * float(n_local)}
        if per_class:
            dom = max(per_class, key=lambda c_: per_class[c_][0])
            ms, n = per_class[dom]
            tb, mb = tcfg.get(dom), model.get(dom)
            roofline = physical_roofline(dom, ms * 1e-3, n, None if tb is None else tb * K, None if mb is None else mb * K, None, note=note)
            roofline["per_block_step_ms"] = round(elapsed / K / sweep_its * 1e3, 3)
            roofline["classes"] = {c_: {"ms_per_sweep": round(v[0] / K, 3), "launches_per_sweep": round(v[1] / K, 1),
                                        "bytes_per_sweep": (tcfg.get(c_) if tcfg.get(c_) is not None else (round(model[c_]) if c_ in model else None)),
                                        "bytes_source": "pmc" if tcfg.get(c_) is not None else ("model" if c_ in model else None),
                                        "frac": (round((tcfg.get(c_) if tcfg.get(c_) is not None else model[c_]) * K / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                                                 if (tcfg.get(c_) is not None or c_ in model) else None)}
                                   for c_, v in per_class.items()}
            roofline["one_sweep_kernel_ms_breakdown"] = {"bracketed": True, "note": "every kernel of ONE warm-up sweep between its own pair of HIP events: the brackets serialise the stream, so the sum exceeds ms_per_step", **breakdown}

    # ---------------- secondary leg: the OTHER execution order of MGS2 on the same workload (the headline ran the library default)
    strict, lowsync_leg = None, None
    headline_is_strict = "k_mgs_persist" in breakdown or "k_mgs_step" in breakdown and breakdown.get("k_mgs_step", 0) > breakdown.get("k_project", 0)
    if args.config == "lanczos" and args.orth == "mgs2" and not args.no_strict_leg:
        other = 1 if headline_is_strict else 0
        ctx.set_option("mgs_mode", other)
        sweep()
        dts, fs = timed(sweep, 1)
        ctx.set_option("mgs_mode", MODE[args.mgs_mode])
        leg = {"value": round(units_per_sweep / dts, 3), "unit": "it/s", "ms_per_step": round(dts * 1e3, 3),
               "algorithmic_equiv_frac_of_peak_per_gpu": round(alg_sweep / dts / 1e9 / (HBM_PEAK_GBPS * world), 4),
               "max_alpha_reldiff_vs_headline": float(np.max(np.abs(np.array(fs.alphas) - np.array(fact.alphas)) / np.abs(np.array(fact.alphas))))}
        if other == 0:
            # strict MGS2 as the reference codes it: (176 + 16 m) N algorithmic bytes per expand as well
            leg["note"] = ("mgs_mode=0: the reference's sequential order (src/orthonormal.jl:414-439), one basis vector after the other; "
                           + ("row-sharded: the persistent kernel with its in-kernel cross-rank reduction where the communicator offers it (`xsync`), "
                              "else one fused axpy+dot kernel and one all-reduce per basis vector (32 N bytes per vector)" if use_dist else
                              "persistent cooperative kernel, w resident in registers"))
            strict = leg
        else:
            leg["note"] = ("mgs_mode=1: low-synchronisation MGS2 -- one projection pass + exact triangular solve with the Gram matrix of the basis "
                           "+ one update pass (k_project / k_unproject, the basis read twice per expand: 16 N bytes per vector); the "
                           "round-1/2 headline configuration")
            lowsync_leg = leg

    # ---------------- physical bytes of one whole sweep (config 2, one GPU): per class, counter traffic where stamped, byte model otherwise
    sweep_physical = None
    if args.config == "lanczos" and world == 1 and not use_dist and roofline and roofline.get("achieved"):
        tj, _ = stamped_traffic("traffic.json")
        tj = tj or {}
        N_ = float(n_local)
        fmt = prob["keep"][0].info()["format"]
        spmv_cls = "k_spmv_dia" if "DIA" in fmt and ctx.get_option("spmv_dia") else "k_spmv_ell"
        spmv_model = {"ELL+DIA const": 24.0, "ELL+DIA": 64.0}.get(fmt, 84.0) * N_ if spmv_cls == "k_spmv_dia" else 84.0 * N_
        per_launch_dom = roofline["traffic"] if roofline.get("traffic") is not None else roofline.get("hbm_model_bytes_per_launch")
        n_scal = 3 if (headline_is_strict and ctx.get_option("fold_scale")) else 2 + sweep_its
        parts = {roofline["kernel"]: per_launch_dom * sweep_its * (2 if roofline["kernel"] in ("k_project", "k_unproject") else 1),
                 spmv_cls: (tj.get(spmv_cls) or spmv_model) * (sweep_its + 1),
                 "k_scal": (tj.get("k_scal") or 16.0 * N_) * n_scal,
                 "initialize (norms, axpby, one vector-vector MGS2 step; model)": 88.0 * N_}
        tot = float(sum(parts.values()))
        sweep_physical = {"bytes_per_sweep": round(tot), "GBps": round(tot * K / elapsed / 1e9, 1),
                          "frac_of_peak": round(tot * K / elapsed / 1e9 / HBM_PEAK_GBPS, 4),
                          "parts_bytes": {k: round(v) for k, v in parts.items()},
                          "note": "bytes the kernels of one sweep really move (counter traffic per launch where profiles/traffic.json is current, the "
                                  "byte models of DESIGN.md section 3 otherwise) over the wall time of the sweep"}

    # ---------------- the second-largest item of the sweep: the sparse apply (its own line and a target)
    second_kernel = None
    if args.config == "lanczos" and world == 1 and not use_dist and roofline:
        fmt = prob["keep"][0].info()["format"]
        cls = "k_spmv_dia" if "DIA" in fmt and ctx.get_option("spmv_dia") else "k_spmv_ell"
        if cls in breakdown:
            tj, note2 = stamped_traffic("traffic.json")
            model_b = ({"ELL+DIA const": 24.0, "ELL+DIA": 64.0}.get(fmt, 84.0) if cls == "k_spmv_dia" else 84.0) * float(n_local)
            pmc_b = (tj or {}).get(cls)
            b_ = pmc_b if pmc_b is not None else model_b
            # duration per launch: the kernel-trace pass of the same command on the same sources where there is one (stamped next to the
            # counter traffic); else the HIP-event brackets of the profiled warm-up sweep, which for a kernel this short also contain
            # the launch gaps around it -- an upper bound of the duration, i.e. a lower bound of the rate
            rp_us = ((tj or {}).get("rocprof_avg_us") or {}).get(cls)
            ms_launch = rp_us * 1e-3 if rp_us else breakdown[cls] / breakdown_launches[cls]
            second_kernel = {"kernel": cls, "format": fmt, "avg_launch_ms": round(ms_launch, 5), "bytes_per_launch": round(b_),
                             "bytes_source": "pmc" if pmc_b is not None else "model",
                             "duration_source": "rocprofv3 kernel trace of the same command (stamped)" if rp_us else "HIP events around each launch (upper bound: includes launch gaps)",
                             "achieved": round(b_ / (ms_launch * 1e-3) / 1e9, 1),
                             "frac": round(b_ / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "target_frac": 0.75,
                             "launches_per_sweep": breakdown_launches[cls], "ms_per_sweep": round(breakdown[cls], 3)}

    # ---------------- `sharded_world1`: the sweep as every N > 1 rank runs it (row-sharded context: RCCL collectives issued by the
    # library, the persistent kernel's inner products summed over the ranks inside the launch -- kk_xsync.h, here through the
    # rank's own sync area) on this ONE GPU, in a child process with KK_BENCH_FORCE_DIST=1
    sharded_leg = None
    if (args.config == "lanczos" and world == 1 and not use_dist and not args.no_sharded_leg and args.ny == NY
            and not os.environ.get("KK_BENCH_FORCE_DIST")):
        import subprocess
        cmd = [sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--steps", "3", "--warmup", "1", "--orth", args.orth,
               "--no-cpu-baseline", "--no-configs", "--no-strict-leg", "--no-sharded-leg"]
        try:
            r_ = subprocess.run(cmd, env=dict(os.environ, KK_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=300)
            ln_ = [l for l in r_.stdout.splitlines() if l.startswith("{")]
            if r_.returncode == 0 and ln_:
                d_ = json.loads(ln_[-1])
                sharded_leg = {"value": d_["value"], "unit": "it/s", "ms_per_step": d_["ms_per_step"], "collectives": d_.get("collectives"),
                               "roofline": {k: d_["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")} if d_.get("roofline") else None,
                               "xsync": d_.get("xsync"),
                               "note": "world size 1 with the collectives forced: the ncclAllReduce of alpha0 + the (self-served) ghost exchange per "
                                       "iteration are really issued, and the persistent strict-MGS kernel runs its cross-rank reduction (one tagged granule "
                                       "per basis vector stored into, and polled from, the rank's own IPC-shareable sync area) as on every N > 1 rank.  "
                                       "Efficiency of an N-GPU weak-scaling line = its value / (N x this value)"}
            else:
                sharded_leg = {"error": (r_.stderr or r_.stdout)[-400:]}
        except Exception as e_:   # the leg must never cost the line
            sharded_leg = {"error": repr(e_)[:400]}

    # ---------------- `configs` block: BASELINE.json configs[2..4] at full size + the general-format leg of configs[1], one GPU
    configs = None
    if args.config == "lanczos" and world == 1 and not use_dist and not args.no_configs and args.ny == NY:
        configs = {}
        for key, name in (("lanczos_general_format", "lanczos_ell"), ("gmres_2M", "gmres"), ("block_10M_bs16", "block"), ("gkl_5Mx1M", "gkl")):
            spec = leg_spec(name)
            try:
                leg, _ = run_leg(ctx, name, spec["sweep"], spec["units"], max(1, args.config_steps) * spec.get("kmult", 1), spec["model"], spec["alg"], spec["meta"])
            finally:
                if spec.get("cleanup"):
                    spec["cleanup"]()
            if name == "block":
                # the figure of earlier rounds: the 6 expand! alone (the line's "step" also holds initialize = one more block QR + apply)
                t6 = min(spec["keep"]["expands_only"]() for _ in range(3))
                leg["ms_per_block_step"] = round(t6 / spec["units"] * 1e3, 3)
                leg["block_step_algorithmic_equiv_frac"] = round(spec["alg"] / t6 / 1e9 / HBM_PEAK_GBPS, 4)
            if name == "gkl":
                gather_roofline(leg, spec["keep"]["nnz"], spec["units"] * 2 + 2, gather_ceiling)
            configs[key] = leg
            del spec

    line = None
    if rank == 0:
        value = units_per_sweep * K / elapsed
        out = {
            "metric": metric, "value": round(value, 3), "unit": unit,
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": workload,
                "orth": {"cgs2": "ClassicalGramSchmidt2", "mgs2": "ModifiedGramSchmidt2 (reference default; "
                         + ("strict sequential order of src/orthonormal.jl:414-439, persistent kernel)" if headline_is_strict else "low-synchronisation form)")}[args.orth],
                "rows_per_gpu": n_local, "parallelism": parallelism,
            },
            "job_iterations_per_second": round(sweep_its * K / elapsed, 3),
            "hbm_physical": sweep_physical,
            # the SURVEY 8(d) contract figure: ALGORITHMIC bytes (every basis vector read twice per pass) over the wall time.  It is
            # what the target of BASELINE.md is written in (>= 0.60) and exceeds 1 once a kernel keeps data on chip between its two uses
            "algorithmic_equiv_GBps": round(alg_sweep * K / elapsed / 1e9, 1),
            "algorithmic_equiv_frac_of_peak_per_gpu": round(alg_sweep * K / elapsed / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "roofline": roofline,
        }
        # (the figures a truncated tail must not hide sit up here, next to the roofline: VERDICT r5 item 6)
        if sharded_leg and "value" in sharded_leg:
            out["sharded_world1_it_per_s"] = sharded_leg["value"]
        if configs:
            out["configs_summary"] = {k: (v.get("ms_per_block_step") if k.startswith("block") else v.get("value")) for k, v in configs.items() if isinstance(v, dict)}
        if overridden:
            out["options_overridden"] = overridden      # NOT the library defaults: an A/B line, not a headline
        if ceiling and roofline and roofline.get("achieved"):
            # against what THIS box's memory system delivers to a pure read stream right now (the kernels read non-temporally)
            roofline["attainable_read_GBps"] = ceiling["nt_read_GBps"]
            roofline["frac_of_attainable"] = round(roofline["achieved"] / ceiling["nt_read_GBps"], 4)
            out["hbm_read_ceiling"] = ceiling
        if second_kernel:
            roofline["second_kernel"] = second_kernel
            if ceiling and second_kernel.get("achieved"):
                second_kernel["frac_of_attainable"] = round(second_kernel["achieved"] / ceiling["nt_read_GBps"], 4)
        if sharded_leg:
            out["sharded_world1"] = sharded_leg
        if configs:
            out["configs"] = configs
        if args.config != "block":
            out["last_alpha"], out["last_beta"] = fact.alphas[-1], fact.betas[-1]
        if world > 1 and args.config == "lanczos":
            out["scaling_note"] = ("rows are sharded: the persistent strict-MGS kernel sums its inner products over the ranks inside the launch (tagged "
                                   "granules stored into the peers' IPC-mapped sync areas over xGMI, csrc/kk_xsync.h) when kk_comm_init could map every "
                                   "peer -- see `xsync`; otherwise mgs_mode auto falls back to the low-synchronisation form (2 RCCL all-reduces per "
                                   "iteration, basis read twice).  The like-for-like single-GPU rate for efficiency accounting is the N = 1 line's "
                                   "`sharded_world1` leg")
        if unbracketed:
            out["without_event_bracketing"] = unbracketed
        if other_leg:
            out["other_scaling_leg"] = other_leg
        if strict:
            out["mgs2_strict"] = strict
        if lowsync_leg:
            out["mgs2_lowsync"] = lowsync_leg
        if comm:
            out["xsync"] = {"active": bool(ctx.get_option("xsync_active")), "persistent_launches_with_cross_rank_reduction": int(ctx.get_option("xsync_launches")),
                            "num_cus": int(ctx.get_option("num_cus")),
                            # measured by kk_comm_init's hand-shake on THIS machine (slowest rank): one in-kernel reduction over the ranks (a store into
                            # every peer's sync area + the poll of the own one: device to device when the ranks own different GPUs) and one small RCCL
                            # all-reduce on the stream -- the two prices the library's route rule weighs (kk_xs_pays)
                            "hop_us": round(ctx.get_option("xsync_hop_us"), 3), "allreduce_us": round(ctx.get_option("comm_allreduce_us"), 2),
                            "ranks_on_this_gpu": int(ctx.get_option("ranks_on_this_gpu")),
                            "rule": "in-kernel route iff reductions x hop_us <= allreduce_us + vector_steps x t_sync x (rows / threshold_rows - 1); option xsync = 2 forces it, 0 disables it"}
            info = comm.info()
            per = {k: (stats1[k] - stats0[k]) / (K * sweep_its) for k in stats1}
            out["collectives"] = {"library": "RCCL inside libkrylov_hip (kk_comm_init)", "rccl_version": info["rccl_version"],
                                  "ranks": info["world"], "per_iteration": {k: round(v, 3) for k, v in per.items()},
                                  # stream time of the collectives in the event-profiled warm-up sweep (waiting for the peers included)
                                  "per_iteration_us": {k: round(breakdown[k] * 1e3 / sweep_its, 2) for k in ("nccl_allreduce", "nccl_p2p", "nccl_gather")
                                                       if k in breakdown},
                                  "ranks_agree_bitwise": ranks_agree}
        if args.config == "lanczos" and world == 1 and not use_dist and not args.no_cpu_baseline:
            x0_host = x0_handle[0].get()                       # the GPU run's own start vector (80 MB over PCIe, once)
            base, al_c, be_c = cpu_baseline(orth.code, x0_host, args.ny)
            out["cpu_baseline"] = base
            par = parity_block(fact.alphas, fact.betas, al_c, be_c) if al_c is not None else None
            if par is not None and args.parity_seeds > 1:
                # The Ritz figure is rounding noise amplified by |T| / theta_min ~ 2e4: one start vector is thin evidence.
                # More start vectors, each a full GPU sweep against a full CPU sweep; the block carries the maximum.
                lib = _load_cpu_ref()
                per_seed = [{"seed": 3, **{k: par[k] for k in ("alpha_relerr", "beta_relerr", "ritz_relerr", "ritz_abserr_over_norm")}}]
                for sd in range(4, 3 + args.parity_seeds):
                    x0_handle[0].rand_(sd)
                    f2 = sweep()
                    _, al2, be2 = _run_cpu_ref(lib, args.ny, x0_handle[0].get(), orth.code, base["cores"])
                    if al2 is None:
                        continue
                    p2 = parity_block(f2.alphas, f2.betas, al2, be2)
                    per_seed.append({"seed": sd, **{k: p2[k] for k in ("alpha_relerr", "beta_relerr", "ritz_relerr", "ritz_abserr_over_norm")}})
                for k in ("alpha_relerr", "beta_relerr", "ritz_relerr", "ritz_abserr_over_norm"):
                    par[k] = max(e[k] for e in per_seed)
                par["per_start_vector"] = per_seed
                par["ok"] = bool(max(par["alpha_relerr"], par["beta_relerr"], par["ritz_relerr"]) <= par["tol"])
                par["note"] = f"maximum over {len(per_seed)} start vectors (rand seeds), each a full {NX}x{args.ny} GPU sweep vs a full CPU sweep"
            out["parity"] = par
        line = json.dumps(out)
    if comm:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        C.CDLL(None).fflush(None)   # RCCL's version banner sits in the C stdio buffer until exit: push it out first
    except Exception:
        pass
    if rank == 0:
        print(line, flush=True)   # the ONE JSON line, last thing on stdout (RCCL prints its banner on stdout too)


if __name__ == "__main__":
    main()
