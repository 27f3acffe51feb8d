"""Run-ahead of a whole Lanczos step (csrc/kk_krylov.hip::la_enqueue): the apply AND the persistent sweep of step k+1 are
enqueued before the host waits for the scalars of step k (factorizations/lanczos.jl:250-272 is called once per step by the
reference's drivers; the GPU must not idle through that round trip).  Same kernels, operands and order as the call-by-call
route, so everything must be bitwise the same; anything that touches the slab between two calls drops the run-ahead."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["persist", "panel"])
def lctx(kk, request):
    c = kk.Context(0)
    if c.get_option("mgs_persist") == 0:
        pytest.skip("no cooperative launch on this device")
    c.set_option("mgs_mode", 0)                      # strict order at every size
    c.set_option("mgs_panel", 1 if request.param == "panel" else 0)
    yield c
    c.close()


def run(kk, ctx, A, x0, steps, lookahead, capacity=None, poke=()):
    ctx.set_option("lookahead", lookahead)
    it = kk.LanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=capacity or steps + 3)
    f = kk.initialize(it)
    seen = {}
    for i in range(steps):
        f = kk.expand_(it, f)
        if i in poke:
            seen[i] = f.r.get().copy()
    return f, seen


def test_bitwise_equal_to_the_call_by_call_route(kk, ko, lctx):
    nx, ny, steps = 52, 40, 28
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    f1, _ = run(kk, lctx, A, x0, steps, 1)
    a1, b1, V1 = np.array(f1.alphas), np.array(f1.betas), f1.V.to_numpy().copy()
    f0, _ = run(kk, lctx, A, x0, steps, 0)
    lctx.set_option("lookahead", 1)
    assert np.array_equal(a1, np.array(f0.alphas)) and np.array_equal(b1, np.array(f0.betas))
    assert np.array_equal(V1, f0.V.to_numpy())
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for _ in range(steps):
        of = ko.lanczos_expand(oit, of)
    assert np.max(np.abs(a1 - of.alphas) / np.abs(of.alphas)) < 1e-10 and np.max(np.abs(b1 - of.betas) / np.abs(of.betas)) < 1e-10
    assert np.max(np.abs(V1.T @ V1 - np.eye(V1.shape[1]))) < 1e-12


def test_stops_at_the_end_of_the_slab_and_survives_interruptions(kk, ko, lctx):
    """capacity = krylovdim + 2 (what the drivers allocate): the run-ahead must not enqueue a sweep beyond the last step;
    reading residual(F) in the middle drops the step enqueued ahead and the run continues unharmed"""
    nx, ny, steps = 40, 30, 18
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(5).random(n)
    lctx.prof_reset(); lctx.prof_enable(1)
    f, seen = run(kk, lctx, A, x0, steps, 1, capacity=steps + 3, poke={4, 5, 11})
    lctx.prof_enable(0)
    launches = lctx.prof_get("k_mgs_persist")[1] + lctx.prof_get("k_mgs_panel")[1]
    assert steps <= launches <= steps + 3     # one per step, + one dropped run-ahead per interruption, none beyond the slab
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for i in range(steps):
        of = ko.lanczos_expand(oit, of)
        if i in seen:
            assert np.max(np.abs(seen[i] - of.r)) < 1e-10 * np.linalg.norm(of.r)
    assert np.max(np.abs(np.array(f.alphas) - of.alphas) / np.abs(of.alphas)) < 1e-10
    assert np.max(np.abs(np.array(f.betas) - of.betas) / np.abs(of.betas)) < 1e-10
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12


def test_timeout_of_a_step_enqueued_ahead_is_recovered(kk, ko, lctx):
    nx, ny, steps = 44, 36, 20
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(7).random(n)
    lctx.set_option("lookahead", 1)
    it = kk.LanczosIterator(kk.SparseOperator(A, lctx, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    f = kk.initialize(it)
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for i in range(steps):
        if i in (6, 13):
            lctx.set_option("persist_fault", 1)    # the next persistent launch (a step enqueued ahead) behaves like a timed-out one
        f = kk.expand_(it, f)
        of = ko.lanczos_expand(oit, of)
    assert lctx.get_option("persist_timeouts") == 2
    assert np.max(np.abs(np.array(f.alphas) - of.alphas) / np.abs(of.alphas)) < 1e-10
    assert np.max(np.abs(np.array(f.betas) - of.betas) / np.abs(of.betas)) < 1e-10
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12


@pytest.mark.parametrize("orth_name", ["mgs", "mgs2"])
def test_arnoldi_run_ahead_bitwise_and_gmres_counts(kk, ko, lctx, orth_name):
    """arnoldi.jl:199-245 with the next step's apply + sweeps enqueued ahead: same H bits as the call-by-call route, the
    oracle's Hessenberg matrix, shrink! in the middle (drops the run-ahead), and linsolve(GMRES) with equal counts"""
    dev = {"mgs": kk.ModifiedGramSchmidt(), "mgs2": kk.ModifiedGramSchmidt2()}[orth_name]
    ref = {"mgs": ko.MGS, "mgs2": ko.MGS2}[orth_name]
    C = ko.convection_diffusion_2d(48, 40)
    x0 = np.random.default_rng(8).random(C.shape[0])
    out = {}
    for la in (1, 0):
        lctx.set_option("lookahead", la)
        it = kk.ArnoldiIterator(kk.SparseOperator(C, lctx), x0, dev, capacity=26)
        f = kk.initialize(it)
        for i in range(14):
            f = kk.expand_(it, f)
        f = kk.shrink_(f, 9)
        for i in range(8):
            f = kk.expand_(it, f)
        out[la] = (np.array(f.H, dtype=float).copy(), f.V.to_numpy().copy(), f.r.get().copy())
    lctx.set_option("lookahead", 1)
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])
    oit = ko.ArnoldiIterator(C, x0.copy(), ref)
    of = ko.arnoldi_initialize(oit)
    for i in range(14):
        of = ko.arnoldi_expand(oit, of)
    of = ko.arnoldi_shrink(of, 9)
    for i in range(8):
        of = ko.arnoldi_expand(oit, of)
    assert np.max(np.abs(out[1][0] - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))
    assert np.max(np.abs(out[1][2] - of.r)) < 1e-10 * np.linalg.norm(of.r)
    if orth_name == "mgs2":
        b = np.random.default_rng(4).random(C.shape[0])
        tol = 1e-10 * np.linalg.norm(b)
        x, info = kk.linsolve(kk.SparseOperator(C, lctx), b, None, kk.GMRES(dev, 20, 25, tol), 0.1, 1.0)
        xo, oinfo = ko.gmres(C, b, None, 0.1, 1.0, krylovdim=25, maxiter=20, tol=tol, orth=ref)
        assert (info.converged, info.numiter, info.numops) == (oinfo.converged, oinfo.numiter, oinfo.numops)
        assert np.linalg.norm(0.1 * x + C @ x - b) <= 1.01 * tol


def test_arnoldi_timeout_of_a_step_enqueued_ahead(kk, ko, lctx):
    C = ko.convection_diffusion_2d(40, 32)
    x0 = np.random.default_rng(8).random(C.shape[0])
    lctx.set_option("lookahead", 1)
    it = kk.ArnoldiIterator(kk.SparseOperator(C, lctx), x0, kk.ModifiedGramSchmidt2(), capacity=22)
    f = kk.initialize(it)
    oit = ko.ArnoldiIterator(C, x0.copy(), ko.MGS2)
    of = ko.arnoldi_initialize(oit)
    for i in range(16):
        if i in (5, 11):
            lctx.set_option("persist_fault", 1)
        f = kk.expand_(it, f)
        of = ko.arnoldi_expand(oit, of)
    assert lctx.get_option("persist_timeouts") == 2
    assert np.max(np.abs(np.asarray(f.H) - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12


@pytest.mark.parametrize("case", ["lanczos", "arnoldi_mgs", "arnoldi_mgs2"])
@pytest.mark.parametrize("fault_steps", [(0,), (0, 9), (3,)])
def test_timeout_of_the_first_launch_of_a_run_ahead_chain(kk, ko, lctx, case, fault_steps):
    """the launch of the CURRENT step times out while the step enqueued behind it goes out -- the
    first expand! after initialize (fault at 0), the first one after an interruption that dropped the chain (residual read
    at step 8 -> fault at 9) and, for comparison, a launch in the middle of a chain (3: that one was enqueued ahead).  The
    nested sweep of la_enqueue used to overwrite (slot, token) of the current launch and clear its pending check: the
    timeout went unnoticed and stale alpha / beta came back with an unorthogonalised column marked as a basis vector."""
    steps = 16
    lctx.set_option("lookahead", 1)
    if case == "lanczos":
        A = ko.laplacian_2d(44, 36, shift_diag=10 * np.linspace(0, 1, 44 * 36) ** 2)
        x0 = np.random.default_rng(11).random(A.shape[0])
        it = kk.LanczosIterator(kk.SparseOperator(A, lctx, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
        oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
        of = ko.lanczos_initialize(oit)
        oexp = ko.lanczos_expand
    else:
        dev, ref = (kk.ModifiedGramSchmidt(), ko.MGS) if case == "arnoldi_mgs" else (kk.ModifiedGramSchmidt2(), ko.MGS2)
        A = ko.convection_diffusion_2d(40, 32)
        x0 = np.random.default_rng(12).random(A.shape[0])
        it = kk.ArnoldiIterator(kk.SparseOperator(A, lctx), x0, dev, capacity=steps + 3)
        oit = ko.ArnoldiIterator(A, x0.copy(), ref)
        of = ko.arnoldi_initialize(oit)
        oexp = ko.arnoldi_expand
    f = kk.initialize(it)
    t0 = lctx.get_option("persist_timeouts")
    for i in range(steps):
        if i in fault_steps:
            lctx.set_option("persist_fault", 1)
        f = kk.expand_(it, f)
        of = oexp(oit, of)
        if i == 8:
            r = f.r.get()     # drops the run-ahead: step 9 starts a new chain
            assert np.max(np.abs(r - of.r)) < 1e-10 * np.linalg.norm(of.r)
    assert lctx.get_option("persist_timeouts") - t0 == len(fault_steps)
    if case == "lanczos":
        assert np.max(np.abs(np.array(f.alphas) - of.alphas) / np.abs(of.alphas)) < 1e-10
        assert np.max(np.abs(np.array(f.betas) - of.betas) / np.abs(of.betas)) < 1e-10
    else:
        assert np.max(np.abs(np.asarray(f.H) - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))
        assert abs(f.normres - of.normres) < 1e-10 * abs(of.normres)
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12


@pytest.mark.parametrize("route", ["cgs2", "mgs2_lowsync"])
def test_projection_route_run_ahead_is_bitwise_the_call_by_call_route(kk, ko, route):
    """la_enqueue_proj (round 5): the projection-based Lanczos step -- CGS2, and MGS2 in its low-synchronisation form, the routes of
    vectors too short for the persistent kernels -- enqueued one call ahead: scale of the new basis vector by the device's 1 / beta,
    projection (+ triangular solve), update, read-back.  Bit-identical to lookahead = 0, the oracle's trajectory, interruptions
    (residual read, shrink!) drop the step enqueued ahead and settle the column it had scaled."""
    c = kk.Context(0)
    try:
        c.set_option("mgs_mode", 1)
        c.set_option("fused_step", 0)     # (round 6: vectors this short take the one-launch step by default -- tests/test_gpu_fstep.py; this test is about the projection pair)
        dev, ref = (kk.ClassicalGramSchmidt2(), ko.CGS2) if route == "cgs2" else (kk.ModifiedGramSchmidt2(), ko.MGS2)
        nx, ny, steps = 52, 40, 28
        n = nx * ny
        A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
        x0 = np.random.default_rng(3).random(n)
        out = {}
        for interrupted in (False, True):
            for la in (1, 0):
                c.set_option("lookahead", la)
                c.prof_reset(); c.prof_enable(1)
                it = kk.LanczosIterator(kk.SparseOperator(A, c, symmetric=True), x0, dev, capacity=steps + 3)
                f = kk.initialize(it)
                r9 = None
                for i in range(steps):
                    f = kk.expand_(it, f)
                    if interrupted and i == 9:
                        r9 = f.r.get().copy()          # settles the scaled column, drops the step enqueued ahead
                    if i == 17:
                        f = kk.shrink_(f, 12)
                c.prof_enable(0)
                out[(interrupted, la)] = (np.array(f.alphas), np.array(f.betas), f.V.to_numpy().copy(), r9, c.prof_get("k_scal")[1])
        c.set_option("lookahead", 1)
        # undisturbed: every bit equal (same kernels, operands and order), and the scale passes are the run-ahead's own launches
        assert np.array_equal(out[(False, 1)][0], out[(False, 0)][0]) and np.array_equal(out[(False, 1)][1], out[(False, 0)][1])
        assert np.array_equal(out[(False, 1)][2], out[(False, 0)][2])
        # reading r at step 9 un-scales and re-scales the column on the run-ahead route only: agreement to rounding from there on
        a1, a0 = out[(True, 1)], out[(True, 0)]
        assert np.max(np.abs(a1[0] - a0[0]) / np.abs(a0[0])) < 1e-13 and np.max(np.abs(a1[1] - a0[1]) / np.abs(a0[1])) < 1e-13
        assert np.max(np.abs(a1[2] - a0[2])) < 1e-12 and np.max(np.abs(a1[3] - a0[3])) < 1e-13 * np.linalg.norm(a0[3])
        out = {1: out[(True, 1)], 0: out[(True, 0)]}
        oit = ko.LanczosIterator(A, x0.copy(), ref)
        of = ko.lanczos_initialize(oit)
        for i in range(steps):
            of = ko.lanczos_expand(oit, of)
            if i == 17:
                of = ko.lanczos_shrink(of, 12)
        assert np.max(np.abs(out[1][0] - of.alphas) / np.abs(of.alphas)) < 1e-10 and np.max(np.abs(out[1][1] - of.betas) / np.abs(of.betas)) < 1e-10
        V1 = out[1][2]
        assert np.max(np.abs(V1.T @ V1 - np.eye(V1.shape[1]))) < 1e-12
    finally:
        c.close()


@pytest.mark.parametrize("route", ["persist", "panel", "mgs2_lowsync", "cgs2"])
def test_two_factorizations_taking_turns_on_one_context(kk, ko, route):
    """Work enqueued ahead for one factorization leaves state in the context's SHARED scalar workspace (alpha0 of the speculative apply,
    |w| and 1 / |w| of the step in flight) that its next call relies on; a second factorization stepping in turns on the same context
    overwrites it.  The library notices (any entry point handed another slab bumps the context's generation, csrc/kk_host.h::
    ctx_foreign_touch) and each call redoes its apply and step from the slab: both trajectories must be the oracle's.  (Found in
    round 5 by tests/test_gpu_parity.py::test_function_operator once the projection route ran ahead: the interleaved run returned the
    other run's beta.)"""
    c = kk.Context(0)
    try:
        c.set_option("mgs_mode", {"persist": 0, "panel": 0, "mgs2_lowsync": 1, "cgs2": 1}[route])
        c.set_option("mgs_panel", 1 if route == "panel" else 0)
        dev, ref = (kk.ClassicalGramSchmidt2(), ko.CGS2) if route == "cgs2" else (kk.ModifiedGramSchmidt2(), ko.MGS2)
        nx, ny, steps = 44, 36, 18
        n = nx * ny
        A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
        op = kk.SparseOperator(A, c, symmetric=True)
        xs = [np.random.default_rng(s).random(n) for s in (3, 4)]
        its = [kk.LanczosIterator(op, x, dev, capacity=steps + 3) for x in xs]
        fs = [kk.initialize(it) for it in its]
        for _ in range(steps):
            fs = [kk.expand_(it, f) for it, f in zip(its, fs)]
        for x, f in zip(xs, fs):
            oit = ko.LanczosIterator(A, x.copy(), ref)
            of = ko.lanczos_initialize(oit)
            for _ in range(steps):
                of = ko.lanczos_expand(oit, of)
            assert np.max(np.abs(np.array(f.alphas) - of.alphas) / np.abs(of.alphas)) < 1e-10
            assert np.max(np.abs(np.array(f.betas) - of.betas) / np.abs(of.betas)) < 1e-10
            V = f.V.to_numpy()
            assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
    finally:
        c.close()


def test_switching_routes_on_one_slab_does_not_repeat_every_step(kk, ko):
    """the two kinds of run-ahead share the slab's bookkeeping: after steps on the projection route (kind 1) the persistent route's
    run-ahead (kind 0) must be RECOGNISED by the next call -- found by bench.py's general-format leg, which ran after the
    `mgs2_lowsync` leg on the same slab: every step was enqueued ahead, not recognised, and executed again (569 instead of 1106 it/s)"""
    c = kk.Context(0)
    try:
        if c.get_option("mgs_persist") == 0:
            pytest.skip("persistent route off on this device")
        c.set_option("mgs_panel", 0)
        c.set_option("fused_step", 0)     # (the low-sync leg of this test is the projection pair, not the one-launch step of round 6)
        nx, ny = 44, 36
        n = nx * ny
        A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
        x0 = np.random.default_rng(3).random(n)
        it = kk.LanczosIterator(kk.SparseOperator(A, c, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=40)
        f = kk.initialize(it)
        c.set_option("mgs_mode", 1)
        for _ in range(8):
            f = kk.expand_(it, f)
        c.set_option("mgs_mode", 0)
        f = kk.expand_(it, f)                       # the switch itself may cost one repeated step
        c.prof_reset(); c.prof_enable(1)
        for _ in range(10):
            f = kk.expand_(it, f)
        c.prof_enable(0)
        assert c.prof_get("k_mgs_persist")[1] == 10, c.prof_get("k_mgs_persist")     # one launch per step: each found the one enqueued for it
        c.set_option("mgs_mode", 1)
        f = kk.expand_(it, f)
        c.prof_reset(); c.prof_enable(1)
        for _ in range(8):
            f = kk.expand_(it, f)
        c.prof_enable(0)
        assert c.prof_get("k_project")[1] == 8 and c.prof_get("k_mgs_persist")[1] == 0, (c.prof_get("k_project"), c.prof_get("k_mgs_persist"))
        oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
        of = ko.lanczos_initialize(oit)
        for _ in range(28):
            of = ko.lanczos_expand(oit, of)
        assert np.max(np.abs(np.array(f.alphas) - of.alphas) / np.abs(of.alphas)) < 1e-10
        assert np.max(np.abs(np.array(f.betas) - of.betas) / np.abs(of.betas)) < 1e-10
    finally:
        c.close()
