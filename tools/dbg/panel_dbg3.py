import sys; sys.path.insert(0,"krylovkit.jl_amd"); sys.path.insert(0,"oracle")
import numpy as np, krylovkit_hip as kk, krylov_oracle as ko
c = kk.Context(0); c.set_option("panel_min_rows", 0)
for (n, m, mode, reps, alg) in [(2500000, 4, 0, 400, "orthonormalize"), (2500000, 4, 0, 200, "orthogonalize"), (6000000, 3, 0, 200, "orthonormalize")]:
    c.set_option("mgs_mode", mode)
    rng = np.random.default_rng(n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = Q @ rng.standard_normal(m) * 3 + rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 3, c)
    for j in range(m): B.upload(j, Q[:, j])
    B.length = m
    wo, xo = ko.orthogonalize(w.copy(), [Q[:, j].copy() for j in range(m)], ko.MGS)
    if alg == "orthonormalize": wo = wo / np.linalg.norm(wo)
    B[m + 1].set(w)
    events = 0
    c.prof_reset(); c.prof_enable(1)
    for rep in range(reps):
        B[m].scale_from_(B[m + 1], 1.0)
        if alg == "orthonormalize": x, nrm, _ = B.orthonormalize(B[m], kk.ModifiedGramSchmidt())
        else: x, nrm, _ = B.orthogonalize(B[m], kk.ModifiedGramSchmidt())
        got = B[m].get()
        bad = np.nonzero(np.abs(got - wo) > 1e-9)[0]
        if bad.size: events += 1; print("  bad", rep, bad.size, bad[:4])
    c.prof_enable(0)
    print(n, m, alg, "events", events, "of", reps, "panel launches", c.prof_get("k_mgs_panel")[1], "persist launches", c.prof_get("k_mgs_persist")[1], flush=True)
    B.free()
