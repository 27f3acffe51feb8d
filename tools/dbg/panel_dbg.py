import sys; sys.path.insert(0,"krylovkit.jl_amd"); sys.path.insert(0,"oracle")
import numpy as np, krylovkit_hip as kk, krylov_oracle as ko
c = kk.Context(0); c.set_option("panel_min_rows", 0)
for (n, m, mode) in [(2500000, 4, 0), (2500000, 4, 2), (3900000, 3, 0), (2000000, 6, 2)]:
    c.set_option("mgs_mode", mode)
    rng = np.random.default_rng(n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = Q @ rng.standard_normal(m) * 3 + rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, c)
    for j in range(m): B.upload(j, Q[:, j])
    B.length = m
    cols = [Q[:, j].copy() for j in range(m)]
    for name, dev, ref in (("mgs", kk.ModifiedGramSchmidt(), ko.MGS), ("mgs2", kk.ModifiedGramSchmidt2(), ko.MGS2)):
        wo, xo = ko.orthogonalize(w.copy(), cols, ref)
        for rep in range(4):
            x, nrm, _ = B.orthogonalize(B[m].set(w), dev)
            got = B[m].get()
            bad = np.nonzero(np.abs(got - wo) > 1e-9)[0]
            print(n, m, mode, name, rep, "xerr", float(np.max(np.abs(x - xo))), "nbad", bad.size, "nrmerr", abs(nrm - np.linalg.norm(wo)), flush=True)
            if bad.size:
                G, PT = 256, 512
                for i in bad[:24]:
                    e = i // 2
                    j = e // (G * PT); b = (e % (G * PT)) // PT; t = e % PT
                    print("   idx", int(i), "gridrow", int(j), "block", int(b), "thread", int(t), "got", got[i], "want", wo[i], "w_in", w[i])
    B.free()
