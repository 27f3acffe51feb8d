import sys; sys.path.insert(0,"krylovkit.jl_amd"); sys.path.insert(0,"oracle")
import numpy as np, krylovkit_hip as kk, krylov_oracle as ko
c = kk.Context(0); c.set_option("panel_min_rows", 0)
G, PT = 256, 512
def where(i):
    e = i // 2
    return (int(e // (G * PT)), int((e % (G * PT)) // PT), int(e % PT), "xy"[i % 2])
for (n, m, mode, reps) in [(2500000, 4, 0, 150), (1200000, 5, 2, 150), (3900000, 3, 0, 100)]:
    c.set_option("mgs_mode", mode)
    rng = np.random.default_rng(n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = Q @ rng.standard_normal(m) * 3 + rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, c)
    for j in range(m): B.upload(j, Q[:, j])
    B.length = m
    cols = [Q[:, j].copy() for j in range(m)]
    wo, xo = ko.orthogonalize(w.copy(), cols, ko.MGS)
    look = {}
    for name, arr in [("wo", wo), ("w", w)] + [(f"q{j}", Q[:, j]) for j in range(m)]:
        for k in range(0, n, 1):
            pass
    B[m].set(w)
    events = 0
    for rep in range(reps):
        B[m].copy_from(B[m]) if False else None
        B[m].set(w)
        x, nrm, _ = B.orthogonalize(B[m], kk.ModifiedGramSchmidt())
        got = B[m].get()
        bad = np.nonzero(np.abs(got - wo) > 1e-9)[0]
        if bad.size:
            events += 1
            print(n, mode, "rep", rep, "nbad", bad.size, "xerr", float(np.max(np.abs(x - xo))), "first", where(bad[0]), "last", where(bad[-1]), flush=True)
            lanes = sorted(set((where(i)[2] % 64) for i in bad)); print("   lanes", lanes, "waves", sorted(set(where(i)[2] // 64 for i in bad)), "rows", sorted(set(where(i)[0] for i in bad)), "blocks", sorted(set(where(i)[1] for i in bad)), "comp", sorted(set(where(i)[3] for i in bad)))
            for i in bad[:6]:
                g = got[i]
                src = []
                for nm, arr in [("wo", wo), ("w", w)] + [(f"q{j}", Q[:, j]) for j in range(m)]:
                    k = np.nonzero(arr == g)[0]
                    if k.size: src.append((nm, [where(int(kk_)) for kk_ in k[:3]]))
                # linear combination check: got = w - sum s_j q_j for a prefix of j (partially updated)?
                part = [w[i] - sum(x[jj] * Q[i, jj] for jj in range(t)) for t in range(m + 1)]
                print("     idx", int(i), where(i), "got", g, "want", wo[i], "exact-src", src, "prefix-updates", [float(p) for p in part])
    print(n, mode, "events", events, "of", reps, flush=True)
    B.free()
