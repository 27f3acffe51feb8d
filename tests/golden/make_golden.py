"""Regenerates tests/golden/*.npy from the reference's own test fixtures (run in the build
container, where /root/reference is mounted; the GPU box never needs it).

issue143_A.npy : the 71x71 symmetric matrix literal embedded in the reference's regression test
                 for issue #143 (test/issues.jl:40-112) -- data, not code.
"""
import re
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/test/issues.jl")
OUT = Path(__file__).resolve().parent


def main():
    lines = REF.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if "Issue #143" in l and "@testset" in l)
    a0 = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("A = ["))
    a1 = next(i for i in range(a0, len(lines)) if lines[i].strip() == "]")
    rows = []
    for l in lines[a0 + 1: a1]:
        vals = [float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", l)]
        if vals:
            rows.append(vals)
    A = np.array(rows)
    assert A.shape == (71, 71), A.shape
    assert np.max(np.abs(A - A.T)) < 1e-6 * np.max(np.abs(A))
    np.save(OUT / "issue143_A.npy", A)
    print("issue143_A.npy", A.shape, "asym", np.max(np.abs(A - A.T)))


if __name__ == "__main__":
    sys.exit(main())
