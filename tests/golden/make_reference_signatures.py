"""Extracts the method signatures of the reference functions that julia/KrylovKitHIP.jl overloads into
tests/golden/reference_signatures.json (run in the build container, where /root/reference is mounted).  Signatures are
facts about the reference's interface (names, arities, keyword names, argument type heads) -- no code is copied.
tests/test_julia_shim_lint.py checks every overload of the shim against this table."""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent / "reference_signatures.json"
NAMES = ["initialize", "initialize!", "expand!", "shrink!", "orthogonalize!!", "orthonormalize!!", "project!!", "unproject!!",
         "rank1update!", "basistransform!", "block_qr!", "block_inner", "block_reorthogonalize!", "apply", "apply_normal",
         "apply_adjoint", "lanczosrecurrence", "arnoldirecurrence!!", "gklrecurrence", "block_lanczosrecurrence"]


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch in ",;" and depth == 0:
            out.append((cur.strip(), ch))
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append((cur.strip(), ""))
    return out


def parse_args(argstr):
    pos, kw, in_kw = [], [], False
    for tok, sep in split_top(argstr):
        if tok:
            name, _, rest = tok.partition("::")
            typ, _, default = rest.partition("=") if rest else ("", "", "")
            if not rest and "=" in name:
                name, _, default = name.partition("=")
            head = re.match(r"\s*([A-Za-z_.0-9]+)", typ)
            entry = {"name": name.strip(), "type": head.group(1) if head else "", "optional": bool(default.strip()) or "=" in tok}
            (kw if in_kw else pos).append(entry)
        if sep == ";":
            in_kw = True
    return pos, kw


def extract(text, name):
    sigs = []
    esc = re.escape(name)
    for m in re.finditer(rf"(?m)^(?:(?:Base\.)?@\w+\s+)?(?:function\s+)?(?:KrylovKit\.)?{esc}\(", text):
        i = m.end() - 1
        depth, j = 0, i
        while j < len(text):
            if text[j] == "(":
                depth += 1
            elif text[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        after = text[j + 1:j + 60]
        if "function" not in m.group(0) and not re.match(r"\s*(where\s*\{[^}]*\}\s*)?=", after):
            continue            # a call, not a definition
        pos, kw = parse_args(" ".join(text[i + 1:j].split()))
        sigs.append({"positional": pos, "keywords": [k["name"] for k in kw]})
    return sigs


def main():
    table = {}
    for f in sorted(REF.rglob("*.jl")):
        text = f.read_text()
        for name in NAMES:
            for s in extract(text, name):
                s["file"] = str(f.relative_to(REF.parent))
                table.setdefault(name, []).append(s)
    OUT.write_text(json.dumps(table, indent=1, ensure_ascii=False) + "\n")
    print({k: len(v) for k, v in table.items()})


if __name__ == "__main__":
    sys.exit(main())
