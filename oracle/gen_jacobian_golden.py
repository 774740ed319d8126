#!/usr/bin/env python3
"""Generate tests/golden/jacobian_h.json from the reference's own arithmetic.

TEST INFRASTRUCTURE.  Run in the authoring container only (needs
/root/reference); the output fixture is what travels.

The reference writes the per-feature measurement Jacobian as two comma
initialisers of closed-form expressions (src/slam/main.cpp:604-617 `Hb`,
:619-623 `Hi`).  Those expressions use only + - * /, parentheses, pow(x, 2) and
the 18 scalar inputs named at :583-601, so each one is also a valid Python
expression.  This script reads that text at run time (nothing of it is stored
in this repository), splits it at top-level commas and evaluates every entry
with math.pow on seeded random inputs.  The fixture holds inputs and the 24+12
resulting doubles per case as C99 hex floats.
"""
import json
import math
import random
import sys

REF = "/root/reference/src/slam/main.cpp"


def top_level_split(text):
    out, depth, cur = [], 0, []
    for ch in text:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return [e.strip() for e in out]


def load_exprs():
    lines = open(REF).read().split("\n")
    # 1-based :604-617 and :619-623
    hb_txt = "\n".join(lines[603:617])
    hi_txt = "\n".join(lines[618:623])

    def body(txt):
        start = txt.index("<<") + 2
        end = txt.rindex(");")
        return txt[start:end]

    hb = top_level_split(body(hb_txt))
    hi = top_level_split(body(hi_txt))
    assert len(hb) == 24 and len(hi) == 12, (len(hb), len(hi))
    return [compile(e, "Hb", "eval") for e in hb], [compile(e, "Hi", "eval") for e in hi]


def unit(rng, scale=1.0):
    q = [rng.gauss(0, 1) for _ in range(4)]
    nrm = math.sqrt(sum(x * x for x in q))
    return [scale * x / nrm for x in q]


def small_rot(rng, ang):
    # scalar-last quaternion of a small rotation
    ax = [rng.gauss(0, 1) for _ in range(3)]
    nrm = math.sqrt(sum(x * x for x in ax))
    a = rng.uniform(-ang, ang)
    s = math.sin(a / 2) / nrm
    return [ax[0] * s, ax[1] * s, ax[2] * s, math.cos(a / 2)]


def main():
    hb, hi = load_exprs()
    rng = random.Random(20240607)
    cases = []
    for c in range(96):
        X = [3 * rng.gauss(0, 1) for _ in range(3)]
        if c % 3 == 0:
            qbw = small_rot(rng, 0.2)
            q0 = small_rot(rng, 0.2)
        elif c % 3 == 1:
            qbw = unit(rng)
            q0 = unit(rng)
        else:
            qbw = unit(rng, 1.001)          # the reference never normalises
            q0 = unit(rng)
        anchor = [x + rng.gauss(0, 1) for x in X]
        pib = [0.3 * rng.gauss(0, 1), 0.3 * rng.gauss(0, 1), rng.uniform(0.01, 1.0)]
        env = {
            "pow": math.pow,
            "xbw1": X[0], "xbw2": X[1], "xbw3": X[2],
            "qbw1": qbw[0], "qbw2": qbw[1], "qbw3": qbw[2], "qbw4": qbw[3],
            "pib1": pib[0], "pib2": pib[1], "pib3": pib[2],
            "xb0w1": anchor[0], "xb0w2": anchor[1], "xb0w3": anchor[2],
            "qb0w1": q0[0], "qb0w2": q0[1], "qb0w3": q0[2], "qb0w4": q0[3],
        }
        Hb = [float(eval(e, {"__builtins__": {}}, env)) for e in hb]
        Hi = [float(eval(e, {"__builtins__": {}}, env)) for e in hi]
        h = lambda v: [float(x).hex() for x in v]
        cases.append({"X": h(X), "qbw": h(qbw), "pib": h(pib), "anchor": h(anchor),
                      "q0": h(q0), "Hb": h(Hb), "Hi": h(Hi)})
    out = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/jacobian_h.json"
    with open(out, "w") as fp:
        json.dump({"source": "values of src/slam/main.cpp:604-623 evaluated by oracle/gen_jacobian_golden.py",
                   "layout": "Hb 4x6 row-major, Hi 4x3 row-major, quaternions scalar-last",
                   "cases": cases}, fp, separators=(",", ":"))
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
