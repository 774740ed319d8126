#!/usr/bin/env python3
"""Generate tests/golden/motion_model.json from the reference's own arithmetic.

TEST INFRASTRUCTURE.  Run in the authoring container only (needs
/root/reference); the output fixture is what travels.

The reference writes the motion model as comma initialisers of closed-form
expressions:
  src/slam/main.cpp:253-264   Fb1     6 x 6   rotation block of the motion Jacobian
  src/slam/main.cpp:290-299   FiTemp  3 x 3   d(feature rate)/d(feature), one per feature
  src/slam/main.cpp:314-317   Fib_ith 3 x 6   d(feature rate)/d(X, V), one per feature
  src/ARC/states.cpp:253-256  the three components of the feature rate in States::dynamics
Those expressions use only + - * /, parentheses, pow(x, 2), and indexed scalars
(qbw.coord[i], w[i], mu.V[i] / V[i], bp[i], pib1..3), so after renaming the
containers each one is a valid Python expression.  This script reads that text
at run time (nothing of it is stored in this repository), splits it at
top-level commas and evaluates every entry on seeded random inputs.  The
fixture holds the inputs and the resulting doubles per case as C99 hex floats.
"""
import json
import math
import random
import re
import sys

MAIN = "/root/reference/src/slam/main.cpp"
STATES = "/root/reference/src/ARC/states.cpp"


def top_level_split(text):
    out, depth, cur = [], 0, []
    for ch in text:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return [e.strip() for e in out]


def initialiser(path, first, last, opener):
    """Entries of the comma list between `opener` on line `first` and the closing ');' on line `last` (1-based)."""
    txt = "\n".join(open(path).read().split("\n")[first - 1:last])
    body = txt[txt.index(opener) + len(opener):txt.rindex(");")]
    body = re.sub(r"//[^\n]*", "", body).replace("\n", " ").strip()
    if body.count(")") > body.count("("):            # the list's own closing parenthesis (nested call)
        body = body[:body.rindex(")")]
    body = body.replace("qbw.coord", "qbw").replace("mu.V", "V")
    return [compile(e, "%s:%d" % (path.rsplit("/", 1)[1], first), "eval") for e in top_level_split(body)]


def main():
    fb1 = initialiser(MAIN, 253, 264, "<<")
    fitemp = initialiser(MAIN, 290, 299, "<<")
    fib = initialiser(MAIN, 314, 317, "<<")
    rate = initialiser(STATES, 253, 257, "cv::Vec3d(")
    assert (len(fb1), len(fitemp), len(fib), len(rate)) == (36, 9, 18, 3), (len(fb1), len(fitemp), len(fib), len(rate))
    rng = random.Random(20240611)
    h = lambda v: [float(x).hex() for x in v]
    ev = lambda exprs, env: [float(eval(e, {"__builtins__": {}}, env)) for e in exprs]
    cases = []
    for c in range(48):
        q = [rng.gauss(0, 1) for _ in range(4)]
        nrm = math.sqrt(sum(x * x for x in q)) / (1.0 if c % 2 == 0 else 1.002)     # the reference never normalises
        q = [x / nrm for x in q]
        w = [0.3 * rng.gauss(0, 1) for _ in range(3)]
        V = [rng.gauss(0, 1) for _ in range(3)]
        nf = 1 + c % 4
        feats = []
        for _ in range(nf):
            p = [0.4 * rng.gauss(0, 1), 0.4 * rng.gauss(0, 1), rng.uniform(0.005, 1.0)]
            env = {"pow": math.pow, "w": w, "V": V, "pib1": p[0], "pib2": p[1], "pib3": p[2], "bp": p}
            feats.append({"pib": h(p), "FiTemp": h(ev(fitemp, env)), "Fib": h(ev(fib, env)), "rate": h(ev(rate, env))})
        env = {"pow": math.pow, "qbw": q, "w": w}
        cases.append({"qbw": h(q), "w": h(w), "V": h(V), "dt": float(rng.uniform(0.005, 0.05)).hex(),
                      "Fb1": h(ev(fb1, env)), "features": feats})
    out = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/motion_model.json"
    with open(out, "w") as fp:
        json.dump({"source": "values of src/slam/main.cpp:253-264,290-299,314-317 and src/ARC/states.cpp:253-256 "
                             "evaluated by oracle/gen_motion_golden.py",
                   "layout": "Fb1 6x6, FiTemp 3x3, Fib 3x6 row-major; rate = d(pib)/dt; quaternions scalar-last",
                   "cases": cases}, fp, separators=(",", ":"))
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
