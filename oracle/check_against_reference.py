#!/usr/bin/env python3
"""Build-container-only guard: the C restatement (oracle/classic_control_ref.c: ref_cartpole_step_f64 and the CP_*
constants) against the TEXT of the reference it claims to follow, src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:24-36,
63-67,137-186 under /root/reference.

The reference is C# and cannot be built or run here (no .NET toolchain), so parity stays "unpinned" — this script does
NOT change that.  What it removes is human transcription as a source of error: both files are parsed, every assignment of
the step (force, costheta, sintheta, temp, thetaacc, xacc, the four Euler updates, done, the reward / steps_beyond_done
machine) is reduced to a token stream under a fixed renaming (Math.Cos -> cos, CP_GRAVITY -> gravity, `(double)` widening
casts dropped, ...), and the two streams must be IDENTICAL — same operands, same operators, same association (parentheses
are tokens).  The float32 constants are evaluated from the C# initialisers in binary32 and compared with what the built
oracle reports.  Nothing of the reference is stored in this repository: the script reads it where it lies and fails loudly
if it ever differs.  Exit status 0 = identical, 1 = mismatch, 2 = reference tree absent (e.g. on the GPU box).

    python oracle/check_against_reference.py [/root/reference]
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REL = "src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"
TOKEN = re.compile(r"\s*(\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\d+[fF]?|[A-Za-z_][A-Za-z_0-9.]*|==|!=|<=|>=|\|\||&&|\+=|[-+*/()<>?:!=])")

# reference name -> canonical name
CS_RENAME = {"Math.Cos": "cos", "Math.Sin": "sin", "iaction": "action", "theta_threshold_radians": "theta_threshold"}
# restatement name -> canonical name
C_RENAME = {"CP_GRAVITY": "gravity", "CP_MASSCART": "masscart", "CP_MASSPOLE": "masspole", "CP_TOTAL_MASS": "total_mass",
            "CP_LENGTH": "length", "CP_POLEMASS_LENGTH": "polemass_length", "CP_FORCE_MAG": "force_mag", "CP_TAU": "tau",
            "CP_THETA_THRESHOLD": "theta_threshold", "CP_X_THRESHOLD": "x_threshold", "sbd": "steps_beyond_done"}
STEP_VARS = ["force", "costheta", "sintheta", "temp", "thetaacc", "xacc", "x", "x_dot", "theta", "theta_dot", "done"]


def tokens(expr, rename):
    out, pos = [], 0
    expr = expr.strip()
    while pos < len(expr):
        m = TOKEN.match(expr, pos)
        if not m:
            raise ValueError(f"cannot tokenize {expr[pos:pos + 30]!r}")
        t = m.group(1)
        pos = m.end()
        out.append(rename.get(t, t))
    return out


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def reference_step(text):
    """{var: token list} for the Step() body of the C# reference, 'euler' branch, plus the reward machine as a list."""
    text = strip_comments(text)
    integ = re.search(r'const string kinematics_integrator = "(\w+)"', text).group(1)
    which = re.search(r'if \(kinematics_integrator == "(\w+)"\)', text).group(1)          # the branch the `if` selects
    text = re.sub(r'\$?"(\\.|[^"\\])*"', '""', text)                                    # string literals carry no arithmetic
    body = text[text.index("public override Step Step(object action)"):]
    body = body[:body.index("return new Step(")]
    m = re.search(r'if \(kinematics_integrator == ""\) \{(.*?)\} else \{(.*?)\}', body, flags=re.S)
    branch = m.group(1) if integ == which else m.group(2)
    head, tail = body[:m.start()], body[m.end():]
    stmts = {}
    for src in (head, branch, tail):
        for name, expr in re.findall(r"(?:var\s+)?([A-Za-z_]\w*)\s*=\s*([^;{}]+);", src):
            if name in STEP_VARS and name not in ("x", "x_dot", "theta", "theta_dot") or (src is branch and name in STEP_VARS):
                stmts.setdefault(name, tokens(expr, CS_RENAME))
    machine = re.sub(r"Console\.WriteLine\(.*?\);", "", tail[tail.index("float reward;"):], flags=re.S)
    machine = re.sub(r"if \(steps_beyond_done == 0\) \{\s*\}", "", machine)          # the warning branch, now empty
    return stmts, tokens(re.sub(r"[{};]", " ", machine.replace("float reward", "")), CS_RENAME), integ


def restatement_step(text):
    text = strip_comments(text)
    body = text[text.index("int ref_cartpole_step_f64("):]
    body = body[:body.index("return done;")]
    body = body.replace("(double)", "")                      # the implicit float -> double widening of the C# constants
    body = re.sub(r"\*(sbd|reward)\b", r"\1", body)
    stmts = {}
    first = re.search(r"double x = state\[0\], x_dot = state\[1\], theta = state\[2\], theta_dot = state\[3\];", body)
    rest = body[first.end():]
    for name, expr in re.findall(r"(?:double|float|int)?\s*\b([A-Za-z_]\w*)\s*=\s*([^;{}]+);", rest):
        if name in STEP_VARS:
            stmts.setdefault(name, tokens(expr, C_RENAME))
    machine = rest[rest.index("if (!done)"):]
    return stmts, tokens(re.sub(r"[{};]", " ", machine), C_RENAME)


def _arith(expr):
    """Value (binary64) of a constant arithmetic expression taken from the reference text: numeric literals, Math.PI,
    + - * /, unary minus and parentheses — and NOTHING else.  The text is untrusted public content, so it is never
    handed to eval(): it is parsed and walked, and any other construct (a name, a call, an attribute, a subscript ...)
    raises."""
    import ast
    import operator
    ops = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv}
    tree = ast.parse(expr.strip().replace("Math.PI", "PI"), mode="eval")

    def walk(node):
        if isinstance(node, ast.Expression):
            return walk(node.body)
        if isinstance(node, ast.Constant) and type(node.value) in (int, float):
            return float(node.value)
        if isinstance(node, ast.Name) and node.id == "PI":
            return float(np.pi)
        if isinstance(node, ast.BinOp) and type(node.op) in ops:
            return ops[type(node.op)](walk(node.left), walk(node.right))
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = walk(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        raise ValueError(f"unsupported construct in a constant initialiser: {ast.dump(node)[:80]}")
    return walk(tree)


def reference_constants(text):
    """The C# `const float` initialisers evaluated in binary32 (C# folds float-typed constant expressions in float)."""
    text = strip_comments(text)
    vals = {}
    for name, expr in re.findall(r"private const float (\w+) = ([^;]+);", text):
        e = expr.strip()
        if e.startswith("(float)"):                           # (float) (12 * 2 * Math.PI / 360): double arithmetic, one cast
            vals[name] = np.float32(_arith(e[len("(float)"):]))
            continue
        toks = tokens(e, {})
        acc, op = None, None
        for t in toks:                                         # the initialisers are `lit` or `a op b`
            if t in "+-*/":
                op = t
                continue
            v = np.float32(t.rstrip("fF")) if re.match(r"\d", t) else vals[t]
            acc = v if acc is None else {"+": acc + v, "-": acc - v, "*": acc * v, "/": acc / v}[op]
        vals[name] = np.float32(acc)
    return vals


def main(root="/root/reference"):
    path = os.path.join(root, REL)
    if not os.path.exists(path):
        print(f"reference not present at {path}: nothing to check (this guard only runs in the build container)")
        return 2
    cs = open(path, encoding="utf-8-sig").read()
    c = open(os.path.join(HERE, "classic_control_ref.c")).read()
    bad = []
    ref, ref_machine, integ = reference_step(cs)
    mine, my_machine = restatement_step(c)
    if integ != "euler":
        bad.append(f"reference integrator is {integ!r}; the restatement (and the kernels) implement explicit Euler")
    for v in STEP_VARS:
        if v not in ref or v not in mine:
            bad.append(f"{v}: statement not found (reference: {v in ref}, restatement: {v in mine})")
        elif ref[v] != mine[v]:
            bad.append(f"{v}:\n    reference   {' '.join(ref[v])}\n    restatement {' '.join(mine[v])}")
    if ref_machine != my_machine:
        bad.append("reward / steps_beyond_done machine:\n    reference   " + " ".join(ref_machine) + "\n    restatement " + " ".join(my_machine))
    # Reset(): steps_beyond_done = -1; state = uniform(-0.05, 0.05, 4)
    rs = re.search(r"public override NDArray Reset\(\) \{(.*?)\}", strip_comments(cs), flags=re.S).group(1)
    if not re.search(r"steps_beyond_done\s*=\s*-1\s*;", rs) or not re.search(r"uniform\(\s*-0\.05\s*,\s*0\.05\s*,\s*4\s*\)", rs):
        bad.append("Reset(): expected steps_beyond_done = -1 and uniform(-0.05, 0.05, 4), found: " + " ".join(rs.split()))
    # constants: reference initialisers in binary32 == what the BUILT oracle reports
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import capi
    capi.build()
    got = capi.cartpole_constants()
    names = ["gravity", "masscart", "masspole", "total_mass", "length", "polemass_length", "force_mag", "tau",
             "theta_threshold_radians", "x_threshold"]
    want = reference_constants(cs)
    for nme in names:
        if nme not in want:
            bad.append(f"constant {nme}: not found in the reference")
        elif float(want[nme]) != float(got[nme]):
            bad.append(f"constant {nme}: reference {float(want[nme])!r} != restatement {float(got[nme])!r}")
    if bad:
        print("MISMATCH between the reference text and oracle/classic_control_ref.c:\n  " + "\n  ".join(bad))
        return 1
    print(f"identical: {len(STEP_VARS)} step assignments, the reward machine, Reset() and {len(names)} constants "
          f"({REL} vs oracle/classic_control_ref.c)")
    return 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:2]))
