#!/usr/bin/env python3
"""Build-container-only: EVALUATES the reference's own source text — the body of CartPoleEnv.Step and the constant
initialisers of src/Gym.Environments/Envs/Classic/CartPoleEnv.cs:24-36,137-186 under /root/reference — statement by
statement, and so produces input -> output vectors whose formulas were never typed by this repository's author.

The reference is C# and no .NET runtime exists in this image, so this is NOT an execution of the reference: it is a small
interpreter (this file) for exactly the language subset that method uses —
    expressions   numeric literals (`1.0f` is binary32, `4.0` binary64, `12` int), names, unary - and !, * / + -,
                  < > <= >= == !=, ||, &&, `c ? a : b`, parentheses, Math.Cos / Math.Sin / Math.PI, `(float)` / `(int)` casts
    statements    `var v = e;`, `v = e;`, `v += e;`, `float v;`, `if (e) {...} else if (e) {...} else {...}`
— with C#'s numeric promotion implemented explicitly (float op float stays binary32; anything with a double is binary64;
`const float` initialisers fold in binary32; `(float)(double expression)` rounds once).  Those typing rules are the residual
assumption; everything else (operands, operators, association, the order of the statements, which integrator branch the
`if` selects, the reward / steps_beyond_done machine) comes from the reference's text at run time.  Nothing of the reference
is stored here, and the text — untrusted public content — is tokenised and parsed, never handed to eval().

    python oracle/evaluate_reference_text.py            # self-check on a few states, prints the parsed statement list

tests/golden/make_reference_text_golden.py uses it to generate tests/golden/cartpole_reference_text.npz (committed).
"""
import math
import os
import re
import sys

import numpy as np

REL = "src/Gym.Environments/Envs/Classic/CartPoleEnv.cs"
TOKEN = re.compile(r"\s*(\d+\.\d*(?:[eE][-+]?\d+)?[fF]?|\d+[fF]?|[A-Za-z_][A-Za-z_0-9.]*|==|!=|<=|>=|\|\||&&|\+=|[-+*/()<>?:!={};,])")
F32, F64, INT, BOOL = "float", "double", "int", "bool"


class Value:
    __slots__ = ("v", "t")

    def __init__(self, v, t):
        self.v, self.t = v, t


def _tokens(text):
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise ValueError(f"cannot tokenize {text[pos:pos + 40]!r}")
        out.append(m.group(1))
        pos = m.end()
    return out


def _promote(a, b):
    if F64 in (a.t, b.t):
        return F64
    if F32 in (a.t, b.t):
        return F32
    if a.t == b.t == INT:
        return INT
    raise ValueError(f"no numeric promotion for {a.t} and {b.t}")


def _as(v, t):
    if t == F64:
        return float(v.v)
    if t == F32:
        return np.float32(v.v)
    return int(v.v)


class Parser:
    """Recursive descent over a token list; evaluates while parsing (no AST kept: every statement runs exactly once)."""

    def __init__(self, toks, env):
        self.t, self.i, self.env = toks, 0, env

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def take(self, want=None):
        tok = self.peek()
        if tok is None or (want is not None and tok != want):
            raise ValueError(f"expected {want!r}, found {tok!r} at token {self.i}")
        self.i += 1
        return tok

    # ---- expressions ------------------------------------------------------------------------
    def expr(self):
        c = self.lor()
        if self.peek() == "?":
            self.take("?")
            a = self.expr()
            self.take(":")
            b = self.expr()
            if c.t != BOOL:
                raise ValueError("condition of ?: is not bool")
            t = a.t if a.t == b.t else _promote(a, b)
            pick = a if c.v else b
            return Value(_as(pick, t), t)
        return c

    def lor(self):
        a = self.land()
        while self.peek() == "||":
            self.take()
            b = self.land()
            a = Value(bool(a.v) or bool(b.v), BOOL)            # no side effects in operands: short-circuit is unobservable
        return a

    def land(self):
        a = self.cmp()
        while self.peek() == "&&":
            self.take()
            b = self.cmp()
            a = Value(bool(a.v) and bool(b.v), BOOL)
        return a

    def cmp(self):
        a = self.add()
        if self.peek() in ("<", ">", "<=", ">=", "==", "!="):
            op = self.take()
            b = self.add()
            if a.t == BOOL or b.t == BOOL:
                x, y = a.v, b.v
            else:
                t = _promote(a, b)
                x, y = _as(a, t), _as(b, t)
            return Value({"<": x < y, ">": x > y, "<=": x <= y, ">=": x >= y, "==": x == y, "!=": x != y}[op], BOOL)
        return a

    def add(self):
        a = self.mul()
        while self.peek() in ("+", "-"):
            op = self.take()
            b = self.mul()
            t = _promote(a, b)
            x, y = _as(a, t), _as(b, t)
            a = Value(x + y if op == "+" else x - y, t)
        return a

    def mul(self):
        a = self.unary()
        while self.peek() in ("*", "/"):
            op = self.take()
            b = self.unary()
            t = _promote(a, b)
            x, y = _as(a, t), _as(b, t)
            if op == "*":
                a = Value(x * y, t)
            elif t == INT:
                a = Value(int(x / y), INT)                       # C# integer division truncates toward zero
            else:
                a = Value(x / y, t)
        return a

    def unary(self):
        if self.peek() == "-":
            self.take()
            a = self.unary()
            return Value(-a.v, a.t)
        if self.peek() == "!":
            self.take()
            a = self.unary()
            return Value(not a.v, BOOL)
        if self.peek() == "(" and self.i + 2 < len(self.t) and self.t[self.i + 1] in ("float", "double", "int") and self.t[self.i + 2] == ")":
            self.take("(")
            to = self.take()
            self.take(")")
            a = self.unary()
            t = {"float": F32, "double": F64, "int": INT}[to]
            return Value(_as(a, t), t)
        return self.primary()

    def primary(self):
        tok = self.take()
        if tok == "(":
            a = self.expr()
            self.take(")")
            return a
        if re.fullmatch(r"\d+\.\d*(?:[eE][-+]?\d+)?[fF]|\d+[fF]", tok):
            return Value(np.float32(tok[:-1]), F32)
        if re.fullmatch(r"\d+\.\d*(?:[eE][-+]?\d+)?", tok):
            return Value(float(tok), F64)
        if re.fullmatch(r"\d+", tok):
            return Value(int(tok), INT)
        if tok in ("Math.Cos", "Math.Sin"):
            self.take("(")
            a = self.expr()
            self.take(")")
            x = float(a.v)                                      # Math.Cos / Math.Sin take and return double
            if not math.isfinite(x):                            # .NET returns NaN for +-Infinity and NaN (Python raises)
                return Value(float("nan"), F64)
            return Value(math.cos(x) if tok == "Math.Cos" else math.sin(x), F64)
        if tok == "Math.PI":
            return Value(math.pi, F64)
        if tok in self.env:
            return self.env[tok]
        raise ValueError(f"unknown name {tok!r}")

    # ---- statements -------------------------------------------------------------------------
    def block(self, run):
        self.take("{")
        while self.peek() != "}":
            self.statement(run)
        self.take("}")

    def skip_expr_parens(self):
        depth = 0
        while True:
            tok = self.take()
            depth += tok == "("
            depth -= tok == ")"
            if depth == 0:
                return

    def statement(self, run):
        tok = self.peek()
        if tok == "if":
            self.take()
            if run:
                self.take("(")
                c = self.expr()
                self.take(")")
                if c.t != BOOL:
                    raise ValueError("if condition is not bool")
                taken = bool(c.v)
            else:
                self.skip_expr_parens()
                taken = False
            self.block(run and taken)
            if self.peek() == "else":
                self.take()
                if self.peek() == "if":
                    self.statement(run and not taken)
                else:
                    self.block(run and not taken)
            return
        if tok in ("float", "double", "int", "bool") and self.t[self.i + 2] == ";":       # declaration without initialiser
            self.take(); self.take(); self.take(";")
            return
        declared = None
        if tok in ("var", "float", "double", "int", "bool"):
            declared = self.take()
        name = self.take()
        op = self.take()
        if op not in ("=", "+="):
            raise ValueError(f"unsupported statement at {name} {op}")
        if run:
            val = self.expr()
            if op == "+=":
                cur = self.env[name]
                t = _promote(cur, val)
                val = Value(_as(Value(_as(cur, t) + _as(val, t), t), cur.t), cur.t)
            elif declared in ("float", "double", "int"):
                t = {"float": F32, "double": F64, "int": INT}[declared]
                val = Value(_as(val, t), t)
            elif declared is None and name in self.env and self.env[name].t != val.t and self.env[name].t != BOOL:
                val = Value(_as(val, self.env[name].t), self.env[name].t)     # assignment converts to the variable's type
            self.env[name] = val
        else:
            while self.peek() != ";":
                self.take()
        self.take(";")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def load_reference(root="/root/reference"):
    path = os.path.join(root, REL)
    if not os.path.exists(path):
        return None
    return _strip_comments(open(path, encoding="utf-8-sig").read())


def reference_constants(text):
    """{name: Value} of the `private const float` / `const string` members, folded in declaration order (C# folds a
    float-typed constant expression in binary32)."""
    env = {}
    for typ, name, init in re.findall(r"private const (float|string) (\w+) = ([^;]+);", text):
        if typ == "string":
            env[name] = Value(init.strip().strip('"'), "string")
            continue
        v = Parser(_tokens(init), env).expr()
        env[name] = Value(np.float32(v.v), F32)
    return env


def step_statements(text):
    """Token list of the body of Step(object action) between the state reads and `return new Step(`, with the things that carry
    no arithmetic removed: string literals, Debug.Assert / Console.WriteLine calls, the np.array repacking of the state."""
    body = text[text.index("public override Step Step(object action)"):]
    body = body[body.index("{") + 1:body.index("return new Step(")]
    body = re.sub(r'\$?"(\\.|[^"\\])*"', "STR", body)
    body = re.sub(r"(Debug\.Assert|Console\.WriteLine)\s*\((?:[^()]|\([^()]*\))*\)\s*;", "", body)
    body = re.sub(r"state\s*=\s*np\.array\([^;]*\);", "", body)
    body = re.sub(r"var (\w+) = state\.GetDouble\(\d\);", "", body)          # x, x_dot, theta, theta_dot: the inputs
    body = re.sub(r"int iaction = \(int\)\s*action;", "", body)              # the unboxed action: an input
    body = re.sub(r"if \(steps_beyond_done == 0\) \{\s*\}", "", body)        # the warning branch, empty once the call is gone
    return body


def state_read_order(text):
    """The names bound to state.GetDouble(0..3), in index order (so x / x_dot / theta / theta_dot come from the text too)."""
    body = text[text.index("public override Step Step(object action)"):]
    pairs = re.findall(r"var (\w+) = state\.GetDouble\((\d)\);", body)
    return [n for n, _ in sorted(pairs, key=lambda p: int(p[1]))]


def written_state_order(text):
    body = text[text.index("public override Step Step(object action)"):]
    m = re.search(r"state\s*=\s*np\.array\(([^;]*)\);", body)
    return [a.strip() for a in m.group(1).split(",")]


class ReferenceText:
    def __init__(self, root="/root/reference"):
        text = load_reference(root)
        if text is None:
            raise FileNotFoundError(os.path.join(root, REL))
        self.constants = reference_constants(text)
        self.reads = state_read_order(text)
        self.writes = written_state_order(text)
        body = step_statements(text)
        # `kinematics_integrator == "euler"`: both sides are strings; compare as the text does
        body = body.replace("STR", "QUOTED_STRING")
        self.integrator_literal = re.search(r'if \(kinematics_integrator == "(\w+)"\)', text).group(1)
        self.tokens = _tokens(body)

    def step(self, state, action, steps_beyond_done):
        """One CartPoleEnv.Step on ONE instance: state (4 binary64), boxed int action, the private steps_beyond_done.
        Returns (new_state[4] float64, reward float32, done bool, steps_beyond_done int)."""
        env = dict(self.constants)
        env["QUOTED_STRING"] = Value(self.integrator_literal, "string")
        for name, v in zip(self.reads, state):
            env[name] = Value(float(v), F64)
        env["iaction"] = Value(int(action), INT)
        env["steps_beyond_done"] = Value(int(steps_beyond_done), INT)
        p = Parser(self.tokens, env)
        # string comparison for the integrator switch
        orig_cmp = p.cmp

        def cmp_with_strings():
            if p.peek() == "kinematics_integrator":
                a = p.take(); p.take("=="); b = p.take()
                return Value(env[a].v == env[b].v, BOOL)
            return orig_cmp()
        p.cmp = cmp_with_strings
        while p.peek() is not None:
            p.statement(True)
        out = [float(env[n].v) for n in self.writes]
        assert env["reward"].t == F32 and env["done"].t == BOOL and all(env[n].t == F64 for n in self.writes)
        return np.array(out, np.float64), np.float32(env["reward"].v), bool(env["done"].v), int(env["steps_beyond_done"].v)


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    try:
        ref = ReferenceText(root)
    except FileNotFoundError as e:
        print(f"reference not present at {e}: nothing to evaluate (build container only)")
        sys.exit(2)
    print("constants:", {k: (float(v.v) if v.t != "string" else v.v) for k, v in ref.constants.items()})
    print("reads:", ref.reads, " writes:", ref.writes)
    print("statements:", " ".join(ref.tokens)[:900])
    print(ref.step([0.0, 0.0, 0.0, 0.0], 1, -1))
    print(ref.step([2.39, 3.0, 0.0, 0.0], 0, -1), ref.step([2.39, 3.0, 0.0, 0.0], 0, 0))
