"""vsim.py -- a small cycle simulator for the Verilog subset that the reference
core generator emits (test infrastructure).

Purpose: a second, independent check of the oracle.  The oracle
(oracle/cordic_oracle.c) is a hand restatement of rtl/cordic.v & co.; this
module instead EXECUTES the Verilog text itself -- the files under
/root/reference/rtl, or what oracle/_ref/gencordic (the real generator, built
by oracle/Makefile) emits for any parameter set -- clock edge by clock edge,
the way the reference's Verilator benches do (bench/cpp/testb.h:87-106).  It is
NOT a reference build and is never used as a baseline: it is this project's
own reading of Verilog semantics, applied to the reference's own text, so a
transcription slip in the oracle (a sign in a case arm, a shift amount, the
rounding vector) shows up as a sample mismatch.

Subset: one module; localparam header; input/output/wire/reg declarations with
optional `signed`, a packed range and one unpacked array range; `assign`;
`initial`; `always @(posedge clk ...)`; `generate for`; begin/end, if/else,
case; blocking and non-blocking assignments; `always @(*)`; continuous
assignments to part selects of a wire; localparam items; expressions with
+ - * unary-, !, reduction & and |, && ||, == != < <= > >=, >> >>> <<,
concatenation, replication, bit and part selects, sized literals and
$signed().  Expressions are sized and signed as IEEE 1364-2005 5.4-5.5 say
(context width = the wider of the two sides of an assignment, an expression
is unsigned as soon as one operand is, shift amounts / concatenation operands /
comparison operand pairs are self-determined); tests/test_vsim_conformance.py
holds the cases with their values worked out from those rules by hand.
"""
import re

TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|`[^\n]*)
  | (?P<str>"[^"\n]*")
  | (?P<num>\d*'[sS]?[bBhHdD][0-9a-fA-F_xXzZ]+|\d+)
  | (?P<id>\$?[A-Za-z_][A-Za-z0-9_$]*)
  | (?P<op>>>>|<<<|<=|>=|==|!=|&&|\|\||>>|<<|[-+*/%!~&|^<>=?:;,.#@(){}\[\]])
""", re.X)


def tokenize(text):
    out, pos = [], 0
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise SyntaxError("bad character %r at %d" % (text[pos], pos))
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group(m.lastgroup)))
    return out


class Node:
    def __init__(self, kind, *args):
        self.kind, self.args = kind, args

    def __repr__(self):
        return "%s%r" % (self.kind, self.args)


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k][1] if self.i + k < len(self.t) else None

    def next(self):
        v = self.t[self.i][1]
        self.i += 1
        return v

    def expect(self, v):
        got = self.next()
        if got != v:
            raise SyntaxError("expected %r, got %r (token %d)" % (v, got, self.i))

    def accept(self, v):
        if self.peek() == v:
            self.i += 1
            return True
        return False

    # ---- expressions
    def expr(self):
        return self.binary(0)

    LEVELS = [["||"], ["&&"], ["==", "!="], ["<", "<=", ">", ">="],
              [">>>", ">>", "<<", "<<<"], ["+", "-"], ["*"]]

    def binary(self, lvl):
        if lvl == len(self.LEVELS):
            return self.unary()
        left = self.binary(lvl + 1)
        while self.peek() in self.LEVELS[lvl]:
            op = self.next()
            left = Node("bin", op, left, self.binary(lvl + 1))
        return left

    def unary(self):
        if self.peek() in ("-", "!", "~", "&", "|", "+"):
            op = self.next()
            return Node("un", op, self.unary())
        return self.primary()

    def primary(self):
        kind, v = self.t[self.i]
        if v == "(":
            self.next()
            e = self.expr()
            self.expect(")")
            return e
        if v == "{":
            self.next()
            first = self.expr()
            if self.peek() == "{":          # replication {n{x}}
                self.next()
                item = self.expr()
                self.expect("}")
                self.expect("}")
                return Node("rep", first, item)
            items = [first]
            while self.accept(","):
                items.append(self.expr())
            self.expect("}")
            return Node("cat", items)
        if kind == "num":
            self.next()
            return Node("num", v)
        if kind == "id":
            self.next()
            if v == "$signed":
                self.expect("(")
                e = self.expr()
                self.expect(")")
                return Node("signed", e)
            n = Node("id", v)
            while self.peek() == "[":
                self.next()
                a = self.expr()
                if self.accept(":"):
                    b = self.expr()
                    self.expect("]")
                    n = Node("part", n, a, b)
                else:
                    self.expect("]")
                    n = Node("idx", n, a)
            return n
        raise SyntaxError("unexpected %r" % v)

    # ---- statements
    def stmt(self):
        v = self.peek()
        if v == "begin":
            self.next()
            if self.accept(":"):
                self.next()
            body = []
            while self.peek() != "end":
                body.append(self.stmt())
            self.next()
            return Node("block", body)
        if v == "if":
            self.next()
            self.expect("(")
            c = self.expr()
            self.expect(")")
            a = self.stmt()
            b = self.stmt() if self.accept("else") else None
            return Node("if", c, a, b)
        if v == "case":
            self.next()
            self.expect("(")
            sel = self.expr()
            self.expect(")")
            arms = []
            while self.peek() != "endcase":
                if self.accept("default"):
                    self.accept(":")
                    arms.append((None, self.stmt()))
                else:
                    labels = [self.expr()]
                    while self.accept(","):
                        labels.append(self.expr())
                    self.expect(":")
                    arms.append((labels, self.stmt()))
            self.next()
            return Node("case", sel, arms)
        lhs = self.primary()
        op = self.next()
        if op not in ("<=", "="):
            raise SyntaxError("expected assignment, got %r" % op)
        rhs = self.expr()
        self.expect(";")
        return Node("nba" if op == "<=" else "ba", lhs, rhs)


def parse_literal(s):
    if "'" not in s:
        return int(s), 32, True
    w, rest = s.split("'")
    signed = rest[0] in "sS"
    if signed:
        rest = rest[1:]
    base = {"b": 2, "h": 16, "d": 10}[rest[0].lower()]
    val = int(rest[1:].replace("_", ""), base)
    width = int(w) if w else 32
    return val & ((1 << width) - 1), width, signed


class Module:
    """Parsed module + simulator state."""

    def __init__(self, text, readmem_dir=None):
        self.readmem_dir = readmem_dir
        self.params = {}
        self.decl = {}          # name -> (width, signed, array_len or None)
        self.assigns = {}       # wire name -> expr
        self.assign_parts = {}  # wire name -> [(hi, lo, expr, env)]
        self.comb = []          # always @(*) blocks
        self.always = []        # (stmt, genvar bindings)
        self.initials = []
        self.state = {}
        self._parse(tokenize(text))
        self.reset_state()

    # ---- constant expressions (parameters, genvars)
    def const(self, node, env=None):
        env = env or {}
        k = node.kind
        if k == "num":
            return parse_literal(node.args[0])[0]
        if k == "id":
            n = node.args[0]
            if n in env:
                return env[n]
            return self.params[n]
        if k == "bin":
            op, a, b = node.args
            a, b = self.const(a, env), self.const(b, env)
            return {"+": a + b, "-": a - b, "*": a * b, "<": int(a < b), "<=": int(a <= b),
                    ">": int(a > b), ">=": int(a >= b), "==": int(a == b),
                    "!=": int(a != b), "<<": a << b if op == "<<" else 0,
                    ">>": a >> b if op == ">>" else 0}[op]
        if k == "un":
            v = self.const(node.args[1], env)
            return -v if node.args[0] == "-" else v
        raise ValueError("not constant: %r" % (node,))

    def _parse(self, toks):
        p = Parser(toks)
        p.expect("module")
        self.name = p.next()
        if p.accept("#"):
            p.expect("(")
            while p.peek() != ")":
                if p.peek() in ("localparam", "parameter", ","):
                    p.next()
                    continue
                name = p.next()
                p.expect("=")
                self.params[name] = self.const(p.expr())
            p.next()
        p.expect("(")
        self._ports(p)
        p.expect(";")
        while p.peek() != "endmodule":
            self._item(p, {})
        # NSTAGES is commented out in the sequential cores' headers

    def _range(self, p):
        if p.peek() != "[":
            return None
        p.next()
        hi = self.const(p.expr())
        p.expect(":")
        lo = self.const(p.expr())
        p.expect("]")
        return hi, lo

    def _declare(self, p, terminators):
        signed = False
        while p.peek() in ("wire", "reg", "signed", "input", "output"):
            if p.next() == "signed":
                signed = True
        rng = self._range(p)
        width = (rng[0] - rng[1] + 1) if rng else 1
        while True:
            name = p.next()
            arr = self._range(p)
            alen = (abs(arr[1] - arr[0]) + 1) if arr else None
            self.decl[name] = (width, signed, alen)
            if p.peek() == "," and p.peek(1) not in (
                    "input", "output", "wire", "reg"):
                p.next()
                continue
            break

    def _ports(self, p):
        while p.peek() != ")":
            if p.peek() == ",":
                p.next()
                continue
            self._declare(p, (",", ")"))
        p.next()

    def _item(self, p, env):
        v = p.peek()
        if v in ("wire", "reg"):
            self._declare(p, (";",))
            p.expect(";")
        elif v == "localparam":
            p.next()
            while True:
                name = p.next()
                p.expect("=")
                self.params[name] = self.const(p.expr(), env)
                if not p.accept(","):
                    break
            p.expect(";")
        elif v == "genvar":
            p.next(); p.next(); p.expect(";")
        elif v == "assign":
            p.next()
            lhs = p.primary()
            p.expect("=")
            rhs = p.expr()
            p.expect(";")
            if lhs.kind == "id":
                self.assigns[lhs.args[0]] = (rhs, dict(env))
            elif lhs.kind == "part" or (lhs.kind == "idx" and self.decl[
                    lhs.args[0].args[0]][2] is None):
                # assign w[hi:lo] = value: the wire is the union of its pieces
                hi = self.const(lhs.args[1], env)
                lo = self.const(lhs.args[2], env) if lhs.kind == "part" else hi
                self.assign_parts.setdefault(lhs.args[0].args[0], []).append(
                    (hi, lo, rhs, dict(env)))
            else:                               # assign mem[k] = value
                name = lhs.args[0].args[0]
                idx = self.const(lhs.args[1], env)
                self.initials.append((Node("ba", lhs, rhs), dict(env)))
                _ = (name, idx)
        elif v == "initial":
            p.next()
            if p.peek() == "begin" and p.peek(1) == "$readmemh":
                p.next()
                while p.peek() == "$readmemh":
                    self._readmem(p)
                p.expect("end")
            elif p.peek() == "$readmemh":
                self._readmem(p)
            else:
                self.initials.append((p.stmt(), dict(env)))
        elif v == "always":
            p.next()
            p.expect("@")
            p.expect("(")
            depth, star = 1, False
            while depth:
                t = p.next()
                star = star or t == "*"
                depth += (t == "(") - (t == ")")
            (self.comb if star else self.always).append((p.stmt(), dict(env)))
        elif v == "generate":
            p.next()
            p.expect("for")
            p.expect("(")
            var = p.next(); p.expect("=")
            start = self.const(p.expr(), env); p.expect(";")
            cond = p.expr(); p.expect(";")
            p.next(); p.expect("="); step = p.expr(); p.expect(")")
            p.expect("begin")
            if p.accept(":"):
                p.next()
            body_start = p.i
            i = start
            while self.const(cond, dict(env, **{var: i})):
                p.i = body_start
                e2 = dict(env, **{var: i})
                while p.peek() != "end":
                    self._item(p, e2)
                i = self.const(step, e2)
            if i == start:                      # zero iterations: skip body
                depth = 1
                while depth:
                    t = p.next()
                    depth += (t == "begin") - (t == "end")
            else:
                p.expect("end")
            p.expect("endgenerate")
        else:
            raise SyntaxError("module item %r" % v)

    def _readmem(self, p):
        p.expect("$readmemh")
        p.expect("(")
        fname = p.next().strip('"')
        p.expect(",")
        mem = p.next()
        p.expect(")")
        p.expect(";")
        self.readmems = getattr(self, "readmems", []) + [(fname, mem)]

    def _load_hex(self, fname, mem):
        """$readmemh: whitespace separated hex words, @addr sets the cursor
        (the format sw/hexfile.cpp:76-88 writes)."""
        import os
        path = os.path.join(self.readmem_dir or ".", fname)
        addr = 0
        for tok in open(path).read().split():
            if tok.startswith("@"):
                addr = int(tok[1:], 16)
            else:
                self.state[mem][addr] = int(tok, 16)
                addr += 1

    # ---- evaluation
    def reset_state(self):
        self.state = {}
        for n, (w, s, alen) in self.decl.items():
            self.state[n] = [0] * alen if alen else 0
        for st, env in self.initials:
            ups = []
            self.exec(st, env, ups, blocking=True)
        for fname, mem in getattr(self, "readmems", []):
            self._load_hex(fname, mem)

    def sig(self, node, env):
        """self-determined (width, signed) of an expression."""
        k = node.kind
        if k == "num":
            _, w, s = parse_literal(node.args[0])
            return w, s
        if k == "id":
            n = node.args[0]
            if n in env or n in self.params:
                return 32, True
            w, s, _ = self.decl[n]
            return w, s
        if k == "idx":
            base = node.args[0]
            if base.kind == "id" and self.decl.get(base.args[0], (0, 0, None))[2]:
                w, s, _ = self.decl[base.args[0]]
                return w, s
            return 1, False
        if k == "part":
            hi = self.const(node.args[1], env)
            lo = self.const(node.args[2], env)
            return hi - lo + 1, False
        if k == "cat":
            return sum(self.sig(i, env)[0] for i in node.args[0]), False
        if k == "rep":
            return self.const(node.args[0], env) * self.sig(node.args[1], env)[0], False
        if k == "signed":
            return self.sig(node.args[0], env)[0], True
        if k == "un":
            if node.args[0] in ("!", "&", "|"):
                return 1, False
            return self.sig(node.args[1], env)
        op, a, b = node.args
        if op in ("||", "&&", "==", "!=", "<", "<=", ">", ">="):
            return 1, False
        wa, sa = self.sig(a, env)
        if op in (">>>", ">>", "<<", "<<<"):
            return wa, sa
        wb, sb = self.sig(b, env)
        return max(wa, wb), sa and sb

    def raw(self, node, env):
        """unsigned bit pattern of a self-determined operand."""
        w, _ = self.sig(node, env)
        return self.val(node, env, False) & ((1 << w) - 1)

    def val(self, node, env, signed=None, cw=None):
        """Value of `node` as IEEE 1364-2005 5.4-5.5 evaluate it: in a context
        of `cw` bits (None: the expression's self-determined width; callers
        pass max(width of the right-hand side, width of the target)) and of the
        given signedness (None: the expression's own -- unsigned as soon as one
        operand is).  Context-determined operands (of + - * ~ unary-, the left
        operand of a shift) are extended to cw bits first -- sign-extended only
        if the WHOLE expression is signed -- and every such operation wraps to
        cw bits, so a carry out of the context is lost exactly as in a
        simulator.  Returns a signed integer in a signed context, else a
        non-negative one below 2^cw."""
        k = node.kind
        w, s = self.sig(node, env)
        if signed is None:
            signed = s
        if cw is None:
            cw = w

        def wrap(v):
            v &= (1 << cw) - 1
            if signed and (v >> (cw - 1)) & 1:
                v -= 1 << cw
            return v

        def leaf(u):
            # a w-bit operand extended to the context
            if signed and s and (u >> (w - 1)) & 1:
                return u - (1 << w)
            return u
        if k == "num":
            return leaf(parse_literal(node.args[0])[0])
        if k == "id":
            n = node.args[0]
            if n in env:
                return env[n]
            if n in self.params:
                return self.params[n]
            if n in self.assigns:
                rhs, e2 = self.assigns[n]
                c2 = max(w, self.sig(rhs, e2)[0])
                return leaf(self.val(rhs, e2, None, c2) & ((1 << w) - 1))
            if n in self.assign_parts:
                u = 0
                for hi, lo, rhs, e2 in self.assign_parts[n]:
                    pw_ = hi - lo + 1
                    c2 = max(pw_, self.sig(rhs, e2)[0])
                    u |= (self.val(rhs, e2, None, c2) & ((1 << pw_) - 1)) << lo
                return leaf(u)
            return leaf(self.state[n])
        if k == "idx":
            base = node.args[0]
            i = self.val(node.args[1], env)
            if base.kind == "id" and self.decl.get(base.args[0], (0, 0, None))[2]:
                return leaf(self.state[base.args[0]][i])
            return (self.raw(base, env) >> i) & 1
        if k == "part":
            hi = self.const(node.args[1], env)
            lo = self.const(node.args[2], env)
            return (self.raw(node.args[0], env) >> lo) & ((1 << (hi - lo + 1)) - 1)
        if k == "cat":                      # operands self-determined, unsigned
            v = 0
            for it in node.args[0]:
                iw = self.sig(it, env)[0]
                v = (v << iw) | self.raw(it, env)
            return v
        if k == "rep":
            n = self.const(node.args[0], env)
            iw = self.sig(node.args[1], env)[0]
            b = self.raw(node.args[1], env)
            v = 0
            for _ in range(n):
                v = (v << iw) | b
            return v
        if k == "signed":                   # $signed(): operand self-determined
            return leaf(self.raw(node.args[0], env))
        if k == "un":
            op, a = node.args
            if op == "!":
                return int(self.raw(a, env) == 0)
            if op == "&":
                aw = self.sig(a, env)[0]
                return int(self.raw(a, env) == (1 << aw) - 1)
            if op == "|":
                return int(self.raw(a, env) != 0)
            if op == "-":
                return wrap(-self.val(a, env, signed, cw))
            if op == "~":
                return wrap(~self.val(a, env, signed, cw))
            return self.val(a, env, signed, cw)
        op, a, b = node.args
        if op == "||":
            return int(self.raw(a, env) != 0 or self.raw(b, env) != 0)
        if op == "&&":
            return int(self.raw(a, env) != 0 and self.raw(b, env) != 0)
        if op in ("==", "!=", "<", "<=", ">", ">="):
            # operands sized to the wider of the two, signed only if both are
            (wa, sa), (wb, sb) = self.sig(a, env), self.sig(b, env)
            sg, cc = sa and sb, max(wa, wb)
            x, y = self.val(a, env, sg, cc), self.val(b, env, sg, cc)
            return int({"==": x == y, "!=": x != y, "<": x < y, "<=": x <= y,
                        ">": x > y, ">=": x >= y}[op])
        if op in (">>>", ">>", "<<", "<<<"):
            sh = self.raw(b, env)           # the amount is self-determined
            x = self.val(a, env, signed, cw)
            if op == ">>>" and signed and s:
                return x >> sh              # arithmetic: the context is signed
            if op in (">>>", ">>"):
                return (x & ((1 << cw) - 1)) >> sh
            return wrap(x << sh)
        x, y = self.val(a, env, signed, cw), self.val(b, env, signed, cw)
        if op == "*":
            return wrap(x * y)
        return wrap(x + y if op == "+" else x - y)

    def exec(self, st, env, ups, blocking=False):
        k = st.kind
        if k == "block":
            for s in st.args[0]:
                self.exec(s, env, ups, blocking)
        elif k == "if":
            c, a, b = st.args
            if self.raw(c, env):
                self.exec(a, env, ups, blocking)
            elif b is not None:
                self.exec(b, env, ups, blocking)
        elif k == "case":
            sel, arms = st.args
            v = self.raw(sel, env)
            for labels, body in arms:
                if labels is None or any(self.raw(l, env) == v for l in labels):
                    self.exec(body, env, ups, blocking)
                    break
        else:
            lhs, rhs = st.args
            now = blocking or k == "ba"
            if lhs.kind == "cat":          # { a, b } <= value
                widths = [self.decl[i.args[0]][0] for i in lhs.args[0]]
                cw = max(sum(widths), self.sig(rhs, env)[0])
                v = self.val(rhs, env, None, cw) & ((1 << sum(widths)) - 1)
                for item, w in zip(reversed(lhs.args[0]), reversed(widths)):
                    part = v & ((1 << w) - 1)
                    v >>= w
                    if now:
                        self._store(item.args[0], None, part)
                    else:
                        ups.append((item.args[0], None, part))
                return
            if lhs.kind == "idx":
                name = lhs.args[0].args[0]
                idx = self.val(lhs.args[1], env)
                if self.decl[name][2] is None:     # bit select of a vector
                    bit = self.val(rhs, env, None,
                                   max(1, self.sig(rhs, env)[0])) & 1
                    if now:
                        self._store(name, ("bit", idx), bit)
                    else:
                        ups.append((name, ("bit", idx), bit))
                    return
            else:
                name, idx = lhs.args[0], None
            w = self.decl[name][0]
            # the RHS is evaluated in its own signedness, in a context as wide
            # as the wider of the two sides, then truncated to the target
            v = self.val(rhs, env, None, max(w, self.sig(rhs, env)[0])) \
                & ((1 << w) - 1)
            if now:
                self._store(name, idx, v)
            else:
                ups.append((name, idx, v))

    def _store(self, name, idx, v):
        if isinstance(idx, tuple):             # ("bit", k)
            k = idx[1]
            self.state[name] = (self.state[name] & ~(1 << k)) | (v << k)
        elif idx is None:
            self.state[name] = v
        else:
            self.state[name][idx] = v

    def tick(self, **inputs):
        """One rising clock edge with the given input port values."""
        for n, v in inputs.items():
            w = self.decl[n][0]
            self.state[n] = v & ((1 << w) - 1)
        self.settle()
        ups = []
        for st, env in self.always:
            self.exec(st, env, ups)
        for name, idx, v in ups:
            self._store(name, idx, v)
        self.settle()

    def settle(self):
        """always @(*) blocks, in source order (the generator never emits one
        that depends on a later one)."""
        for st, env in self.comb:
            self.exec(st, env, [], blocking=True)

    def get(self, name):
        """current raw value of a reg, or of a wire driven by `assign`."""
        if name in self.assigns or name in self.assign_parts:
            return self.raw(Node("id", name), {})
        return self.state[name]

    def out(self, name):
        w, s, _ = self.decl[name]
        v = self.get(name)
        if s and (v >> (w - 1)) & 1:
            v -= 1 << w
        return v


# ---------------------------------------------------------------- drivers

def has(m, port):
    return port in m.decl


def run_pipelined(m, samples):
    """samples: list of dicts of input ports.  One sample per clock with
    i_ce = 1 (bench/cpp/cordic_tb.cpp:136-176); outputs are collected when the
    aux bit that went in with the sample comes out."""
    assert has(m, "i_aux") and has(m, "o_aux"), "generate the core with -a"
    outs = [n for n in m.decl if n.startswith("o_") and n != "o_aux"]
    ctl = {}
    if has(m, "i_reset"):
        ctl["i_reset"] = 0
    if has(m, "i_areset_n"):
        ctl["i_areset_n"] = 1
    res = []
    zero = {k: 0 for k in samples[0]}
    for k in range(len(samples) + 200):
        s = samples[k] if k < len(samples) else zero
        m.tick(i_ce=1, i_aux=1 if k < len(samples) else 0, **ctl, **s)
        if m.get("o_aux"):
            res.append({n: m.out(n) for n in outs})
        if len(res) == len(samples):
            break
    assert len(res) == len(samples)
    return res


def run_sequential(m, samples, clocks_per_output):
    """bench/cpp/cordic_tb.cpp:146-159: i_stb for one tick, o_done exactly on
    tick CLOCKS_PER_OUTPUT."""
    outs = [n for n in m.decl if n.startswith("o_")
            and n not in ("o_aux", "o_busy", "o_done")]
    ctl = {"i_reset": 0} if has(m, "i_reset") else {}
    res = []
    for s in samples:
        for j in range(clocks_per_output):
            m.tick(i_stb=1 if j == 0 else 0, i_aux=1, **ctl, **s)
            done = m.state["o_done"]
            assert bool(done) == (j == clocks_per_output - 1), (j, done)
        res.append({n: m.out(n) for n in outs})
    return res
