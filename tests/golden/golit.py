"""A small reader for the Go the reference's table tests are written in — enough of the
expression and composite-literal syntax to lift test tables out of *_test.go files as data,
so that no fixture value has to be typed by hand (tests/golden/make_golden.py).

Only syntax is handled, nothing is executed: integer / bool / string literals, identifiers
(looked up in an environment, else kept as their name), + - * /, conversions such as uint64(x),
calls (kept as {"call": name, "args": [...]}), composite literals with or without a type
(keyed -> dict, positional -> list), slices, maps, nil.
"""
import re

TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F]+|\d+)
  | (?P<str>"(?:[^"\\]|\\.)*"|`[^`]*`)
  | (?P<id>[A-Za-z_][A-Za-z0-9_]*(?:\.[A-Za-z_][A-Za-z0-9_]*)*)
  | (?P<op>:=|\.\.\.|[{}()\[\],:&+\-*/=<>!;])
""", re.X | re.S)

CONVERSIONS = {"uint64", "int", "int64", "uint32", "uint8", "StateType", "Index"}


class GoSyntaxError(ValueError):
    pass


def tokenize(text):
    out, pos = [], 0
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise GoSyntaxError("cannot tokenize at %r" % text[pos:pos + 30])
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group(m.lastgroup)))
    return out


class Parser:
    def __init__(self, tokens, env=None):
        self.t, self.i, self.env = tokens, 0, dict(env or {})

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def take(self, val=None):
        kind, v = self.peek()
        if val is not None and v != val:
            raise GoSyntaxError("expected %r, found %r (token %d)" % (val, v, self.i))
        self.i += 1
        return v

    # -- types (skipped, only their extent matters) -------------------------------------
    def skip_type(self):
        """[]T, map[K]V, *T, T — returns a short tag ('slice', 'map', 'name')."""
        kind, v = self.peek()
        if v == "[":
            self.take("[")
            self.take("]")
            self.skip_type()
            return "slice"
        if v == "*":
            self.take()
            return self.skip_type()
        if v == "map":
            self.take()
            self.take("[")
            self.skip_type()
            self.take("]")
            self.skip_type()
            return "map"
        if kind == "id":
            self.take()
            return "name"
        raise GoSyntaxError("type expected, found %r" % v)

    # -- expressions ---------------------------------------------------------------------
    def expr(self):
        v = self.term()
        while self.peek()[1] in ("+", "-"):
            op = self.take()
            r = self.term()
            v = v + r if op == "+" else v - r
        return v

    def term(self):
        v = self.unary()
        while self.peek()[1] in ("*", "/"):
            if self.take() == "*":
                v = v * self.unary()
            else:
                v = v // self.unary()  # Go's integer division (operands here are unsigned)
        return v

    def unary(self):
        kind, v = self.peek()
        if v == "&":
            self.take()
            return self.unary()
        if v == "-":
            self.take()
            return -self.unary()
        if v == "{":  # untyped composite (element of a slice / map of structs)
            return self.composite()
        if v == "[" or v == "map":
            self.skip_type()
            return self.composite()
        if v == "(":
            self.take()
            x = self.expr()
            self.take(")")
            return x
        if kind == "num":
            self.take()
            return int(v, 0)
        if kind == "str":
            self.take()
            return v[1:-1]
        if kind == "id":
            self.take()
            nxt = self.peek()[1]
            if nxt == "(":
                args = self.args()
                if v in CONVERSIONS and len(args) == 1:
                    return args[0]
                return {"call": v, "args": args}
            if nxt == "{" and v[:1].isupper() or (nxt == "{" and "." in v):  # Type{...}
                return self.composite()
            if v == "true":
                return True
            if v == "false":
                return False
            if v == "nil":
                return None
            return self.env.get(v, v)
        raise GoSyntaxError("expression expected, found %r (token %d)" % (v, self.i))

    def args(self):
        self.take("(")
        out = []
        while self.peek()[1] != ")":
            if self.peek()[1] == "[" or self.peek()[1] == "map":  # a type argument, e.g. make([]inflight, 10)
                save = self.i
                self.skip_type()
                if self.peek()[1] == "{":
                    self.i = save
                    out.append(self.expr())
                else:
                    out.append("<type>")
            else:
                out.append(self.expr())
            if self.peek()[1] == ",":
                self.take()
        self.take(")")
        return out

    def composite(self):
        self.take("{")
        keyed, items = None, []
        while self.peek()[1] != "}":
            # key: value ?  (key is an identifier or a constant expression)
            save = self.i
            is_key = False
            if self.peek(1)[1] == ":" and self.peek()[0] in ("id", "num", "str"):
                is_key = True
            if is_key:
                kkind, k = self.peek()
                self.take()
                self.take(":")
                key = int(k, 0) if kkind == "num" else (k[1:-1] if kkind == "str" else self.env.get(k, k))
                items.append((key, self.expr()))
                keyed = True if keyed is None else keyed
            else:
                self.i = save
                items.append(self.expr())
                keyed = False if keyed is None else keyed
            if self.peek()[1] == ",":
                self.take()
        self.take("}")
        if keyed:
            return dict(items)
        return items


def parse_expr(text, env=None):
    p = Parser(tokenize(text), env)
    v = p.expr()
    if p.peek()[0] != "eof":
        raise GoSyntaxError("trailing tokens after expression: %r" % (p.peek(),))
    return v


# -- source navigation ------------------------------------------------------------------

def match_brace(text, i):
    """Index of the `}` that closes the `{` at text[i] (strings and comments skipped)."""
    assert text[i] == "{"
    depth, n = 0, len(text)
    while i < n:
        c = text[i]
        if c == "/" and text[i:i + 2] == "//":
            i = text.index("\n", i)
            continue
        if c == '"':
            i += 1
            while text[i] != '"':
                i += 2 if text[i] == "\\" else 1
        elif c == "`":
            i = text.index("`", i + 1)
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise GoSyntaxError("unbalanced braces")


def func_source(text, name):
    """(body text, 1-based line of `func name(`, 1-based last line) of a top-level Go function."""
    m = re.search(r"^func %s\(" % re.escape(name), text, re.M)
    if not m:
        raise GoSyntaxError("func %s not found" % name)
    start = m.start()
    depth, i, n = 0, text.index("{", m.end()), len(text)
    body_start = i
    while i < n:
        c = text[i]
        if c == "/" and text[i:i + 2] == "//":
            i = text.index("\n", i)
            continue
        if c == '"':
            i += 1
            while text[i] != '"':
                i += 2 if text[i] == "\\" else 1
        elif c == "`":
            i = text.index("`", i + 1)
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    first = text.count("\n", 0, start) + 1
    last = text.count("\n", 0, i) + 1
    return text[body_start + 1:i], first, last


def struct_table(body, env=None, anchor=r"\[\]struct\s*\{"):
    """The first `[]struct { fields }{ rows }` of a function body -> list of dicts keyed by field name."""
    m = re.search(anchor, body)
    if not m:
        raise GoSyntaxError("no []struct table")
    f0 = m.end() - 1                      # the `{` of the field list
    f1 = match_brace(body, f0)
    r0 = body.index("{", f1)              # the `{` of the rows
    r1 = match_brace(body, r0)
    toks = tokenize(body[f0 + 1:r1 + 1])
    p = Parser(toks, env)
    fields = []
    while p.peek()[1] != "}":
        names = [p.take()]
        while p.peek()[1] == ",":
            p.take()
            names.append(p.take())
        p.skip_type()
        fields.extend(names)
    p.take("}")
    rows = p.composite()
    out = []
    for r in rows:
        if isinstance(r, dict):
            unknown = set(r) - set(fields)
            if unknown:
                raise GoSyntaxError("unknown fields %s" % unknown)
            out.append({f: r.get(f) for f in fields})  # absent = Go zero value: None here, the caller decides
        else:
            if len(r) != len(fields):
                raise GoSyntaxError("row of %d values for %d fields" % (len(r), len(fields)))
            out.append(dict(zip(fields, r)))
    return out


def statements(body):
    """Top-level statements of a block as (text, line offset) — split on newlines at brace/paren depth 0."""
    out, depth, cur, line0, line = [], 0, [], 0, 0
    for raw in body.split("\n"):
        code = re.sub(r"//.*$", "", raw)
        if not cur:
            line0 = line
        if code.strip():
            cur.append(code.strip())
        depth += code.count("{") + code.count("(") - code.count("}") - code.count(")")
        if depth == 0 and cur:
            out.append((" ".join(cur), line0))
            cur = []
        line += 1
    if cur:
        raise GoSyntaxError("unbalanced block")
    return out
