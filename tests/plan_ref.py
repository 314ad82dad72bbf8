"""A plain-Python reading of the plan language (lingo-db_amd/host/ldb_plan.cpp) — TEST INFRASTRUCTURE, like oracle/: it lets the CPU suite run a
step list without a device, so that a plan file and the plan `ldb_subop_translate` makes of the corresponding sub-operator dump can both be held
against the oracle legs (oracle/tpch_legs.py) on a small generated database.  Rows are Python lists, decimals unscaled Python integers with their
(precision, scale), typed by the HOST LIBRARY's own rules (`ldb_host_decimal_type`: typeAfterMul / typeAfterDiv / getHigherDecimalType / the AVG
type, lingo-db_amd/host/ldb_host.cpp after sql_analyzer.cpp:3058-3159) — only the evaluation is restated here, not the typing.  Nothing under
lingo-db_amd/ imports this."""
import ctypes as C
import datetime
import re

import pyarrow as pa

from lingodb_amd import capi

EPOCH = datetime.date(1970, 1, 1)
INT, DATE, STR, BOOL, CH = ("int",), ("date",), ("str",), ("bool",), ("ch",)


def dec(p, s): return ("dec", p, s)


def _rule(op, a, b=(0, 0)):
    p, s = C.c_int32(), C.c_int32()
    capi.host_lib().ldb_host_decimal_type(op, a[0], a[1], b[0], b[1], C.byref(p), C.byref(s))
    return p.value, s.value


def tdiv(a, b):  # truncating division (sdiv)
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


class Rel:
    """row-aligned sides; a side is {name: (type, values)}"""
    def __init__(self, sides, n):
        self.sides, self.n = sides, n

    def col(self, ref):
        if ":" in ref and ref.split(":", 1)[0].isdigit():
            k, name = ref.split(":", 1)
            return self.sides[int(k)][name]
        for s in self.sides:
            if ref in s:
                return s[ref]
        raise KeyError("no column '%s'" % ref)

    def take(self, idx, null_ok=False):
        out = []
        for s in self.sides:
            out.append({n: (t, [None if i is None else v[i] for i in idx]) for n, (t, v) in s.items()})
        return Rel(out, len(idx))


def table_from_arrow(t):
    side = {}
    for name, col in zip(t.schema.names, t.columns):
        ty, arr = col.type, col.combine_chunks()
        if pa.types.is_decimal(ty):
            side[name] = (dec(ty.precision, ty.scale), [None if v.as_py() is None else int(v.as_py().scaleb(ty.scale)) for v in arr])
        elif pa.types.is_date32(ty):
            side[name] = (DATE, [None if v.as_py() is None else (v.as_py() - EPOCH).days for v in arr])
        elif pa.types.is_fixed_size_binary(ty):
            side[name] = (CH, [None if v.as_py() is None else int.from_bytes(v.as_py(), "little", signed=True) for v in arr])
        elif pa.types.is_string(ty):
            side[name] = (STR, arr.to_pylist())
        else:
            side[name] = (INT, arr.to_pylist())
    return Rel([side], t.num_rows)


def _like(pattern):
    rx = "".join(".*" if c == "%" else "." if c == "_" else re.escape(c) for c in pattern)
    return re.compile("^" + rx + "$", re.S)


def _const(ty, v):
    """a predicate constant in the column's domain (Restrictions::create: dates parsed, decimals rescaled to the column, char(1) as its 4 bytes)"""
    if ty == DATE:
        return (datetime.date.fromisoformat(v) - EPOCH).days if isinstance(v, str) else v
    if ty[0] == "dec":
        txt = str(v)
        neg = txt.startswith("-")
        whole, _, frac = txt.lstrip("-").partition(".")
        frac = (frac + "0" * ty[2])[: ty[2]]
        x = int(whole or "0") * 10 ** ty[2] + int(frac or "0")
        return -x if neg else x
    if ty == CH:
        return int.from_bytes(str(v).encode().ljust(4, b"\0")[:4], "little", signed=True) if isinstance(v, str) else v
    return v


CMP = {"EQ": lambda a, b: a == b, "NEQ": lambda a, b: a != b, "LT": lambda a, b: a < b, "LTE": lambda a, b: a <= b, "GT": lambda a, b: a > b, "GTE": lambda a, b: a >= b}


class Interp:
    def __init__(self, tables):
        self.env = dict(tables)  # name → Rel | ("ht", Rel, keys)

    # ------------------------------------------------------------ predicates
    def pred_mask(self, rel, p):
        ty, v = rel.col(p["col"])
        op = p["op"]
        if op == "NOTNULL":
            return [x is not None for x in v]
        if op in ("LIKE", "NOT LIKE"):
            rx, want = _like(p["value"]), op == "LIKE"
            return [x is not None and (rx.match(x) is not None) == want for x in v]
        if op == "IN":
            vals = {_const(ty, c) for c in p["values"]}
            return [x in vals for x in v]
        f = CMP[op]
        if "rhs_col" in p:
            _, w = rel.col(p["rhs_col"])
            return [a is not None and b is not None and f(a, b) for a, b in zip(v, w)]
        if "scalar" in p:  # column x 10^k OP the unscaled scalar (ldb_plan.cpp: the comparison after the cast to the common scale)
            sc = p["scalar"]
            src = self.env[sc["from"]]
            _, sv = src.col(sc["col"])
            if src.n < 1 or sv[0] is None:
                return [False] * rel.n
            m = 10 ** sc.get("div_pow10", 0)
            return [a is not None and f(a * m, sv[0]) for a in v]
        c = _const(ty, p["value"])
        return [a is not None and f(a, c) for a in v]

    def conj_mask(self, rel, preds):
        m = [True] * rel.n
        for p in preds:
            pm = self.pred_mask(rel, p)
            m = [a and b for a, b in zip(m, pm)]
        return m

    # ------------------------------------------------------------ expressions (compileX)
    def expr(self, rel, e):
        """→ (type, values)"""
        if isinstance(e, bool):
            raise ValueError("boolean literal")
        if isinstance(e, int):
            return INT, [e] * rel.n
        if isinstance(e, str):
            if re.fullmatch(r"-?\d*\.\d+", e) or re.fullmatch(r"-?\d+\.\d*", e):
                s = len(e) - e.index(".") - 1
                p = len(e) - 1 - (1 if e.startswith("-") else 0)
                return dec(p, s), [_const(dec(p, s), e)] * rel.n
            return rel.col(e)
        (op, args), = e.items()
        as_dec = lambda t: (t[1], t[2]) if t[0] == "dec" else (19, 0)
        scale = lambda t: t[2] if t[0] == "dec" else 0

        def cast(t, v, to):
            k = to[1] - scale(t)
            return v if k == 0 else [None if x is None else x * 10 ** k for x in v]
        if op in ("add", "sub"):
            (lt, lv), (rt, rv) = self.expr(rel, args[0]), self.expr(rel, args[1])
            sgn = 1 if op == "add" else -1
            if lt[0] != "dec" and rt[0] != "dec":
                return INT, [None if a is None or b is None else a + sgn * b for a, b in zip(lv, rv)]
            t = _rule(2, as_dec(lt), as_dec(rt))
            lv, rv = cast(lt, lv, t), cast(rt, rv, t)
            return dec(*t), [None if a is None or b is None else a + sgn * b for a, b in zip(lv, rv)]
        if op == "mul":
            (lt, lv), (rt, rv) = self.expr(rel, args[0]), self.expr(rel, args[1])
            for extra in args[2:]:  # n-ary products fold left to right
                lt, lv = self._mul(lt, lv, rt, rv)
                rt, rv = self.expr(rel, extra)
            return self._mul(lt, lv, rt, rv)
        if op == "div":
            (lt, lv), (rt, rv) = self.expr(rel, args[0]), self.expr(rel, args[1])
            dl, dr = as_dec(lt), as_dec(rt)
            t = _rule(1, dl, dr)
            k = t[1] + dr[1] - dl[1]
            return dec(*t), [None if a is None or b is None or b == 0 else tdiv(a * 10 ** k, b) for a, b in zip(lv, rv)]
        if op == "cmp":
            (lt, lv), (rt, rv) = self.expr(rel, args[1]), self.expr(rel, args[2])
            if lt[0] == "dec" or rt[0] == "dec":
                t = _rule(2, as_dec(lt), as_dec(rt))
                lv, rv = cast(lt, lv, t), cast(rt, rv, t)
            f = CMP[args[0]]
            return BOOL, [None if a is None or b is None else int(f(a, b)) for a, b in zip(lv, rv)]
        if op in ("and", "or"):
            (_, lv), (_, rv) = self.expr(rel, args[0]), self.expr(rel, args[1])
            g = (lambda a, b: int(bool(a) and bool(b))) if op == "and" else (lambda a, b: int(bool(a) or bool(b)))
            return BOOL, [g(a, b) for a, b in zip(lv, rv)]
        if op == "not":
            _, v = self.expr(rel, args[0])
            return BOOL, [None if a is None else int(not a) for a in v]
        if op == "isnull":
            _, v = self.expr(rel, args[0])
            return BOOL, [int(a is None) for a in v]
        if op in ("coalesce", "case"):
            off = 1 if op == "case" else 0
            (lt, lv), (rt, rv) = self.expr(rel, args[off]), self.expr(rel, args[off + 1])
            out = lt
            if lt[0] == "dec" or rt[0] == "dec":
                t = _rule(2, as_dec(lt), as_dec(rt))
                lv, rv, out = cast(lt, lv, t), cast(rt, rv, t), dec(*t)
            if op == "coalesce":
                return out, [b if a is None else a for a, b in zip(lv, rv)]
            _, cv = self.expr(rel, args[0])
            return out, [a if c else b for c, a, b in zip(cv, lv, rv)]
        if op == "neg":
            t, v = self.expr(rel, args[0])
            return t, [None if a is None else -a for a in v]
        if op == "idiv":  # arith.divsi over two integers
            (_, lv), (_, rv) = self.expr(rel, args[0]), self.expr(rel, args[1])
            return INT, [None if a is None or b is None or b == 0 else tdiv(a, b) for a, b in zip(lv, rv)]
        if op == "row_number":
            return INT, list(range(rel.n))
        raise ValueError("expression operator '%s'" % op)

    @staticmethod
    def _mul(lt, lv, rt, rv):
        if lt[0] != "dec" and rt[0] != "dec":
            return INT, [None if a is None or b is None else a * b for a, b in zip(lv, rv)]
        dl = (lt[1], lt[2]) if lt[0] == "dec" else (19, 0)
        dr = (rt[1], rt[2]) if rt[0] == "dec" else (19, 0)
        t = _rule(0, dl, dr)
        cut = 10 ** max(0, dl[1] + dr[1] - t[1])  # a clamped scale: truncating divide (DecimalMulOpLowering)
        return dec(*t), [None if a is None or b is None else tdiv(a * b, cut) for a, b in zip(lv, rv)]

    def agg_expr(self, rel, e):
        """the aggregate normal form: in (integer ± decimal column) the literal takes the column's scale (`1 - l_discount`)"""
        if isinstance(e, dict):
            (op, args), = e.items()
            if op in ("add", "sub") and isinstance(args[0], int) and isinstance(args[1], str) and not re.fullmatch(r"-?\d*\.\d+", args[1]):
                ty, v = rel.col(args[1])
                if ty[0] == "dec":
                    t = _rule(2, (19, 0), (ty[1], ty[2]))
                    k, sgn = args[0] * 10 ** ty[2], (1 if op == "add" else -1)
                    return dec(*t), [None if a is None else k + sgn * a for a in v]
            if op == "mul":
                t, v = self.agg_expr(rel, args[0])
                for a in args[1:]:
                    rt, rv = self.agg_expr(rel, a)
                    t, v = self._mul(t, v, rt, rv)
                return t, v
            if op in ("add", "sub"):
                (lt, lv), (rt, rv) = self.agg_expr(rel, args[0]), self.agg_expr(rel, args[1])
                if lt[0] == "dec" or rt[0] == "dec":
                    dl = (lt[1], lt[2]) if lt[0] == "dec" else (19, 0)
                    dr = (rt[1], rt[2]) if rt[0] == "dec" else (19, 0)
                    t = _rule(2, dl, dr)
                    lv = [None if a is None else a * 10 ** (t[1] - dl[1]) for a in lv]
                    rv = [None if a is None else a * 10 ** (t[1] - dr[1]) for a in rv]
                    sgn = 1 if op == "add" else -1
                    return dec(*t), [None if a is None or b is None else a + sgn * b for a, b in zip(lv, rv)]
        return self.expr(rel, e)

    # ------------------------------------------------------------ steps
    def rel(self, name):
        v = self.env[name]
        if not isinstance(v, Rel):
            raise TypeError("'%s' is not a relation" % name)
        return v

    def run(self, plan):
        for st in plan["steps"]:
            getattr(self, "op_" + st["op"])(st)
        return self.env[plan.get("result", "result")]

    def op_filter(self, st):
        r = self.rel(st["in"])
        m = self.conj_mask(r, st["preds"])
        self.env[st["out"]] = r.take([i for i in range(r.n) if m[i]])

    def op_filter_dnf(self, st):
        r = self.rel(st["in"])
        m = [False] * r.n
        for cl in st["clauses"]:
            cm = self.conj_mask(r, cl)
            m = [a or b for a, b in zip(m, cm)]
        self.env[st["out"]] = r.take([i for i in range(r.n) if m[i]])

    def op_join_build(self, st):
        self.env[st["out"]] = ("ht", self.rel(st["in"]), list(st["keys"]))

    def op_join_probe(self, st):
        _, b, bkeys = self.env[st["ht"]]
        p = self.rel(st["in"])
        kind = st.get("kind", "inner")
        bk = list(zip(*[b.col(k)[1] for k in bkeys])) if b.n else []
        pk = list(zip(*[p.col(k)[1] for k in st["keys"]])) if p.n else []
        index = {}
        for j, k in enumerate(bk):
            if None not in k:
                index.setdefault(k, []).append(j)
        resid = [(p.col(r["probe"])[1], CMP[r["op"]], b.col(r["build"])[1]) for r in st.get("residual", [])]
        ok = lambda i, j: all(pv[i] is not None and bv[j] is not None and f(pv[i], bv[j]) for pv, f, bv in resid)
        pairs, hit_b = [], set()
        matched = []
        for i, k in enumerate(pk):
            ms = [j for j in index.get(k, []) if ok(i, j)] if None not in k else []
            matched.append(bool(ms))
            for j in ms:
                pairs.append((i, j))
                hit_b.add(j)
        if kind == "semi":
            out = p.take([i for i in range(p.n) if matched[i]])
        elif kind == "anti":
            out = p.take([i for i in range(p.n) if not matched[i]])
        elif kind == "semi_anti_build":  # a partner among the probe rows, none among those that also pass anti_preds
            am = self.conj_mask(p, st["anti_preds"])
            hit_anti = {j for i, j in pairs if am[i]}
            out = b.take([j for j in sorted(hit_b) if j not in hit_anti])
        elif kind == "semi_build":
            out = b.take(sorted(hit_b))
        elif kind == "anti_build":
            out = b.take([j for j in range(b.n) if j not in hit_b])
        else:
            if kind in ("left_outer", "single", "full_outer"):
                seen = {i for i, _ in pairs}
                pairs = sorted(pairs + [(i, None) for i in range(p.n) if i not in seen], key=lambda t: t[0])
            if kind in ("right_outer", "full_outer"):
                pairs = pairs + [(None, j) for j in range(b.n) if j not in hit_b]
            if kind == "mark":
                out = Rel(p.sides + [{st["mark_as"]: (BOOL, [int(x) for x in matched])}], p.n)
            else:
                left, right = p.take([i for i, _ in pairs]), b.take([j for _, j in pairs])
                out = Rel(left.sides + right.sides, len(pairs))
        self.env[st["out"]] = out

    def op_join_nl(self, st):
        p, b = self.rel(st["in"]), self.rel(st["build"])
        resid = [(p.col(r["probe"])[1], CMP[r["op"]], b.col(r["build"])[1]) for r in st.get("residual", [])]
        pairs = [(i, j) for i in range(p.n) for j in range(b.n) if all(f(pv[i], bv[j]) for pv, f, bv in resid)]
        left, right = p.take([i for i, _ in pairs]), b.take([j for _, j in pairs])
        self.env[st["out"]] = Rel(left.sides + right.sides, len(pairs))

    def op_map(self, st):
        r = self.rel(st["in"])
        if st.get("fn") == "extract_year":
            _, v = r.col(st["col"])
            col = (INT, [None if d is None else (EPOCH + datetime.timedelta(days=d)).year for d in v])
        elif st.get("fn") == "substr":
            _, v = r.col(st["col"])
            a, n = st.get("from", 1) - 1, st.get("for", 1 << 30)
            col = (STR, [None if x is None else x[a:a + n] for x in v])
        else:
            col = self.expr(r, st["expr"])
        self.env[st["out"]] = Rel(r.sides + [{st["as"]: col}], r.n)

    def op_groupby(self, st):
        r = self.rel(st["in"])
        if st.get("preds"):
            m = self.conj_mask(r, st["preds"])
            r = r.take([i for i in range(r.n) if m[i]])
        keys = [r.col(k) for k in st.get("keys", [])]
        groups, order = {}, []
        if not keys:
            groups[()] = list(range(r.n))
            order.append(())
        else:
            for i in range(r.n):
                k = tuple(c[1][i] for c in keys)
                if k not in groups:
                    groups[k] = []
                    order.append(k)
                groups[k].append(i)
        side = {}
        names = st.get("key_names", [])
        for n, (kname, (ty, _)) in enumerate(zip(st.get("keys", []), keys)):
            name = names[n] if n < len(names) else kname.split(":", 1)[-1]
            side[name] = (ty, [k[n] for k in order])
        for n, a in enumerate(st["aggs"]):
            fn = a["fn"]
            when = self.conj_mask(r, a["when"]) if a.get("when") else [True] * r.n
            if fn == "count_star":
                side[a.get("as") or "agg%d" % n] = (INT, [sum(1 for i in groups[k] if when[i]) for k in order])
                continue
            ty, v = self.agg_expr(r, a["expr"])
            out, oty = [], ty
            for k in order:
                rows = groups[k]
                vals = [v[i] for i in rows if when[i] and v[i] is not None]
                if fn == "count":
                    out.append(len(vals))
                    oty = INT
                elif fn == "sum":
                    # a conditional SUM (case … else 0) is a non-NULL 0 as soon as one row of the group fails its predicates
                    zero = a.get("when") and any(not when[i] for i in rows)
                    out.append(sum(vals) if vals else (0 if zero else None))
                elif fn == "min":
                    out.append(min(vals) if vals else None)
                elif fn == "max":
                    out.append(max(vals) if vals else None)
                elif fn == "any":
                    out.append(v[rows[0]] if rows else None)
                elif fn == "avg":
                    if ty[0] != "dec":
                        raise ValueError("avg over integers")
                    t = _rule(3, (ty[1], ty[2]))
                    oty = dec(*t)
                    if "count" in a:
                        _, cv = self.agg_expr(r, a["count"])
                        cnt = sum(cv[i] for i in rows if cv[i] is not None)
                    else:
                        cnt = len(vals)
                    out.append(tdiv(sum(vals) * 10 ** (t[1] - ty[2]), cnt) if vals and cnt else None)
                else:
                    raise ValueError(fn)
            side[a.get("as") or "agg%d" % n] = (oty, out)
        self.env[st["out"]] = Rel([side], len(order))

    def _sorted(self, r, by):
        idx = list(range(r.n))
        for b in reversed(by):  # stable, last key first
            name, desc = (b, False) if isinstance(b, str) else (b["col"], b.get("desc", False))
            _, v = r.col(name)
            idx.sort(key=lambda i: v[i], reverse=desc)
        return idx

    def op_sort(self, st):
        r = self.rel(st["in"])
        self.env[st["out"]] = r.take(self._sorted(r, st["by"]))

    def op_topk(self, st):
        r = self.rel(st["in"])
        self.env[st["out"]] = r.take(self._sorted(r, st["by"])[: st["k"]])

    def op_materialize(self, st):
        r = self.rel(st["in"])
        side = {}
        for c in st["cols"]:
            name, alias = (c, c) if isinstance(c, str) else (c["col"], c.get("as", c["col"]))
            side[alias.split(":", 1)[-1] if isinstance(c, str) else alias] = r.col(name)
        self.env[st["out"]] = Rel([side], r.n)


    def op_nested_map(self, st):
        """NestedMapLowering de-correlated as ldb_plan.cpp does it: (outer tuple x scanned state) pairs passing the residual, the nested maps over the
        pairs, and — with `reduce` — one group per OUTER tuple (its row number) carrying the kept outer columns"""
        outer, inner = self.rel(st["in"]), self.rel(st["scan"])
        resid = [(outer.col(r["probe"])[1], CMP[r["op"]], inner.col(r["build"])[1]) for r in st.get("residual", [])]
        pairs = [(i, j) for i in range(outer.n) for j in range(inner.n) if all(f(pv[i], bv[j]) for pv, f, bv in resid)]
        left, right = outer.take([i for i, _ in pairs]), inner.take([j for _, j in pairs])
        cur = Rel(left.sides + right.sides + [{"__tuple": (INT, [i for i, _ in pairs])}], len(pairs))
        for m in st.get("map", []):
            cur = Rel(cur.sides + [{m["as"]: self.expr(cur, m["expr"])}], cur.n)
        red = st.get("reduce")
        if red is None:
            self.env[st["out"]] = cur
            return
        self.env["__nm"] = cur
        self.op_groupby({"in": "__nm", "keys": ["__tuple"], "aggs": list(red["aggs"]) + [{"fn": "any", "expr": c, "as": c} for c in red.get("keep", [])], "out": st["out"]})

    def op_loop(self, st):
        """LoopLowering: the body once per iteration over the loop-carried tables; the results are the states handed to the LAST loop_continue"""
        state = {v["name"]: self.env[v["init"]] for v in st["vars"]}
        for _ in range(st.get("max_iterations", 1000)):
            body = Interp({**self.env, **state})
            for b in st["body"]:
                getattr(body, "op_" + b["op"])(b)
            state = {**state, **{n["var"]: body.env[n["from"]] for n in st["next"]}}
            cond = body.rel(st["continue"]["from"]).col(st["continue"]["col"])[1]
            if not (cond and cond[0]):
                break
        else:
            raise RuntimeError("no fixpoint after %d iterations" % st.get("max_iterations", 1000))
        for r in st["results"]:
            self.env[r["name"]] = state[r["var"]]

    def op_set_op(self, st):
        """UNION [ALL] / INTERSECT [ALL] / EXCEPT [ALL] over two column lists (NULLs compare equal); the result carries the left names"""
        import collections

        l, r = self.rel(st["left"]), self.rel(st["right"])
        lc, rc = [l.col(c) for c in st["left_cols"]], [r.col(c) for c in st["right_cols"]]
        lrows = list(zip(*[v for _, v in lc])) if l.n else []
        rrows = list(zip(*[v for _, v in rc])) if r.n else []
        cl, cr = collections.Counter(lrows), collections.Counter(rrows)
        kind = st["kind"]
        if kind == "union_all":
            out = lrows + rrows
        elif kind == "union":
            out = list(dict.fromkeys(lrows + rrows))
        elif kind == "intersect":
            out = [k for k in dict.fromkeys(lrows) if k in cr]
        elif kind == "except":
            out = [k for k in dict.fromkeys(lrows) if k not in cr]
        elif kind == "intersect_all":
            out = list((cl & cr).elements())
        elif kind == "except_all":
            out = list((cl - cr).elements())
        else:
            raise ValueError(kind)
        names = st.get("as") or [c.split(":", 1)[-1] for c in st["left_cols"]]
        self.env[st["out"]] = Rel([{n: (lc[i][0], [row[i] for row in out]) for i, n in enumerate(names)}], len(out))

    def op_window(self, st):
        """rows ordered by (partition, order); per row the frame [from, to] as offsets clamped into its partition; rank = rows from the frame begin to the
        current row, the aggregates over the frame (ldb_gpu_window after WindowLowering)"""
        r = self.rel(st["in"])
        part = [r.col(c)[1] for c in st.get("partition_by", [])]
        idx = list(range(r.n))
        for b in reversed(st.get("order_by", [])):
            name, desc = (b, False) if isinstance(b, str) else (b["col"], b.get("desc", False))
            v = r.col(name)[1]
            idx.sort(key=lambda i: v[i], reverse=desc)
        idx.sort(key=lambda i: tuple(p[i] for p in part))
        out = r.take(idx)
        keys = [tuple(p[i] for p in part) for i in idx]
        end = lambda v: -(1 << 62) if v == "unbounded_preceding" else (1 << 62) if v == "unbounded_following" else (0 if v == "current_row" else v)
        frm, to = end(st.get("frame_from", "unbounded_preceding")), end(st.get("frame_to", 0))
        cols = {}
        for f in st["fns"]:
            arg = out.col(f["col"]) if "col" in f else None
            vals = []
            lo = 0
            for pos in range(out.n):
                if pos == 0 or keys[pos] != keys[pos - 1]:
                    lo = pos
                    hi = pos
                    while hi + 1 < out.n and keys[hi + 1] == keys[pos]:
                        hi += 1
                a, b = max(lo, min(hi, pos + frm)), max(lo, min(hi, pos + to))
                if f["fn"] == "rank":
                    vals.append(pos - a + 1)
                    continue
                frame = [x for x in (arg[1][a:b + 1] if arg else [1] * (b - a + 1)) if x is not None]
                vals.append({"sum": lambda: sum(frame) if frame else None, "min": lambda: min(frame) if frame else None, "max": lambda: max(frame) if frame else None,
                             "count": lambda: len(frame), "count_star": lambda: b - a + 1}[f["fn"]]())
            cols[f["as"]] = (arg[0] if arg and f["fn"] in ("sum", "min", "max") else INT, vals)
        self.env[st["out"]] = Rel(out.sides + [cols], out.n)


def rows(rel):
    cols = [v for s in rel.sides for (_, v) in s.values()]
    return list(zip(*cols)) if cols else []


def run_sharded(plan, rank_tables):
    """the SAME step list on every rank over its shard, in lockstep (ldb_plan_run_json_comm): `allgather` hands every rank all ranks' rows in rank
    order, `shuffle` sends each row to the rank its key hashes to (any function of the key will do for correctness: equal keys meet on one rank).
    rank_tables: one {input name: Rel} per rank → one result Rel per rank"""
    world = len(rank_tables)
    ranks = [Interp(t) for t in rank_tables]

    def concat(rels):
        sides = []
        for k in range(len(rels[0].sides)):
            sides.append({n: (t, [x for r in rels for x in r.sides[k][n][1]]) for n, (t, _) in rels[0].sides[k].items()})
        return Rel(sides, sum(r.n for r in rels))
    for st in plan["steps"]:
        if st["op"] == "allgather":
            mine = []
            for it in ranks:  # a table travels whole: every column of its single side
                r = it.rel(st["in"])
                mine.append(Rel([{n: c for s in r.sides for n, c in s.items()}], r.n))
            everybody = concat(mine)
            for it in ranks:
                it.env[st["out"]] = everybody
        elif st["op"] == "shuffle":
            parts = [[] for _ in range(world)]
            for it in ranks:
                r = it.rel(st["in"])
                cols = {c: r.col(c) for c in st["cols"]}
                keys = [r.col(k)[1] for k in st["keys"]]
                dest = [hash(tuple(k[i] for k in keys)) % world for i in range(r.n)]
                for d in range(world):
                    idx = [i for i in range(r.n) if dest[i] == d]
                    parts[d].append(Rel([{c: (t, [v[i] for i in idx]) for c, (t, v) in cols.items()}], len(idx)))
            for d, it in enumerate(ranks):
                it.env[st["out"]] = concat(parts[d])
        else:
            for it in ranks:
                getattr(it, "op_" + st["op"])(st)
    return [it.env[plan.get("result", "result")] for it in ranks]
