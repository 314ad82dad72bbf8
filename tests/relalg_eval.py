"""Direct evaluation of the small relational-algebra trees of tools/subop_lower.py over Python rows — TEST INFRASTRUCTURE for tests/test_translator_fuzz.py:
an answer that does not pass through the lowering, the dump, the translator or the plan language.  Rows are dicts {column display name: value}; decimals
are exact Fractions here (the plan side carries unscaled integers; results are compared after scaling), dates are days, char(1) its character."""
import datetime
import fractions

import pyarrow as pa

import subop_lower as L

EPOCH = datetime.date(1970, 1, 1)


def table_rows(arrow, alias):
    cols = {}
    for name, col in zip(arrow.schema.names, arrow.columns):
        ty = col.type
        if pa.types.is_decimal(ty):
            cols[name] = [None if v is None else fractions.Fraction(v) for v in col.to_pylist()]
        elif pa.types.is_date32(ty):
            cols[name] = [None if v is None else (v - EPOCH).days for v in col.to_pylist()]
        elif pa.types.is_fixed_size_binary(ty):
            cols[name] = [None if v is None else v.rstrip(b"\0").decode() for v in col.to_pylist()]
        else:
            cols[name] = col.to_pylist()
    return [{"%s::%s" % (alias, n): cols[n][i] for n in cols} for i in range(arrow.num_rows)]


def _constant(e, like):
    """a dump constant in the domain of the value it is compared with"""
    v, ty = e["value"], e.get("data_type", "")
    if ty == "date" or (isinstance(v, str) and isinstance(like, int) and len(v) == 10 and v[4] == "-"):
        return (datetime.date.fromisoformat(v) - EPOCH).days
    if ty.startswith("decimal") or isinstance(like, fractions.Fraction):
        return fractions.Fraction(str(v))
    return v


CMP = {"=": lambda a, b: a == b, "<>": lambda a, b: a != b, "<": lambda a, b: a < b, "<=": lambda a, b: a <= b, ">": lambda a, b: a > b, ">=": lambda a, b: a >= b}


def ev(e, row):
    """the value of a dump expression on one row (None = NULL; comparisons with NULL are NULL, AND / OR three-valued)"""
    if e["type"] == "expression_leaf":
        if e["leaf_type"] == "column":
            return row[e["displayName"]]
        if e["leaf_type"] == "constant":
            return e  # resolved against the other operand
        if e["leaf_type"] == "null":
            return None
        raise ValueError(e["leaf_type"])
    ss, subs = e["strings"], e["subExpressions"]
    val = lambda k, like=None: (lambda x: _constant(x, like) if isinstance(x, dict) else x)(ev(subs[k], row))
    if len(ss) == 3 and ss[0] == "" and ss[2] == "" and ss[1] in CMP:
        a, b = ev(subs[0], row), ev(subs[1], row)
        if isinstance(a, dict):
            a = _constant(a, None if isinstance(b, dict) else b)
        if isinstance(b, dict):
            b = _constant(b, a)
        return None if a is None or b is None else CMP[ss[1]](a, b)
    if len(ss) >= 3 and ss[0] == "" and ss[1] == " and ":
        vs = [val(k) for k in range(len(subs))]
        return False if any(v is False for v in vs) else (None if any(v is None for v in vs) else True)
    if len(ss) >= 3 and ss[0] == "(" and ss[1] == " or ":
        vs = [val(k) for k in range(len(subs))]
        return True if any(v is True for v in vs) else (None if any(v is None for v in vs) else False)
    if ss == ["not ", ""]:
        v = val(0)
        return None if v is None else not v
    if ss == ["", " is null"]:
        return val(0) is None
    if len(ss) >= 3 and ss[1] == " in [":
        a = val(0)
        return None if a is None else any(a == _constant(ev(s, row), a) for s in subs[1:])
    if ss == ["", " between ", " and ", ""]:
        a = val(0)
        lo, hi = _constant(ev(subs[1], row), a), _constant(ev(subs[2], row), a)
        return None if a is None else lo <= a <= hi
    if len(ss) == 3 and ss[0] == "" and ss[2] == "" and ss[1] in (" + ", " - ", " * "):
        a, b = val(0), val(1)
        return None if a is None or b is None else (a + b if ss[1] == " + " else a - b if ss[1] == " - " else a * b)
    if ss[0] in ("ConstLike(", "Like(") and len(subs) == 2:
        import re

        a, pat = val(0), ev(subs[1], row)["value"]
        rx = "".join(".*" if c == "%" else "." if c == "_" else re.escape(c) for c in pat)
        return None if a is None else re.fullmatch(rx, a, re.S) is not None
    raise ValueError("expression %r" % ss)


def truth(v):
    return v is True or (v not in (None, False) and bool(v))


def evaluate(node, tables):
    """rows of a relational-algebra tree; tables: {table name: pyarrow table}"""
    if isinstance(node, L.Table):
        rows = table_rows(tables[node.table], node.alias)
        for c, op, v in node.filters:
            name = "%s::%s" % (node.alias, c)
            sym = {"EQ": "=", "NEQ": "<>", "LT": "<", "LTE": "<=", "GT": ">", "GTE": ">="}.get(op)
            if op == "IN":
                rows = [r for r in rows if r[name] is not None and any(r[name] == _constant({"value": x}, r[name]) for x in v)]
            else:
                rows = [r for r in rows if r[name] is not None and CMP[sym](r[name], _constant({"value": v}, r[name]))]
        return rows
    if isinstance(node, L.Select):
        return [r for r in evaluate(node.child, tables) if all(truth(ev(e, r)) for e in node.conjuncts)]
    if isinstance(node, L.Map):
        out = []
        for r in evaluate(node.child, tables):
            r = dict(r)
            for c, e in node.computed:
                v = ev(e, r)
                r[c.name] = v
            out.append(r)
        return out
    if isinstance(node, L.Rename):
        return [{**r, **{n.name: r[o.name] for n, o in node.renamed}} for r in evaluate(node.child, tables)]
    if isinstance(node, L.Join):
        probe, build = evaluate(node.probe, tables), evaluate(node.build, tables)
        index = {}
        for j, b in enumerate(build):
            k = tuple(b[bk.name] for _, bk in node.keys)
            if None not in k:
                index.setdefault(k, []).append(j)
        matches = []
        for p in probe:
            k = tuple(p[pk.name] for pk, _ in node.keys)
            ms = []
            if None not in k:
                for j in index.get(k, []):
                    row = {**build[j], **p}
                    if all(truth(ev(e, row)) for e in node.residual):
                        ms.append(j)
            matches.append(ms)
        hit = {j for ms in matches for j in ms}
        kind = node.kind
        if kind == "inner":
            return [{**build[j], **p} for p, ms in zip(probe, matches) for j in ms]
        if kind in ("semi", "anti"):
            if node.reverse:
                return [b for j, b in enumerate(build) if (j in hit) == (kind == "semi")]
            return [p for p, ms in zip(probe, matches) if bool(ms) == (kind == "semi")]
        if kind == "mark":
            return [{**p, node.mark.name: bool(ms)} for p, ms in zip(probe, matches)]
        pa_, ba = node.probe.avail(), node.build.avail()
        if kind in ("outer", "single") and not node.reverse:
            out = []
            for p, ms in zip(probe, matches):
                for j in ms:
                    out.append({**p, **{n.name: build[j][o.name] for n, o in node.mapping}})
                if not ms:
                    out.append({**p, **{n.name: None for n, _ in node.mapping}})
            return out
        if kind in ("outer", "single"):  # reverseSides: the build side is preserved, the mapping renames PROBE columns
            out = [{**build[j], **{n.name: p[o.name] for n, o in node.mapping}} for p, ms in zip(probe, matches) for j in ms]
            return out + [{**b, **{n.name: None for n, _ in node.mapping}} for j, b in enumerate(build) if j not in hit]
        if kind == "full":
            both = lambda p, b: {n.name: (p[o.name] if o.name in pa_ else b[o.name]) if (p if o.name in pa_ else b) is not None else None for n, o in node.mapping}
            out = [both(p, build[j]) for p, ms in zip(probe, matches) for j in ms]
            out += [both(p, None) for p, ms in zip(probe, matches) if not ms]
            return out + [both(None, b) for j, b in enumerate(build) if j not in hit]
        raise ValueError(kind)
    if isinstance(node, L.Aggregate):
        groups, order = {}, []
        rows = evaluate(node.child, tables)
        for r in rows:
            k = tuple(r[c.name] for c in node.keys)
            if k not in groups:
                groups[k] = []
                order.append(k)
            groups[k].append(r)
        if not node.keys and not order:
            groups[()] = []
            order.append(())
        out = []
        for k in order:
            row = {c.name: k[i] for i, c in enumerate(node.keys)}
            for fn, a, o in node.aggs:
                vals = [r[a.name] for r in groups[k] if r[a.name] is not None] if a is not None else None
                row[o.name] = (len(groups[k]) if fn == "count_star" else len(vals) if fn == "count" else (sum(vals) if vals else None) if fn == "sum" else
                               (min(vals) if vals else None) if fn == "min" else (max(vals) if vals else None) if fn == "max" else (groups[k][0][a.name] if groups[k] else None))
            out.append(row)
        return out
    if isinstance(node, L.Distinct):
        seen = {}
        for r in evaluate(node.child, tables):
            seen.setdefault(tuple(r[c.name] for c in node.keys), None)
        return [{c.name: k[i] for i, c in enumerate(node.keys)} for k in seen]
    if isinstance(node, (L.Sort, L.Tmp)):
        return evaluate(node.child, tables)
    if isinstance(node, L.SetOp):
        import collections

        l = [tuple(r[lc.name] for _, lc, _ in node.mapping) for r in evaluate(node.left, tables)]
        r = [tuple(x[rc.name] for _, _, rc in node.mapping) for x in evaluate(node.right, tables)]
        cl, cr = collections.Counter(l), collections.Counter(r)
        rows = {"union_all": l + r, "union": list(dict.fromkeys(l + r)), "intersect": [k for k in dict.fromkeys(l) if k in cr], "except": [k for k in dict.fromkeys(l) if k not in cr],
                "intersect_all": list((cl & cr).elements()), "except_all": list((cl - cr).elements())}[node.kind]
        return [{n.name: k[i] for i, (n, _, _) in enumerate(node.mapping)} for k in rows]
    if isinstance(node, L.ConstJoin):
        one = evaluate(node.single, tables)
        assert len(one) == 1, "a constant join's single side has one row"
        return [{**r, **{n.name: one[0][o.name] for n, o in node.mapping}} for r in evaluate(node.left, tables)]
    if isinstance(node, L.GroupJoin):
        left, right = evaluate(node.left, tables), evaluate(node.right, tables)
        groups = {}
        for l in left:  # one entry per key: the map keeps the first tuple's stored columns … the inputs here have unique keys
            groups.setdefault(tuple(l[lk.name] for lk, _ in node.keys), (l, []))
        for r in right:
            k = tuple(r[rk.name] for _, rk in node.keys)
            if None in k or k not in groups:
                continue
            row = {**groups[k][0], **r, **{lk.name: r[rk.name] for lk, rk in node.keys}}
            if all(truth(ev(e, row)) for e in node.predicate):
                groups[k][1].append(r)
        out = []
        for k, (l, rs) in groups.items():
            if node.inner and not rs:
                continue
            row = {rk.name: k[i] for i, (_, rk) in enumerate(node.keys)}
            row.update({lk.name: k[i] for i, (lk, _) in enumerate(node.keys)})
            row.update({c.name: l[c.name] for c in node.stored})
            for fn, a, o in node.aggs:
                vals = [r[a.name] for r in rs if r[a.name] is not None] if a is not None else None
                row[o.name] = len(rs) if fn == "count_star" else len(vals) if fn == "count" else (sum(vals) if vals else None) if fn == "sum" else (min(vals) if vals else None) if fn == "min" else (max(vals) if vals else None)
            out.append(row)
        return out
    if isinstance(node, L.Window):
        rows = evaluate(node.child, tables)
        order = sorted(range(len(rows)), key=lambda i: tuple(rows[i][c.name] for c in node.partition_by) + tuple(rows[i][c.name] for c, _ in node.order_by))
        assert all(d == "asc" for _, d in node.order_by)
        out = []
        pos = 0
        while pos < len(order):
            end = pos
            key = tuple(rows[order[pos]][c.name] for c in node.partition_by)
            while end + 1 < len(order) and tuple(rows[order[end + 1]][c.name] for c in node.partition_by) == key:
                end += 1
            part = [rows[i] for i in order[pos:end + 1]]
            frm, to = node.frame
            for p, r in enumerate(part):
                a = 0 if frm == L.I64_MIN else max(0, min(len(part) - 1, p + frm))
                b = len(part) - 1 if to == L.I64_MAX else max(0, min(len(part) - 1, p + to))
                row = dict(r)
                for fn, c, o in node.fns:
                    vals = [x[c.name] for x in part[a:b + 1] if x[c.name] is not None] if c is not None else None
                    row[o.name] = (p - a + 1 if fn == "rank" else b - a + 1 if fn == "count_star" else len(vals) if fn == "count" else (sum(vals) if vals else None) if fn == "sum" else
                                   (min(vals) if vals else None) if fn == "min" else (max(vals) if vals else None))
                out.append(row)
            pos = end + 1
        return out
    raise TypeError(type(node).__name__)
