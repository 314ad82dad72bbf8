#!/usr/bin/env python3
"""RelAlg → SubOp, restated in Python for the operators TPC-H needs, printing the JSON of the reference's
`tools/ct/mlir-subop-to-json.cpp` (the tool cannot be built here: MLIR).  tools/write_subop_dumps.py authors eight dumps
sub-operator by sub-operator; this file is the other way round — each class below restates ONE lowering pattern of
src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp (cited per class) over a small relational-algebra tree, so a query
is ~20 lines (tools/write_subop_dumps_relalg.py) and every query exercises the same sub-operator sequences the reference's
lowering would print.  Test infrastructure only: the product is the CONSUMER of these documents (lingo-db_amd/host/ldb_subop.cpp).

What is modelled (after the Specialize pass turned MultiMap + insert into Buffer + materialize + create_hash_indexed_view,
the form write_subop_dumps.py already uses):

  BaseTableLowering        :103-141   get_external (+ pushed-down FilterDescriptions) → scan
  translateSelection       :173-258   one map(pred) + filter(all_true) PER CONJUNCT
  MapLowering              :276-301   map
  translateHJ              :1097-1128 build: map hash → materialize(Buffer) → create_hash_indexed_view;
                                      probe: map hash → lookup → nested_map{scan_list → gather → combine_tuple → selection}
  anyTuple                 :1296-1305 create_simple_state(marker) → map true → lookup → scatter → scan(marker)
  SemiJoin / AntiSemiJoin  :1340-1445 probe side kept: anyTuple + filter all_true / none_true; reverseSides: flag member in
                                      the build buffer, scatter true, scan of the buffer + filter on the flag
  MarkJoin                 :1376-1408 anyTuple defining the mark column
  OuterJoin / SingleJoin   :1486-1587 anyTuple + filter none_true + map(nulls), map(as-nullable) of the matches, union;
                                      reverseSides: flag member, matches → map(as-nullable); unmatched build rows → map(nulls); union
                                      constantJoin: simple state scattered by the one-row side, lookup + gather by the other
  AggregationLowering      :2128-2190, 2554-2680 create_simple_state | generic_create(map) → lookup | lookup_or_insert → reduce → scan
  ProjectionDistinct       :337-393   lookup_or_insert with an empty reduce → scan of the keys
  SortLowering / TopK      :1668-1740 materialize(Buffer) → create_sorted_view → scan | create_heap → materialize → scan
  TmpLowering              :1742-1760 materialize(Buffer) once, one scan per consumer
  UnionAll                 :622-634   renaming of both inputs to the result columns + union
  MaterializeLowering      :1762-1784 materialize(ResultTable)

Emitter extensions (fields the tool does not print today): E1 " - " for db.sub, E4 sortBy / maxRows,
E5 primaryKey, E6 combine_tuple — as in write_subop_dumps.py — and E7: arith.ori / andi / cmpi / subi / addi (the nullable MIN / MAX / SUM
bodies, RelAlgToSubOp.cpp:1865-1870; the counters of INTERSECT / EXCEPT; the rank) printed like db.or / db.and / db.cmp / db.sub / db.add instead of an
"unknown" leaf; E8: `aggregates` on create_segment_tree_view and `keys` on lookup; E9: `materialized` on the reduce that fills a window partition's buffer."""
import collections

import write_subop_dumps as W
from write_subop_dumps import add, and_, arg, cast, column, const, div, eq, hash_, if_, inner, isnull, lt, member, mul, neq, node, or_, select, sub, unknown

W.TYPES.update({"p_partkey": "int32", "p_size": "int32", "p_retailprice": "decimal(12,2)", "p_name": "str", "p_type": "str", "p_brand": "str", "p_container": "str", "p_mfgr": "str",
                "ps_partkey": "int32", "ps_suppkey": "int32", "ps_availqty": "int32", "ps_supplycost": "decimal(12,2)"})
W.TABLES.update({"part": [c for c in W.TYPES if c.startswith("p_")], "partsupp": [c for c in W.TYPES if c.startswith("ps_")]})
W.PKEY.update({"part": ["p_partkey"], "partsupp": ["ps_partkey", "ps_suppkey"]})
# the generated test database has fewer columns than dbgen's (tests/tpch_data.py): get_external maps what exists
for _t, _drop in (("lineitem", ["l_comment"]), ("orders", ["o_clerk"]), ("customer", ["c_address", "c_comment"]), ("nation", ["n_comment"]), ("region", ["r_comment"])):
    W.TABLES[_t] = [c for c in W.TABLES[_t] if c not in _drop]


def gt(a, b): return inner(["", ">", ""], [a, b])
def gte(a, b): return inner(["", ">=", ""], [a, b])
def lte(a, b): return inner(["", "<=", ""], [a, b])
def not_(a): return inner(["not ", ""], [a])
def between(x, lo, hi, lower_inclusive=True, upper_inclusive=True):
    """db.between with its two I1 attributes (DBOps.td:507) — EXT E10: the unpatched tool prints neither"""
    return dict(inner(["", " between ", " and ", ""], [x, lo, hi]), lowerInclusive=lower_inclusive, upperInclusive=upper_inclusive)
def one_of(x, vals): return inner(["", " in ["] + [", "] * (len(vals) - 1) + ["]"], [x] + list(vals))  # db.oneof
def call(fn, *args): return inner([fn + "("] + [", "] * (len(args) - 1) + [")"], list(args))  # db.runtime_call
def null(): return {"type": "expression_leaf", "leaf_type": "null"}
def sconst(s): return const(s, "string")
def date(s): return const(s, "date")
def dec(text, p, s): return const(text, "decimal(%d,%d)" % (p, s))  # db.constant("0.2") : !db.decimal<p,s> (a StringAttr; integers come as IntegerAttr: const(1, "decimal(12,2)"))


def refs(e, out=None):
    """display names of the columns an expression reads"""
    out = set() if out is None else out
    if isinstance(e, dict):
        if e.get("leaf_type") == "column":
            out.add(e["displayName"])
        for s in e.get("subExpressions", []):
            refs(s, out)
    return out


TYPE_OF = {}  # display name → type of every column defined so far


class C:
    """a tuples column: scope::name + type"""
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype
        self.j = column(name, dtype)
        TYPE_OF[name] = dtype  # buffers re-define a column with the type it was created with

    @property
    def base(self): return self.name.split("::")[-1]


class Cx:
    def __init__(self, name):
        self.d = W.Dump(name)
        self.n = collections.Counter()
        self.externals = {}

    def scope(self, stem):
        self.n[stem] += 1
        return "%s%d" % (stem, self.n[stem] - 1)

    def col(self, stem, name, dtype): return C("%s::%s" % (self.scope(stem), name), dtype)

    def state(self, kind, ty="?", **fields):
        """a state created in its own execution step; returns (step ref, type)"""
        op = self.d.subop(kind, **fields)
        return self.d.step([op], results=[(ty, op["ref"], 0)]), ty


class Pipe:
    """the sub-operators of the execution step under construction"""
    def __init__(self, cx):
        self.cx, self.ops, self.inputs, self.last = cx, [], [], None

    def state_arg(self, step_ref, ty="?", resnr=0):
        for i, (_, s, r) in enumerate(self.inputs):
            if s == step_ref and r == resnr:
                return arg(i)
        self.inputs.append((ty, step_ref, resnr))
        return arg(len(self.inputs) - 1)

    def raw(self, kind, streams=(), **fields):
        return self.cx.d.subop(kind, streams=list(streams), **fields)

    def op(self, kind, source=False, **fields):
        o = self.raw(kind, [] if source or self.last is None else [self.last], **fields)
        self.ops.append(o)
        self.last = o["ref"]
        return o

    def close(self):
        return self.cx.d.step(self.ops, inputs=self.inputs)

    def absorb(self, other):
        """the sub-operators of another pipeline of the same execution step (two scans meeting in a union): its state arguments are renumbered"""
        remap = {}
        for i, (ty, s, r) in enumerate(other.inputs):
            remap[i] = self.state_arg(s, ty, r)["argnr"]

        def walk(x):
            if isinstance(x, dict):
                if x.get("type") == "parentArg":
                    x["argnr"] = remap[x["argnr"]]
                for v in x.values():
                    walk(v)
            elif isinstance(x, list):
                for v in x:
                    walk(v)
        walk(other.ops)
        self.ops += other.ops


class Node:
    def avail(self): raise NotImplementedError
    def lower(self, cx, required): raise NotImplementedError


def selection(p, conjuncts, ops=None, last=None):
    """translateSelection: map + filter per conjunct; inside a nested_map body the ops go to `ops` and chain from `last`"""
    for e in conjuncts:
        pred = p.cx.col("map", "pred", "int1")
        if ops is None:
            p.op("map", computed=[{"computed": pred.j, "expression": e}])
            p.op("filter", semantic="all_true", columns=[pred.j])
        else:
            m = p.raw("map", [last], computed=[{"computed": pred.j, "expression": e}])
            f = p.raw("filter", [m["ref"]], semantic="all_true", columns=[pred.j])
            ops += [m, f]
            last = f["ref"]
    return last


class Table(Node):
    """BaseTableLowering; `alias` is the scope of the columns (n1, n2 …), `filters` the pushed-down restrictions"""
    def __init__(self, table, alias=None, filters=()):
        self.table, self.alias, self.filters = table, alias or table, list(filters)

    def c(self, name): return C("%s::%s" % (self.alias, name), W.TYPES[name])
    def __getitem__(self, name): return self.c(name)
    def avail(self): return {"%s::%s" % (self.alias, c) for c in W.TABLES[self.table]}

    def lower(self, cx, required):
        step, ty = W.get_external(cx.d, self.table, self.filters)
        p = Pipe(cx)
        cols = [c for c in W.TABLES[self.table] if "%s::%s" % (self.alias, c) in required] or W.TABLES[self.table][:1]
        p.op("scan", source=True, accesses=[p.state_arg(step, ty)], mapping=[{"member": "%s$0" % c, "column": self.c(c).j} for c in cols])
        return p


class Select(Node):
    def __init__(self, child, *conjuncts):
        self.child, self.conjuncts = child, list(conjuncts)

    def avail(self): return self.child.avail()

    def lower(self, cx, required):
        need = set(required)
        for e in self.conjuncts:
            need |= refs(e)
        p = self.child.lower(cx, need & self.child.avail())
        selection(p, self.conjuncts)
        return p


class Map(Node):
    def __init__(self, child, computed):
        self.child, self.computed = child, list(computed)  # [(C, expression)]

    def avail(self): return self.child.avail() | {c.name for c, _ in self.computed}

    def lower(self, cx, required):
        need = set(required) - {c.name for c, _ in self.computed}
        for _, e in self.computed:
            need |= refs(e)
        p = self.child.lower(cx, need & self.child.avail())
        p.op("map", computed=[{"computed": c.j, "expression": e} for c, e in self.computed])
        return p


class Join(Node):
    """hash joins.  kind: inner | semi | anti | mark | outer | single; `keys` = [(probe column, build column)], `residual`
    further conjuncts of the join predicate; reverse = the reference's `reverseSides` (semi / anti / outer: the BUILD side is
    the preserved one and carries a flag member); `mapping` = [(new column, build column)] for outer / single (their
    `mapping` attribute: the nullable copies), `mark` the mark column of a mark join"""
    def __init__(self, kind, probe, build, keys, residual=(), reverse=False, mapping=(), mark=None):
        self.kind, self.probe, self.build, self.keys, self.residual, self.reverse = kind, probe, build, list(keys), list(residual), reverse
        self.mapping, self.mark = list(mapping), mark

    def avail(self):
        if self.kind == "inner":
            return self.probe.avail() | self.build.avail()
        if self.kind in ("semi", "anti"):
            return self.build.avail() if self.reverse else self.probe.avail()
        if self.kind == "mark":
            return self.probe.avail() | {self.mark.name}
        if self.kind == "full":  # every column of both sides comes out as its nullable copy
            return {n.name for n, _ in self.mapping}
        if self.reverse:  # outer with reverseSides: the build side is preserved, the probe side's columns come out nullable
            return self.build.avail() | {n.name for n, _ in self.mapping}
        return self.probe.avail() | {n.name for n, _ in self.mapping}

    def lower(self, cx, required):
        pa, ba = self.probe.avail(), self.build.avail()
        need = set(required)
        for n, old in self.mapping:  # the mapping's sources are used columns of the join, whether or not the copies are read above
            need.add(old.name)
        for pk, bk in self.keys:
            need |= {pk.name, bk.name}
        for e in self.residual:
            need |= refs(e)
        flagged = (self.reverse and self.kind in ("semi", "anti", "outer", "single")) or self.kind == "full"
        # ---- build side: translateHJ / translateHJWithMarker
        bp = self.build.lower(cx, need & ba)
        s_buf, _ = cx.state("generic_create", "Buffer[...]")
        n = cx.scope("b")
        flag = None
        if flagged:
            flag = cx.col("materialized", "marker", "int1")
            bp.op("map", computed=[{"computed": flag.j, "expression": const(False, "int1")}])  # mapBool(left, false, marker)
        bkeys = [bk for _, bk in self.keys]
        h = cx.col("hj", "hash", "index")
        bp.op("map", computed=[{"computed": h.j, "expression": hash_(*[k.j for k in bkeys])}])
        payload, seen = [], set()
        for name in [k.name for k in bkeys] + sorted((need & ba) - {k.name for k in bkeys}):
            if name not in seen:
                seen.add(name)
                payload.append(name)
        types = TYPE_OF
        members = {name: "%s$%s" % (name.replace("::", "_"), n) for name in payload}
        mapping = [{"member": "hash$%s" % n, "column": h.j}] + [{"member": members[name], "column": column(name, types.get(name, "?"))} for name in payload]
        flag_member = "flag$%s" % n
        if flagged:
            mapping.append({"member": flag_member, "column": flag.j})
        bp.op("materialize", accesses=[bp.state_arg(s_buf, "Buffer[...]")], stateType="Buffer", mapping=mapping)
        bp.close()
        v = cx.d.subop("create_hash_indexed_view", accesses=[arg(0)])
        s_v = cx.d.step([v], inputs=[("Buffer[...]", s_buf, 0)], results=[("?", v["ref"], 0)])
        # ---- probe side
        pp = self.probe.lower(cx, need & pa)
        hp = cx.col("hj", "hash", "index")
        pp.op("map", computed=[{"computed": hp.j, "expression": hash_(*[pk.j for pk, _ in self.keys])}])
        sc = cx.scope("lookup")
        lst, ent = C(sc + "::list", "?"), C(sc + "::entryref", "?")
        lk = pp.op("lookup", accesses=[pp.state_arg(s_v)], stateType="HashIndexedView", reference=lst.j)
        body = []
        sl = pp.raw("scan_list", accesses=[{"type": "nested_map_arg", "column": lst.j, "id": "pending"}], elem=ent.j)
        ga = pp.raw("gather", [sl["ref"]], reference=ent.j, mapping=[{"member": members[name], "column": column(name, types.get(name, "?"))} for name in payload])
        ct = pp.raw("combine_tuple", [ga["ref"]])  # EXT E6
        body += [sl, ga, ct]
        filtered = selection(pp, [eq(pk.j, bk.j) for pk, bk in self.keys] + self.residual, body, ct["ref"])
        out = filtered
        if self.kind in ("semi", "anti", "mark") and not self.reverse:
            mk = self.mark if self.kind == "mark" else cx.col("marker", "marker", "int1")
            out = self._any_tuple(pp, body, filtered, mk)
            if self.kind != "mark":
                f = pp.raw("filter", [out], semantic="all_true" if self.kind == "semi" else "all_false", columns=[mk.j])
                body.append(f)
                out = f["ref"]
        elif self.kind in ("outer", "single") and not self.reverse:
            mk = cx.col("marker", "marker", "int1")
            m = self._any_tuple(pp, body, filtered, mk)
            f = pp.raw("filter", [m], semantic="all_false", columns=[mk.j])
            nulls = pp.raw("map", [f["ref"]], computed=[{"computed": nw.j, "expression": null()} for nw, _ in self.mapping])  # mapColsToNull
            nullable = pp.raw("map", [filtered], computed=[{"computed": nw.j, "expression": old.j} for nw, old in self.mapping])  # mapColsToNullable (db.as_nullable prints its operand)
            u = pp.raw("union", [nullable["ref"], nulls["ref"]])
            body += [f, nulls, nullable, u]
            out = u["ref"]
        elif flagged:  # the matches set the flag of their build entry
            mk = cx.col("marker", "marker", "int1")
            mb = pp.raw("map", [filtered], computed=[{"computed": mk.j, "expression": const(True, "int1")}])
            sca = pp.raw("scatter", [mb["ref"]], reference=ent.j, mapping=[{"member": flag_member, "column": mk.j}])
            body += [mb, sca]
            out = None
            if self.kind in ("outer", "single"):
                nullable = pp.raw("map", [filtered], computed=[{"computed": nw.j, "expression": old.j} for nw, old in self.mapping])
                body.append(nullable)
                out = nullable["ref"]
            if self.kind == "full":  # FullOuterJoinLowering (:1446-1484): the matches ∪ the probe rows without a partner, build side NULL
                probe_side = [(nw, old) for nw, old in self.mapping if old.name in pa]
                build_side = [(nw, old) for nw, old in self.mapping if old.name not in pa]
                nullable = pp.raw("map", [filtered], computed=[{"computed": nw.j, "expression": old.j} for nw, old in self.mapping])
                mk2 = cx.col("marker", "marker", "int1")
                body.append(nullable)
                m = self._any_tuple(pp, body, filtered, mk2)
                f = pp.raw("filter", [m], semantic="all_false", columns=[mk2.j])
                ct2 = pp.raw("combine_tuple", [f["ref"]])
                keep = pp.raw("map", [ct2["ref"]], computed=[{"computed": nw.j, "expression": old.j} for nw, old in probe_side])
                nulls = pp.raw("map", [keep["ref"]], computed=[{"computed": nw.j, "expression": null()} for nw, _ in build_side])
                u = pp.raw("union", [nullable["ref"], nulls["ref"]])
                body += [f, ct2, keep, nulls, u]
                out = u["ref"]
        nm = pp.raw("nested_map", [lk["ref"]], inputs=[], subops=body)
        sl["accesses"][0]["id"] = nm["ref"] + "_0"
        pp.ops.append(nm)
        pp.last = nm["ref"]
        if not flagged:
            return pp
        if self.kind in ("semi", "anti"):
            pp.close()
            sp = Pipe(cx)
            keep = [name for name in payload if name in required] or payload[:1]
            sp.op("scan", source=True, accesses=[sp.state_arg(s_buf, "Buffer[...]")],
                  mapping=[{"member": members[name], "column": column(name, types.get(name, "?"))} for name in keep] + [{"member": flag_member, "column": flag.j}])
            sp.op("filter", semantic="all_true" if self.kind == "semi" else "all_false", columns=[flag.j])
            return sp
        # outer with reverseSides: the stream of matches ∪ the unmatched build rows with NULLs for the probe side
        sb = pp.raw("scan", accesses=[pp.state_arg(s_buf, "Buffer[...]")],
                    mapping=[{"member": members[name], "column": column(name, types.get(name, "?"))} for name in payload] + [{"member": flag_member, "column": flag.j}])
        fb = pp.raw("filter", [sb["ref"]], semantic="all_false", columns=[flag.j])
        if self.kind == "full":  # the unmatched build rows: their own columns as nullable copies, the probe side NULL
            keep = pp.raw("map", [fb["ref"]], computed=[{"computed": nw.j, "expression": old.j} for nw, old in self.mapping if old.name not in pa])
            nulls = pp.raw("map", [keep["ref"]], computed=[{"computed": nw.j, "expression": null()} for nw, old in self.mapping if old.name in pa])
            u = pp.raw("union", [pp.last, nulls["ref"]])
            pp.ops += [sb, fb, keep, nulls, u]
            pp.last = u["ref"]
            return pp
        nulls = pp.raw("map", [fb["ref"]], computed=[{"computed": nw.j, "expression": null()} for nw, _ in self.mapping])
        u = pp.raw("union", [pp.last, nulls["ref"]])
        pp.ops += [sb, fb, nulls, u]
        pp.last = u["ref"]
        return pp

    @staticmethod
    def _any_tuple(pp, body, stream, mk):
        cx = pp.cx
        ms = pp.raw("create_simple_state")  # createMarkerState: <[marker$0 : i1]>, initial false
        bv = cx.col("map", "boolval", "int1")
        mb = pp.raw("map", [stream], computed=[{"computed": bv.j, "expression": const(True, "int1")}])
        mref = cx.col("lookup", "ref", "?")
        ml = pp.raw("lookup", [mb["ref"]], accesses=[node(ms["ref"])], stateType="SimpleState", reference=mref.j)
        sca = pp.raw("scatter", [ml["ref"]], reference=mref.j, mapping=[{"member": "marker$0", "column": bv.j}])
        sm = pp.raw("scan", accesses=[node(ms["ref"])], mapping=[{"member": "marker$0", "column": mk.j}])
        body += [ms, mb, ml, sca, sm]
        return sm["ref"]


class ConstJoin(Node):
    """SingleJoinLowering with `constantJoin` (:1540-1556): the one-row side is scattered into a simple state, every tuple
    of the other side gathers it; `mapping` = [(new nullable column, column of the one-row side)]"""
    def __init__(self, left, single, mapping):
        self.left, self.single, self.mapping = left, single, list(mapping)

    def avail(self): return self.left.avail() | {n.name for n, _ in self.mapping}

    def lower(self, cx, required):
        s_cs, _ = cx.state("create_simple_state")
        sp = self.single.lower(cx, {old.name for _, old in self.mapping})
        ref = cx.col("lookup", "entryref", "?")
        sp.op("lookup", accesses=[sp.state_arg(s_cs)], stateType="SimpleState", reference=ref.j)
        members = {old.name: "%s$c" % old.base for _, old in self.mapping}
        sp.op("scatter", reference=ref.j, mapping=[{"member": members[old.name], "column": old.j} for _, old in self.mapping])
        sp.close()
        lp = self.left.lower(cx, set(required) & self.left.avail())
        ref2 = cx.col("lookup", "entryref", "?")
        lp.op("lookup", accesses=[lp.state_arg(s_cs)], stateType="SimpleState", reference=ref2.j)
        lp.op("gather", reference=ref2.j, mapping=[{"member": members[old.name], "column": old.j} for _, old in self.mapping])
        lp.op("map", computed=[{"computed": nw.j, "expression": old.j} for nw, old in self.mapping])  # mapColsToNullable
        return lp


def agg_body(fn, m, a, nullable_state, nullable_arg=False):
    """the reduce expression of one aggregate (Sum / Count / CountStar / Min / Max / Any AggrFunc::aggregate, :1809-2025)"""
    mem = member(m)
    if fn == "count_star":
        return add(mem, const(1, "int64"))
    if fn == "count":
        return select(isnull(a), mem, add(mem, const(1, "int64"))) if nullable_arg else add(mem, const(1, "int64"))
    if fn == "sum":
        if nullable_state and nullable_arg:  # as_nullable(select(isnull(state), 0, val(state)) + select(isnull(arg), 0, val(arg)), …)
            return add(select(isnull(mem), const(0, "int64"), unknown()), select(isnull(a), const(0, "int64"), unknown()))
        return select(isnull(mem), a, add(unknown(), a)) if nullable_state else add(mem, a)
    if fn in ("min", "max"):
        cmp = inner(["", ">" if fn == "min" else "<", ""], [mem, a])
        return select(or_(cmp, isnull(mem)), a, mem) if nullable_state else select(cmp, a, mem)  # EXT E7: arith.ori
    if fn == "any":
        return a
    raise ValueError(fn)


class Aggregate(Node):
    """AggregationLowering + performAggregation; aggs = [(fn, argument column | None, result column)]; SQL's result types:
    a key-less aggregation has nullable SUM / MIN / MAX states (no row → NULL)"""
    def __init__(self, child, keys, aggs, nullable_args=()):
        self.child, self.keys, self.aggs, self.nullable_args = child, list(keys), list(aggs), {c.name for c in nullable_args}

    def avail(self): return {k.name for k in self.keys} | {o.name for _, _, o in self.aggs}

    def lower(self, cx, required):
        need = {k.name for k in self.keys} | {a.name for _, a, _ in self.aggs if a is not None}
        p = self.child.lower(cx, need)
        ref = cx.col("lookup", "ref", "?")
        if self.keys:
            s_st, _ = cx.state("generic_create")
            p.op("lookup_or_insert", accesses=[p.state_arg(s_st)], stateType="HashMap", reference=ref.j)
        else:
            s_st, _ = cx.state("create_simple_state")
            p.op("lookup", accesses=[p.state_arg(s_st)], stateType="SimpleState", reference=ref.j)
        upd = []
        for i, (fn, a, _) in enumerate(self.aggs):
            m = "aggrVal$%d" % i
            na = a is not None and a.name in self.nullable_args
            upd.append({"member": m, "expression": agg_body(fn, m, a.j if a is not None else None, not self.keys or na, na)})
        p.op("reduce", reference=ref.j, updated=upd)
        p.close()
        sp = Pipe(cx)
        sp.op("scan", source=True, accesses=[sp.state_arg(s_st)],
              mapping=[{"member": "keyval$%d" % i, "column": k.j} for i, k in enumerate(self.keys)] + [{"member": "aggrVal$%d" % i, "column": o.j} for i, (_, _, o) in enumerate(self.aggs)])
        return sp


class Distinct(Node):
    """ProjectionDistinctLowering: a map with keys only and an empty reduce"""
    def __init__(self, child, keys):
        self.child, self.keys = child, list(keys)

    def avail(self): return {k.name for k in self.keys}

    def lower(self, cx, required):
        p = self.child.lower(cx, {k.name for k in self.keys})
        s_st, _ = cx.state("generic_create")
        ref = cx.col("lookup", "ref", "?")
        p.op("lookup_or_insert", accesses=[p.state_arg(s_st)], stateType="HashMap", reference=ref.j)
        p.op("reduce", reference=ref.j, updated=[])
        p.close()
        sp = Pipe(cx)
        sp.op("scan", source=True, accesses=[sp.state_arg(s_st)], mapping=[{"member": "keyval$%d" % i, "column": k.j} for i, k in enumerate(self.keys)])
        return sp


def _materialize_all(cx, p, names, state_step, state_ty, state_type, tag):
    members = {name: "%s$%s" % (name.replace("::", "_"), tag) for name in names}
    p.op("materialize", accesses=[p.state_arg(state_step, state_ty)], stateType=state_type, mapping=[{"member": members[n], "column": column(n, TYPE_OF.get(n, "?"))} for n in names])
    return members


class Sort(Node):
    def __init__(self, child, by):
        self.child, self.by = child, list(by)  # [(C, "asc" | "desc")]

    def avail(self): return self.child.avail()

    def lower(self, cx, required):
        names = sorted(set(required) | {c.name for c, _ in self.by})
        p = self.child.lower(cx, set(names))
        s_buf, ty = cx.state("generic_create", "Buffer[...]")
        tag = cx.scope("s")
        members = _materialize_all(cx, p, names, s_buf, ty, "Buffer", tag)
        p.close()
        sv = cx.d.subop("create_sorted_view", accesses=[arg(0)], sortBy=[{"member": members[c.name], "direction": dr} for c, dr in self.by])  # EXT E4
        s_sv = cx.d.step([sv], inputs=[(ty, s_buf, 0)], results=[("SortedView Buffer[...]", sv["ref"], 0)])
        sp = Pipe(cx)
        sp.op("scan", source=True, accesses=[sp.state_arg(s_sv, "SortedView Buffer[...]")], mapping=[{"member": members[n], "column": column(n, TYPE_OF.get(n, "?"))} for n in names])
        return sp


class TopK(Node):
    def __init__(self, child, by, k):
        self.child, self.by, self.k = child, list(by), k

    def avail(self): return self.child.avail()

    def lower(self, cx, required):
        names = sorted(set(required) | {c.name for c, _ in self.by})
        tag = cx.scope("t")
        members = {n: "%s$%s" % (n.replace("::", "_"), tag) for n in names}
        s_hp, ty = cx.state("create_heap", maxRows=self.k, sortBy=[{"member": members[c.name], "direction": dr} for c, dr in self.by])  # EXT E4
        p = self.child.lower(cx, set(names))
        _materialize_all(cx, p, names, s_hp, ty, "Heap", tag)
        p.close()
        sp = Pipe(cx)
        sp.op("scan", source=True, accesses=[sp.state_arg(s_hp, ty)], mapping=[{"member": members[n], "column": column(n, TYPE_OF.get(n, "?"))} for n in names])
        return sp


class Tmp(Node):
    """TmpLowering: materialised once (by the first consumer that is lowered), scanned by every consumer"""
    def __init__(self, child, cols):
        self.child, self.cols, self.done = child, [c.name for c in cols], None

    def avail(self): return set(self.cols)

    def lower(self, cx, required):
        if self.done is None:
            p = self.child.lower(cx, set(self.cols))
            s_buf, ty = cx.state("generic_create", "Buffer[...]")
            tag = cx.scope("tmp")
            members = _materialize_all(cx, p, self.cols, s_buf, ty, "Buffer", tag)
            p.close()
            self.done = (s_buf, ty, members)
        s_buf, ty, members = self.done
        sp = Pipe(cx)
        sp.op("scan", source=True, accesses=[sp.state_arg(s_buf, ty)], mapping=[{"member": members[n], "column": column(n, TYPE_OF.get(n, "?"))} for n in self.cols])
        return sp


class Rename(Node):
    """RenamingLowering (:303-310): new column definitions for existing columns"""
    def __init__(self, child, renamed):
        self.child, self.renamed = child, list(renamed)  # [(new C, old C)]

    def avail(self): return (self.child.avail() - {o.name for _, o in self.renamed}) | {n.name for n, _ in self.renamed}

    def lower(self, cx, required):
        need = (set(required) - {n.name for n, _ in self.renamed}) | {o.name for n, o in self.renamed if n.name in required}
        p = self.child.lower(cx, need & self.child.avail())
        p.op("renaming", renamed=[{"new": n.j, "old": o.j} for n, o in self.renamed])
        return p


class SetOp(Node):
    """UnionAllLowering (:622-634), UnionDistinctLowering (:636-727), CountingSetOperationLowering (:728-915); kind: union_all | union |
    intersect | except | intersect_all | except_all; mapping = [(result column, left column, right column)]"""
    def __init__(self, kind, left, right, mapping):
        self.kind, self.left, self.right, self.mapping = kind, left, right, list(mapping)

    def avail(self): return {n.name for n, _, _ in self.mapping}

    def lower(self, cx, required):
        lp = self.left.lower(cx, {l.name for _, l, _ in self.mapping})
        lp.op("map", computed=[{"computed": n.j, "expression": l.j} for n, l, _ in self.mapping])  # mapColsToNullable(…, 0)
        rp = self.right.lower(cx, {r.name for _, _, r in self.mapping})
        rp.op("map", computed=[{"computed": n.j, "expression": r.j} for n, _, r in self.mapping])  # mapColsToNullable(…, 1)
        if self.kind == "union_all":
            lp.absorb(rp)
            u = lp.raw("union", [lp.last, rp.last])
            lp.ops.append(u)
            lp.last = u["ref"]
            return lp
        counting = self.kind != "union"
        s_st, _ = cx.state("generic_create")
        for i, p in enumerate((lp, rp)):
            ref = cx.col("lookup", "ref", "?")
            p.op("lookup_or_insert", accesses=[p.state_arg(s_st)], stateType="HashMap", reference=ref.j)
            upd = []
            if counting:  # the input's own counter + 1, the other one returned unchanged
                upd = [{"member": "counter$%d" % k, "expression": add(member("counter$%d" % k), const(1, "int64")) if k == i else member("counter$%d" % k)} for k in (0, 1)]
            p.op("reduce", reference=ref.j, updated=upd)
            p.close()
        sp = Pipe(cx)
        c1, c2 = cx.col("set", "counter", "int64"), cx.col("set", "counter", "int64")
        sp.op("scan", source=True, accesses=[sp.state_arg(s_st)],
              mapping=[{"member": "keyval$%d" % i, "column": n.j} for i, (n, _, _) in enumerate(self.mapping)] + ([{"member": "counter$0", "column": c1.j}, {"member": "counter$1", "column": c2.j}] if counting else []))
        if not counting:
            return sp
        zero = const(0, "int64")
        if self.kind in ("intersect", "except"):  # EXT E7: arith.cmpi / arith.andi
            pred = cx.col("set", "predicate", "int1")
            sp.op("map", computed=[{"computed": pred.j, "expression": and_(gt(c1.j, zero), gt(c2.j, zero) if self.kind == "intersect" else eq(c2.j, zero))}])
            sp.op("filter", semantic="all_true", columns=[pred.j])
            return sp
        rep = cx.col("set", "repeat", "index")  # EXT E7: arith.subi / arith.select
        sp.op("map", computed=[{"computed": rep.j, "expression": select(lt(sub(c1.j, c2.j), zero), zero, sub(c1.j, c2.j)) if self.kind == "except_all" else select(gt(c1.j, c2.j), c2.j, c1.j)}])
        gen = sp.raw("generate", generated=[])
        nm = sp.raw("nested_map", [sp.last], inputs=[rep.j], subops=[gen])
        sp.ops.append(nm)
        sp.last = nm["ref"]
        return sp


class GroupJoin(Node):
    """GroupJoinLowering (:2682-2950), inner behaviour: the left input creates one map entry per key (its `stored` columns are members), the
    right input looks its group up (an optional reference), gathers the stored columns, takes the left key's name for its own key
    column (renaming), applies the predicate, marks the group and reduces the aggregates into it; the map is scanned, filtered on the
    marker and the right key's name is restored.  keys = [(left column, right column)], aggs = [(fn, right column | None, result column)]"""
    def __init__(self, left, right, keys, aggs, stored=(), predicate=(), behavior="inner"):
        self.left, self.right, self.keys, self.aggs, self.stored, self.predicate = left, right, list(keys), list(aggs), list(stored), list(predicate)
        self.inner = behavior == "inner"  # outer behaviour: no marker member — every left key comes out, with default aggregates when no tuple joined it

    def avail(self): return {r.name for _, r in self.keys} | {l.name for l, _ in self.keys} | {c.name for c in self.stored} | {o.name for _, _, o in self.aggs}

    def lower(self, cx, required):
        s_st, _ = cx.state("generic_create")
        lp = self.left.lower(cx, {l.name for l, _ in self.keys} | {c.name for c in self.stored})
        ref = cx.col("lookup", "ref", "?")
        lp.op("lookup_or_insert", accesses=[lp.state_arg(s_st)], stateType="HashMap", reference=ref.j)
        members = (["gjvalmarker$0"] if self.inner else []) + ["gjval$%d" % k for k in range(len(self.stored))] + ["aggrval$%d" % i for i in range(len(self.aggs))]
        stored_m = {"gjval$%d" % k: c for k, c in enumerate(self.stored)}
        lp.op("reduce", reference=ref.j, updated=[{"member": m, "expression": stored_m[m].j if m in stored_m else member(m)} for m in members])  # stores the columns, keeps the rest
        lp.close()
        need = {r.name for _, r in self.keys} | {a.name for _, a, _ in self.aggs if a is not None}
        for e in self.predicate:
            need |= refs(e)
        rp = self.right.lower(cx, need & self.right.avail())
        opt, ref2 = cx.col("lookup", "ref", "?"), cx.col("lookup", "ref", "?")
        rp.op("lookup", accesses=[rp.state_arg(s_st)], stateType="HashMap", reference=opt.j)
        rp.op("unwrap_optional_ref", reference=ref2.j, optionalRef=opt.j)
        if self.stored:
            rp.op("gather", reference=ref2.j, mapping=[{"member": m, "column": c.j} for m, c in stored_m.items()])
        rp.op("renaming", renamed=[{"new": l.j, "old": r.j} for l, r in self.keys if l.name not in {c.name for c in self.stored}])
        selection(rp, self.predicate)
        if self.inner:
            bv = cx.col("map", "boolval", "int1")
            rp.op("map", computed=[{"computed": bv.j, "expression": const(True, "int1")}])
            rp.op("scatter", reference=ref2.j, mapping=[{"member": "gjvalmarker$0", "column": bv.j}])
        upd = []
        for i, (fn, a, _) in enumerate(self.aggs):
            m = "aggrval$%d" % i
            upd.append({"member": m, "expression": agg_body(fn, m, a.j if a is not None else None, fn in ("sum", "min", "max"))})
        rp.op("reduce", reference=ref2.j, updated=([{"member": "gjvalmarker$0", "expression": member("gjvalmarker$0")}] if self.inner else []) + [{"member": m, "expression": member(m)} for m in stored_m] + upd)
        rp.close()
        sp = Pipe(cx)
        marker = cx.col("groupjoin", "marker", "int1")
        sp.op("scan", source=True, accesses=[sp.state_arg(s_st)],
              mapping=([{"member": "gjvalmarker$0", "column": marker.j}] if self.inner else []) + [{"member": m, "column": c.j} for m, c in stored_m.items()] +
                      [{"member": "aggrval$%d" % i, "column": o.j} for i, (_, _, o) in enumerate(self.aggs)] + [{"member": "gjkeyval$%d" % i, "column": l.j} for i, (l, _) in enumerate(self.keys)])
        if self.inner:
            sp.op("filter", semantic="all_true", columns=[marker.j])
        sp.op("renaming", renamed=[{"new": r.j, "old": l.j} for l, r in self.keys])
        return sp


I64_MIN, I64_MAX = -(1 << 63), (1 << 63) - 1


class Window(Node):
    """WindowLowering (:2193-2553).  Without PARTITION BY: materialize(Buffer) → create_sorted_view → create_continuous_view [→ create_segment_tree_view]
    → scan_ref → gather → get_begin_reference → get_end_reference → [map const + offset_reference_by per bounded frame end] → rank: entries_between +
    map(+ 1); aggregates: lookup(SegmentTreeView, keys = the frame's references) + gather.  With PARTITION BY the buffer is the value of a map keyed by
    the partition columns (lookup_or_insert + a reduce that materializes into it), the map is scanned and the chain above runs inside a nested_map over
    the buffer column.  frame = (from, to) as row offsets, I64_MIN / I64_MAX = unbounded; fns = [(fn, argument column | None, result column)].
    Emitter extensions: E4 (sortBy), E7 (arith.addi of the rank), E8 (`aggregates` of create_segment_tree_view, `keys` of its lookup), E9 (`materialized` of the
    reduce that fills a partition's buffer: the region is printed as an unknown expression today)."""
    def __init__(self, child, partition_by, order_by, frame, fns):
        self.child, self.partition_by, self.order_by, self.frame, self.fns = child, list(partition_by), list(order_by), frame, list(fns)

    def avail(self): return self.child.avail() | {o.name for _, _, o in self.fns}

    def lower(self, cx, required):
        outs = {o.name for _, _, o in self.fns}
        need = (set(required) - outs) | {c.name for c in self.partition_by} | {c.name for c, _ in self.order_by} | {a.name for _, a, _ in self.fns if a is not None}
        p = self.child.lower(cx, need & self.child.avail())
        pkeys = {c.name for c in self.partition_by}
        names = sorted(need - pkeys)
        tag = cx.scope("w")
        members = {n: "%s$%s" % (n.replace("::", "_"), tag) for n in names}
        colmap = [{"member": members[n], "column": column(n, TYPE_OF.get(n, "?"))} for n in names]
        if not self.partition_by:
            s_buf, ty = cx.state("generic_create", "Buffer[...]")
            p.op("materialize", accesses=[p.state_arg(s_buf, ty)], stateType="Buffer", mapping=colmap)
            p.close()
            ep = Pipe(cx)
            return self._evaluate(cx, ep, ep.ops, ep.state_arg(s_buf, ty), members, colmap, None)
        s_map, _ = cx.state("generic_create")
        ref = cx.col("lookup", "ref", "?")
        p.op("lookup_or_insert", accesses=[p.state_arg(s_map)], stateType="HashMap", reference=ref.j)
        p.op("reduce", reference=ref.j, updated=[{"member": "buffer$0", "expression": unknown()}], materialized=colmap)  # EXT E9
        p.close()
        sp = Pipe(cx)
        buf = cx.col("window", "buffer", "?")
        sp.op("scan", source=True, accesses=[sp.state_arg(s_map)], mapping=[{"member": "keyval$%d" % i, "column": k.j} for i, k in enumerate(self.partition_by)] + [{"member": "buffer$0", "column": buf.j}])
        body = []
        outer = sp.last
        access = {"type": "nested_map_arg", "column": buf.j, "id": "pending"}
        self._evaluate(cx, sp, body, access, members, colmap, body)
        nm = sp.raw("nested_map", [outer], inputs=[buf.j], subops=body)
        access["id"] = nm["ref"] + "_0"
        sp.ops.append(nm)
        sp.last = nm["ref"]
        return sp

    def _evaluate(self, cx, p, ops, buffer_access, members, colmap, body):
        """the evaluation over one (partition's) buffer; `ops` collects the sub-operators (the step's own list, or a nested_map body)"""
        def state(kind, accesses, **fields):  # a view: inside a body an ordinary sub-operator, otherwise its own execution step
            if body is not None:
                o = p.raw(kind, accesses=accesses, **fields)
                ops.append(o)
                return node(o["ref"])
            o = cx.d.subop(kind, accesses=[arg(0)], **fields)
            step = cx.d.step([o], inputs=[("?", accesses[0]["_step"], 0)], results=[("?", o["ref"], 0)])
            return {"_step": step}

        def acc(a):  # how the evaluating pipeline names a view
            return a if body is not None else p.state_arg(a["_step"])

        if body is None:
            buffer_access = {"_step": p.inputs[buffer_access["argnr"]][1]}
        view = buffer_access
        if self.order_by:
            view = state("create_sorted_view", [view], sortBy=[{"member": members[c.name], "direction": dr} for c, dr in self.order_by])  # EXT E4
        cv = state("create_continuous_view", [view])
        frm, to = self.frame
        aggs = [(fn, a, o) for fn, a, o in self.fns if fn != "rank"]
        stv = static = None
        if aggs and frm == I64_MIN and to == I64_MAX:  # the whole partition: aggregated once by a scan of the view into a simple state (:2497-2499), looked up by every row
            upd = [{"member": "aggrVal$%d" % i, "expression": agg_body(fn, "aggrVal$%d" % i, a.j if a is not None else None, fn in ("sum", "min", "max"))} for i, (fn, a, _) in enumerate(aggs)]
            aref = cx.col("lookup", "ref", "?")
            if body is not None:
                cs = p.raw("create_simple_state")
                sc = p.raw("scan", accesses=[acc(cv)], mapping=colmap)
                lk = p.raw("lookup", [sc["ref"]], accesses=[node(cs["ref"])], stateType="SimpleState", reference=aref.j)
                rd = p.raw("reduce", [lk["ref"]], reference=aref.j, updated=upd)
                ops.extend([cs, sc, lk, rd])
                static = node(cs["ref"])
            else:
                s_cs, _ = cx.state("create_simple_state")
                ap = Pipe(cx)
                ap.op("scan", source=True, accesses=[ap.state_arg(cv["_step"])], mapping=colmap)
                ap.op("lookup", accesses=[ap.state_arg(s_cs)], stateType="SimpleState", reference=aref.j)
                ap.op("reduce", reference=aref.j, updated=upd)
                ap.close()
                static = {"_step": s_cs}
        elif aggs:
            stv = state("create_segment_tree_view", [cv], aggregates=[{"member": "aggrVal$%d" % i, "fn": fn, "source": members[a.name] if a is not None else ""} for i, (fn, a, _) in enumerate(aggs)])  # EXT E8
        chain = []

        def op(kind, source=False, **fields):
            o = p.raw(kind, [] if source or not chain else [chain[-1]["ref"]], **fields)
            chain.append(o)
            ops.append(o)
            return o
        cur, begin, end = cx.col("scan", "ref", "?"), cx.col("view", "begin", "?"), cx.col("view", "end", "?")
        op("scan_ref", source=True, accesses=[acc(cv)], stateType="ContinuousView", reference=cur.j)
        op("gather", reference=cur.j, mapping=colmap)
        op("get_begin_reference", accesses=[acc(cv)], reference=begin.j)
        op("get_end_reference", accesses=[acc(cv)], reference=end.j)

        def frame_ref(k, name):
            if k == I64_MIN:
                return begin
            if k == I64_MAX:
                return end
            if k == 0:
                return cur
            iv, nr = cx.col("map", "ival", "index"), cx.col("frame", name, "?")
            op("map", computed=[{"computed": iv.j, "expression": const(k, "index")}])
            op("offset_reference_by", reference=cur.j, offset=iv.j, newRef=nr.j)
            return nr
        fb, fe = frame_ref(frm, "from"), frame_ref(to, "to")
        for fn, a, o in self.fns:
            if fn == "rank":
                between = cx.col("window", "entries_between", "index")
                op("entries_between", leftRef=fb.j, rightRef=cur.j, between=between.j)
                op("map", computed=[{"computed": o.j, "expression": add(between.j, const(1, "index"))}])  # EXT E7
        if static is not None:
            lref = cx.col("lookup", "ref", "?")
            op("lookup", accesses=[acc(static)], stateType="SimpleState", reference=lref.j)
            op("gather", reference=lref.j, mapping=[{"member": "aggrVal$%d" % i, "column": o.j} for i, (_, _, o) in enumerate(aggs)])
        elif aggs:
            lref = cx.col("lookup", "ref", "?")
            op("lookup", accesses=[acc(stv)], stateType="SegmentTreeView", reference=lref.j, keys=[fb.j, fe.j])  # EXT E8
            op("gather", reference=lref.j, mapping=[{"member": "aggrVal$%d" % i, "column": o.j} for i, (_, _, o) in enumerate(aggs)])
        if body is None:
            p.last = chain[-1]["ref"]
        return p


def result(cx, child, outs, write=True):
    """MaterializeLowering: outs = [(result column name, C)]; writes tests/golden/subop_<name>.json (or returns the document's text)"""
    p = child.lower(cx, {c.name for _, c in outs})
    s_rt, ty = cx.state("generic_create", "ResultTable[...]")
    p.op("materialize", accesses=[p.state_arg(s_rt, ty)], stateType="ResultTable", mapping=[{"member": "%s$r" % n, "column": c.j} for n, c in outs])
    p.close()
    if not write:
        import json

        return json.dumps(cx.d.document())
    return cx.d.write()
