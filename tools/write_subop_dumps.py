#!/usr/bin/env python3
"""Hand-authored sub-operator dumps of TPC-H Q6, Q1 and Q3 in the schema of the reference's
`tools/ct/mlir-subop-to-json.cpp` (run by tools/ct/ct.py:134 on the snapshot after `subop-prepare-lowering`).
The tool cannot be built here (MLIR), so every helper below restates ONE function of it and emits the same
fields; the sub-operator sequences follow the reference's lowerings:

  BaseTableLowering          RelAlgToSubOp.cpp:103-141   get_external (+ pushed-down FilterDescriptions) → scan
  performAggregation         RelAlgToSubOp.cpp:2130-2190 create_simple_state | generic_create(map) → lookup |
                                                         lookup_or_insert → reduce; members "aggrVal$n" / "keyval$n"
  Sum/Count aggregate bodies RelAlgToSubOp.cpp:1805-2020 state + arg, state + 1, nullable-state select
  translateHJ + Specialize   RelAlgToSubOp.cpp:1097-1128 materialize(buffer) → create_hash_indexed_view → lookup →
                                                         nested_map { scan_list → gather → combine_tuple → map → filter }
  Pushdown                   Pushdown.cpp:309-411        selections folded into the base table's filters

Fields the tool does NOT emit today and a GPU backend needs (INTEGRATION.md §1b lists the one-line emitter
changes); they are marked EXT below:
  E1  db.sub is printed with the separator " + " (mlir-subop-to-json.cpp:334) — dumps here use " - "
  (E3 is withdrawn: earlier rounds modelled the thread-local access as a `get_local` sub-operator; the reference has none — a step input is
      marked thread-local on the execution step itself, SubOpToControlFlow.cpp:4367-4378 — and a GPU consumer does not need the mark)
  E4  create_sorted_view / create_heap carry no sort criteria → "sortBy": [{"member", "direction"}], "maxRows"
  E5  get_external meta has no key information → optional "primaryKey": [identifiers]
  E6  subop.combine_tuple has no case → {"subop": "combine_tuple"}

Q4 comes in the two forms the reference has for a semi join (SemiJoinLowering, RelAlgToSubOp.cpp:1340-1375): with
`reverseSides` the ORDERS are the hash-table side and matched entries get a flag member scattered to true, the orders are
then scanned back with a filter on the flag (translateNLWithMarker, :1217-1294) — tpch_q4; without it the lineitems are the
hash-table side and every probing order carries a marker state set by the first partner (anyTuple, :1296-1305) —
tpch_q4_probe_side.

Q5 is the widest shape: six tables, five hash joins chained through build buffers, one of them on a composite key.

Q12 adds the conditional aggregate: `sum(case when … then 1 else 0 end)` = a map computing the case + a plain SUM.

Q18 chains an aggregation with HAVING into the build side of a semi join, two inner joins, a five-key GROUP BY and a heap.

Writes tests/golden/subop_tpch_q{6,1,3,4,5,12,18}.json and subop_tpch_q4_probe_side.json."""
import json
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
# the last element of a document from the patched emitter (integration/mlir-subop-to-json.patch, run with --gpu-manifest): the extensions it applied.
# ldb_subop_translate refuses a document without it — the unpatched tool's " + " for db.sub and flag-less db.between would be mis-read silently
MANIFEST = {"type": "emitter_manifest", "extensions": ["E1", "E4", "E5", "E6", "E7", "E8", "E9", "E10"]}


def column(name, datatype):  # columnToJSON, mlir-subop-to-json.cpp:395-409
    return {"datatype": datatype, "type": "expression_leaf", "leaf_type": "column", "displayName": name}


def const(value, data_type):  # convertConstant, :204-228
    return {"type": "expression_leaf", "leaf_type": "constant", "data_type": data_type, "value": value}


def member(name):  # the `member` leaf of a reduce region argument, :826
    return {"type": "expression_leaf", "leaf_type": "member", "member": name}


def unknown():  # what the tool prints for an op it has no case for (db.nullable_get_val …), :344-348
    return {"type": "expression_leaf", "leaf_type": "unknown"}


def inner(strings, subs):  # innerExpression, :183-203
    return {"type": "expression_inner", "strings": strings, "subExpressions": subs}


def mul(a, b): return inner(["", " * ", ""], [a, b])
def add(a, b): return inner(["", " + ", ""], [a, b])
def sub(a, b): return inner(["", " - ", ""], [a, b])  # EXT E1
def div(a, b): return inner(["", " / ", ""], [a, b])
def cast(a): return inner(["cast(", ")"], [a])
def eq(a, b): return inner(["", "=", ""], [a, b])  # convertCmpPredicate prints no spaces, :169-182
def lt(a, b): return inner(["", "<", ""], [a, b])
def neq(a, b): return inner(["", "<>", ""], [a, b])
def or_(*xs): return inner(["("] + [" or "] * (len(xs) - 1) + [")"], list(xs))
def if_(c, a, b): return inner(["if ", " then ", " else ", ""], [c, a, b])  # scf.if with one result, :336-343
def hash_(*cols): return inner(["hash("] + [")"], [cols[0]]) if len(cols) == 1 else inner(["hash(", ")"], [inner(["pack("] + [", "] * (len(cols) - 1) + [")"], list(cols))])
def isnull(a): return inner(["", " is null"], [a])
def select(c, a, b): return inner(["", " ? ", " : ", ""], [c, a, b])


class Dump:
    def __init__(self, name):
        self.name, self.line, self.plan = name, 0, []

    def ref(self):  # getOperationReference: "<file stem>:<line>", :367-381
        self.line += 1
        return "%s:%d" % (self.name, self.line)

    def subop(self, kind, streams=(), accesses=(), **fields):  # the envelope of convertOperation, :433-445
        r = self.ref()
        node = {"ref": r, "type": "suboperator",
                "outerEdges": [{"type": "stream", "input": {"type": "node", "ref": s, "resnr": 0}, "output": {"type": "node", "ref": r}} for s in streams],
                "accesses": list(accesses), "subop": kind}
        node.update(fields)
        return node

    def step(self, subops, inputs=(), results=()):
        """execution_step, :446-500.  inputs: [(type string, producing step ref, result number)];
        results: [(type string, ref of the inner op, its result number)]"""
        r = self.ref()
        node = {"ref": r, "type": "execution_step", "outerEdges": [], "accesses": [], "subops": subops, "inputs": [], "results": [], "innerEdges": []}
        for i, (ty, src, resnr) in enumerate(inputs):
            node["inputs"].append({"type": ty})
            node["outerEdges"].append({"type": "requiredInput", "input": {"type": "node", "ref": src, "resnr": resnr}, "output": {"type": "node", "ref": r, "argnr": i}})
        for i, (ty, src, resnr) in enumerate(results):
            node["results"].append({"type": ty})
            node["innerEdges"].append({"type": "resultEdge", "input": {"type": "node", "ref": src, "resnr": resnr}, "output": {"type": "parentResult", "resnr": i}})
        if self.plan:  # ToJson::run chains the steps with "order" edges, :856-858
            node["outerEdges"].append({"type": "order", "input": {"type": "node", "ref": self.plan[-1]["ref"]}, "output": {"type": "node", "ref": r}})
        self.plan.append(node)
        return r

    def document(self):
        """the array ToJson::run prints, closed by the manifest of the patched emitter (integration/mlir-subop-to-json.patch, --gpu-manifest)"""
        return self.plan + [MANIFEST]

    def write(self):
        path = os.path.join(OUT, "subop_%s.json" % self.name)
        with open(path, "w") as f:
            json.dump(self.document(), f, indent=1)
            f.write("\n")
        return path


def arg(n): return {"type": "parentArg", "argnr": n}
def node(ref, resnr=0): return {"type": "node", "ref": ref, "resnr": resnr}


TYPES = {"l_orderkey": "int32", "l_partkey": "int32", "l_suppkey": "int32", "l_linenumber": "int32", "l_quantity": "decimal(12,2)",
         "l_extendedprice": "decimal(12,2)", "l_discount": "decimal(12,2)", "l_tax": "decimal(12,2)", "l_returnflag": "char1",
         "l_linestatus": "char1", "l_shipdate": "date", "l_commitdate": "date", "l_receiptdate": "date", "l_shipinstruct": "str",
         "l_shipmode": "str", "l_comment": "str",
         "o_orderkey": "int32", "o_custkey": "int32", "o_orderstatus": "char1", "o_totalprice": "decimal(12,2)", "o_orderdate": "date",
         "o_orderpriority": "str", "o_clerk": "str", "o_shippriority": "int32", "o_comment": "str",
         "c_custkey": "int32", "c_name": "str", "c_address": "str", "c_nationkey": "int32", "c_phone": "str", "c_acctbal": "decimal(12,2)",
         "c_mktsegment": "str", "c_comment": "str",
         "s_suppkey": "int32", "s_name": "str", "s_address": "str", "s_nationkey": "int32", "s_phone": "str", "s_acctbal": "decimal(12,2)", "s_comment": "str",
         "n_nationkey": "int32", "n_name": "str", "n_regionkey": "int32", "n_comment": "str",
         "r_regionkey": "int32", "r_name": "str", "r_comment": "str"}
TABLES = {"lineitem": [c for c in TYPES if c.startswith("l_")], "orders": [c for c in TYPES if c.startswith("o_")], "customer": [c for c in TYPES if c.startswith("c_")],
          "supplier": [c for c in TYPES if c.startswith("s_")], "nation": [c for c in TYPES if c.startswith("n_")], "region": [c for c in TYPES if c.startswith("r_")]}
PKEY = {"orders": ["o_orderkey"], "customer": ["c_custkey"], "supplier": ["s_suppkey"], "nation": ["n_nationkey"], "region": ["r_regionkey"]}


def get_external(d, table, filters):
    """one step holding subop.get_external; meta = the deserialised ExternalDatasourceProperty (:508-560)"""
    cols = TABLES[table]
    meta = {"tableName": table, "mapping": [{"memberName": "%s$0" % c, "identifier": c} for c in cols],
            "filters": [{"columnName": c, "columnId": 0, "op": op, ("values" if isinstance(v, list) else "value"): v} for c, op, v in filters]}  # IN carries `values` (:541-552)
    if table in PKEY:
        meta["primaryKey"] = PKEY[table]  # EXT E5
    op = d.subop("get_external", meta=meta)
    ty = "Table[" + "".join("%s$0:%s," % (c, TYPES[c]) for c in cols) + "]"
    return d.step([op], results=[(ty, op["ref"], 0)]), ty


def col(table, c): return column("%s::%s" % (table, c), TYPES[c])
def scan_mapping(table, cols): return [{"member": "%s$0" % c, "column": col(table, c)} for c in cols]


def q6():
    d = Dump("tpch_q6")
    t, tty = get_external(d, "lineitem", [("l_shipdate", "GTE", "1994-01-01"), ("l_shipdate", "LT", "1995-01-01"), ("l_discount", "GTE", "0.05"),
                                          ("l_discount", "LTE", "0.07"), ("l_quantity", "LT", "24")])
    tl = d.subop("create_thread_local")  # the tool's first create_thread_local case prints no resultType (:501-504)
    s_tl = d.step([tl], results=[("?", tl["ref"], 0)])
    scan = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_extendedprice", "l_discount"]))
    prod = column("map0::tmp_attr0", "decimal(24,4)")
    mp = d.subop("map", streams=[scan["ref"]], computed=[{"computed": prod, "expression": mul(col("lineitem", "l_extendedprice"), col("lineitem", "l_discount"))}])
    ref = column("lookup0::ref", "?")
    # the step's second input is the thread-local state: handleExecutionStepCPU hands the block argument the calling thread's instance
    # (step.getIsThreadLocal(), SubOpToControlFlow.cpp:4367-4378) — there is no sub-operator for that, the lookup accesses the argument
    lk = d.subop("lookup", streams=[mp["ref"]], accesses=[arg(1)], stateType="SimpleState", reference=ref)
    # SumAggrFunc::aggregate with a nullable state and a non-nullable argument (RelAlgToSubOp.cpp:1996-2002):
    # select(isnull(state), arg, nullable_get_val(state) + arg); the tool has no case for nullable_get_val
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref,
                 updated=[{"member": "aggrVal$0", "expression": select(isnull(member("aggrVal$0")), prod, add(unknown(), prod))}])
    d.step([scan, mp, lk, rd], inputs=[(tty, t, 0), ("?", s_tl, 0)])
    mg = d.subop("merge", accesses=[arg(0)], stateType="SimpleState")
    s_mg = d.step([mg], inputs=[("?", s_tl, 0)], results=[("?", mg["ref"], 0)])
    rt = d.subop("generic_create")
    rty = "ResultTable[revenue$0:nullable(decimal(24,4)),]"
    s_rt = d.step([rt], results=[(rty, rt["ref"], 0)])
    revenue = column("aggr0::tmp_attr1", "nullable(decimal(24,4))")
    sc2 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "aggrVal$0", "column": revenue}])
    mat = d.subop("materialize", streams=[sc2["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "revenue$0", "column": revenue}])
    d.step([sc2, mat], inputs=[("?", s_mg, 0), (rty, s_rt, 0)])
    return d.write()


def q1():
    d = Dump("tpch_q1")
    t, tty = get_external(d, "lineitem", [("l_shipdate", "LTE", "1998-09-02")])
    tl = d.subop("create_thread_local")
    s_tl = d.step([tl], results=[("?", tl["ref"], 0)])
    L = lambda c: col("lineitem", c)
    scan = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_discount", "l_tax"]))
    one = const(1, "decimal(12,2)")
    disc = column("map0::tmp_attr0", "decimal(24,4)")
    charge = column("map0::tmp_attr1", "decimal(36,6)")
    mp = d.subop("map", streams=[scan["ref"]], computed=[
        {"computed": disc, "expression": mul(L("l_extendedprice"), sub(one, L("l_discount")))},
        {"computed": charge, "expression": mul(mul(L("l_extendedprice"), sub(one, L("l_discount"))), add(one, L("l_tax")))}])
    ref = column("lookup0::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[mp["ref"]], accesses=[arg(1)], stateType="HashMap", reference=ref)
    # the frontend splits avg(x) into sum(x) / count(x) (sql_analyzer); CountAggrFunc over a non-nullable
    # argument and CountStarAggrFunc are both state + 1 (RelAlgToSubOp.cpp:1815-1837)
    srcs = [L("l_quantity"), L("l_extendedprice"), disc, charge, L("l_quantity"), None, L("l_extendedprice"), None, L("l_discount"), None, None]
    upd = []
    for i, s in enumerate(srcs):
        m = "aggrVal$%d" % i
        upd.append({"member": m, "expression": add(member(m), s if s is not None else const(1, "int64"))})
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=upd)
    d.step([scan, mp, lk, rd], inputs=[(tty, t, 0), ("?", s_tl, 0)])
    mg = d.subop("merge", accesses=[arg(0)], stateType="HashMap")
    s_mg = d.step([mg], inputs=[("?", s_tl, 0)], results=[("?", mg["ref"], 0)])
    buf = d.subop("generic_create")
    s_buf = d.step([buf], results=[("Buffer[...]", buf["ref"], 0)])
    A = [column("aggr0::tmp_attr%d" % i, "decimal(38,2)" if i < 4 or i in (4, 6, 8) else "int64") for i in range(11)]
    sc2 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": L("l_returnflag")}, {"member": "keyval$1", "column": L("l_linestatus")}] +
                  [{"member": "aggrVal$%d" % i, "column": A[i]} for i in range(11)])
    avgs = [column("map1::tmp_attr%d" % i, "decimal(38,8)") for i in range(3)]
    mp2 = d.subop("map", streams=[sc2["ref"]], computed=[{"computed": avgs[i], "expression": div(A[4 + 2 * i], cast(A[5 + 2 * i]))} for i in range(3)])
    outs = [("l_returnflag", L("l_returnflag")), ("l_linestatus", L("l_linestatus")), ("sum_qty", A[0]), ("sum_base_price", A[1]), ("sum_disc_price", A[2]),
            ("sum_charge", A[3]), ("avg_qty", avgs[0]), ("avg_price", avgs[1]), ("avg_disc", avgs[2]), ("count_order", A[10])]
    mat = d.subop("materialize", streams=[mp2["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "%s$1" % n, "column": c} for n, c in outs])
    d.step([sc2, mp2, mat], inputs=[("?", s_mg, 0), ("Buffer[...]", s_buf, 0)])
    sv = d.subop("create_sorted_view", accesses=[arg(0)], sortBy=[{"member": "l_returnflag$1", "direction": "asc"}, {"member": "l_linestatus$1", "direction": "asc"}])  # EXT E4
    s_sv = d.step([sv], inputs=[("Buffer[...]", s_buf, 0)], results=[("SortedView Buffer[...]", sv["ref"], 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("sorted0::%s" % n, c["datatype"])) for n, c in outs]
    sc3 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$1" % n, "column": c} for n, c in final])
    mat2 = d.subop("materialize", streams=[sc3["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$2" % n, "column": c} for n, c in final])
    d.step([sc3, mat2], inputs=[("SortedView Buffer[...]", s_sv, 0), ("ResultTable[...]", s_rt, 0)])
    return d.write()


def hash_join_probe(d, stream_ref, hiv_arg, probe_key, build_cols, n):
    """lookup into a hash-indexed view and the nested_map translateHJ builds around the matches; returns (ops, ref of the
    nested_map, whose result stream the next sub-operator consumes)"""
    lst = column("lookup%d::list" % n, "?")
    ent = column("lookup%d::entryref" % n, "?")
    lk = d.subop("lookup", streams=[stream_ref], accesses=[arg(hiv_arg)], stateType="HashIndexedView", reference=lst)
    nm_ref = None
    sl = d.subop("scan_list", accesses=[{"type": "nested_map_arg", "column": lst, "id": "pending"}], elem=ent)
    ga = d.subop("gather", streams=[sl["ref"]], reference=ent, mapping=[{"member": m, "column": c} for m, c in build_cols])
    ct = d.subop("combine_tuple", streams=[ga["ref"]])  # EXT E6
    pred = column("map_hj%d::pred" % n, "int1")
    mp = d.subop("map", streams=[ct["ref"]], computed=[{"computed": pred, "expression": eq(probe_key, build_cols[0][1])}])
    fl = d.subop("filter", streams=[mp["ref"]], semantic="all_true", columns=[pred])
    nm = d.subop("nested_map", streams=[lk["ref"]], inputs=[], subops=[sl, ga, ct, mp, fl])
    sl["accesses"][0]["id"] = nm["ref"] + "_0"
    return [lk, nm], nm["ref"]


def q3():
    d = Dump("tpch_q3")
    C = lambda c: col("customer", c)
    O = lambda c: col("orders", c)
    L = lambda c: col("lineitem", c)
    tc, tcty = get_external(d, "customer", [("c_mktsegment", "EQ", "BUILDING")])
    to, toty = get_external(d, "orders", [("o_orderdate", "LT", "1995-03-15")])
    tl_, tlty = get_external(d, "lineitem", [("l_shipdate", "GT", "1995-03-15")])
    # build side 1: customer
    b1 = d.subop("generic_create")
    s_b1 = d.step([b1], results=[("Buffer[...]", b1["ref"], 0)])
    sc = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("customer", ["c_custkey"]))
    h1 = column("hj0::hash", "index")
    m1 = d.subop("map", streams=[sc["ref"]], computed=[{"computed": h1, "expression": hash_(C("c_custkey"))}])
    mt1 = d.subop("materialize", streams=[m1["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "hash$0", "column": h1}, {"member": "c_custkey$1", "column": C("c_custkey")}])
    d.step([sc, m1, mt1], inputs=[(tcty, tc, 0), ("Buffer[...]", s_b1, 0)])
    v1 = d.subop("create_hash_indexed_view", accesses=[arg(0)])
    s_v1 = d.step([v1], inputs=[("Buffer[...]", s_b1, 0)], results=[("?", v1["ref"], 0)])
    # orders probe customer, the matches are build side 2
    b2 = d.subop("generic_create")
    s_b2 = d.step([b2], results=[("Buffer[...]", b2["ref"], 0)])
    so = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("orders", ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"]))
    h2 = column("hj1::hash", "index")
    m2 = d.subop("map", streams=[so["ref"]], computed=[{"computed": h2, "expression": hash_(O("o_custkey"))}])
    probe1, after1 = hash_join_probe(d, m2["ref"], 1, O("o_custkey"), [("c_custkey$1", C("c_custkey"))], 0)
    h3 = column("hj2::hash", "index")
    m3 = d.subop("map", streams=[after1], computed=[{"computed": h3, "expression": hash_(O("o_orderkey"))}])
    mt2 = d.subop("materialize", streams=[m3["ref"]], accesses=[arg(2)], stateType="Buffer",
                  mapping=[{"member": "hash$2", "column": h3}, {"member": "o_orderkey$3", "column": O("o_orderkey")}, {"member": "o_orderdate$3", "column": O("o_orderdate")},
                           {"member": "o_shippriority$3", "column": O("o_shippriority")}])
    d.step([so, m2] + probe1 + [m3, mt2], inputs=[(toty, to, 0), ("?", s_v1, 0), ("Buffer[...]", s_b2, 0)])
    v2 = d.subop("create_hash_indexed_view", accesses=[arg(0)])
    s_v2 = d.step([v2], inputs=[("Buffer[...]", s_b2, 0)], results=[("?", v2["ref"], 0)])
    # lineitem probes the joined orders and aggregates
    hm = d.subop("generic_create")
    s_hm = d.step([hm], results=[("?", hm["ref"], 0)])
    sl = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_extendedprice", "l_discount"]))
    h4 = column("hj3::hash", "index")
    m4 = d.subop("map", streams=[sl["ref"]], computed=[{"computed": h4, "expression": hash_(L("l_orderkey"))}])
    probe2, after2 = hash_join_probe(d, m4["ref"], 1, L("l_orderkey"),
                                     [("o_orderkey$3", O("o_orderkey")), ("o_orderdate$3", O("o_orderdate")), ("o_shippriority$3", O("o_shippriority"))], 1)
    rev_in = column("map0::tmp_attr0", "decimal(24,4)")
    m5 = d.subop("map", streams=[after2], computed=[{"computed": rev_in, "expression": mul(L("l_extendedprice"), sub(const(1, "decimal(12,2)"), L("l_discount")))}])
    ref = column("lookup2::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[m5["ref"]], accesses=[arg(2)], stateType="HashMap", reference=ref)
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=[{"member": "aggrVal$0", "expression": add(member("aggrVal$0"), rev_in)}])
    d.step([sl, m4] + probe2 + [m5, lk, rd], inputs=[(tlty, tl_, 0), ("?", s_v2, 0), ("?", s_hm, 0)])
    # top 10 by revenue desc, o_orderdate
    revenue = column("aggr0::tmp_attr1", "decimal(38,4)")
    hp = d.subop("create_heap", maxRows=10, sortBy=[{"member": "revenue$4", "direction": "desc"}, {"member": "o_orderdate$4", "direction": "asc"}])  # EXT E4
    s_hp = d.step([hp], results=[("?", hp["ref"], 0)])
    sg = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": L("l_orderkey")}, {"member": "keyval$1", "column": O("o_orderdate")},
                                                      {"member": "keyval$2", "column": O("o_shippriority")}, {"member": "aggrVal$0", "column": revenue}])
    outs = [("l_orderkey", L("l_orderkey")), ("revenue", revenue), ("o_orderdate", O("o_orderdate")), ("o_shippriority", O("o_shippriority"))]
    mh = d.subop("materialize", streams=[sg["ref"]], accesses=[arg(1)], stateType="Heap", mapping=[{"member": "%s$4" % n, "column": c} for n, c in outs])
    d.step([sg, mh], inputs=[("?", s_hm, 0), ("?", s_hp, 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("heap0::%s" % n, c["datatype"])) for n, c in outs]
    sh = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$4" % n, "column": c} for n, c in final])
    mr = d.subop("materialize", streams=[sh["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$5" % n, "column": c} for n, c in final])
    d.step([sh, mr], inputs=[("?", s_hp, 0), ("ResultTable[...]", s_rt, 0)])
    return d.write()


def q4_tail(d, stream_ref, step_subops, step_inputs, O):
    """GROUP BY o_orderpriority COUNT(*) ORDER BY o_orderpriority over `stream_ref`; appends to the given pipeline step"""
    hm = d.subop("generic_create")
    s_hm = d.step([hm], results=[("?", hm["ref"], 0)])
    ref = column("lookup9::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[stream_ref], accesses=[arg(len(step_inputs))], stateType="HashMap", reference=ref)
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=[{"member": "aggrVal$0", "expression": add(member("aggrVal$0"), const(1, "int64"))}])
    d.step(step_subops + [lk, rd], inputs=step_inputs + [("?", s_hm, 0)])
    buf = d.subop("generic_create")
    s_buf = d.step([buf], results=[("Buffer[...]", buf["ref"], 0)])
    cnt = column("aggr0::tmp_attr0", "int64")
    sc = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": O("o_orderpriority")}, {"member": "aggrVal$0", "column": cnt}])
    outs = [("o_orderpriority", O("o_orderpriority")), ("order_count", cnt)]
    mat = d.subop("materialize", streams=[sc["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "%s$7" % n, "column": c} for n, c in outs])
    d.step([sc, mat], inputs=[("?", s_hm, 0), ("Buffer[...]", s_buf, 0)])
    sv = d.subop("create_sorted_view", accesses=[arg(0)], sortBy=[{"member": "o_orderpriority$7", "direction": "asc"}])  # EXT E4
    s_sv = d.step([sv], inputs=[("Buffer[...]", s_buf, 0)], results=[("SortedView Buffer[...]", sv["ref"], 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("sorted0::%s" % n, c["datatype"])) for n, c in outs]
    s3 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$7" % n, "column": c} for n, c in final])
    m3 = d.subop("materialize", streams=[s3["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$8" % n, "column": c} for n, c in final])
    d.step([s3, m3], inputs=[("SortedView Buffer[...]", s_sv, 0), ("ResultTable[...]", s_rt, 0)])


ORDERS_Q4 = [("o_orderdate", "GTE", "1993-07-01"), ("o_orderdate", "LT", "1993-10-01")]


def q4():
    """reverseSides: orders are the build side, a flag member marks the orders that found a late lineitem"""
    d = Dump("tpch_q4")
    O = lambda c: col("orders", c)
    L = lambda c: col("lineitem", c)
    to, toty = get_external(d, "orders", ORDERS_Q4)
    tl_, tlty = get_external(d, "lineitem", [])
    b1 = d.subop("generic_create")
    s_b1 = d.step([b1], results=[("Buffer[...]", b1["ref"], 0)])
    so = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("orders", ["o_orderkey", "o_orderpriority"]))
    h1 = column("hj0::hash", "index")
    f0 = column("marker0::init", "int1")
    m1 = d.subop("map", streams=[so["ref"]], computed=[{"computed": h1, "expression": hash_(O("o_orderkey"))}, {"computed": f0, "expression": const(False, "int1")}])
    mt1 = d.subop("materialize", streams=[m1["ref"]], accesses=[arg(1)], stateType="Buffer",
                  mapping=[{"member": "hash$0", "column": h1}, {"member": "o_orderkey$1", "column": O("o_orderkey")}, {"member": "o_orderpriority$1", "column": O("o_orderpriority")},
                           {"member": "flag$2", "column": f0}])
    d.step([so, m1, mt1], inputs=[(toty, to, 0), ("Buffer[...]", s_b1, 0)])
    v1 = d.subop("create_hash_indexed_view", accesses=[arg(0)])
    s_v1 = d.step([v1], inputs=[("Buffer[...]", s_b1, 0)], results=[("?", v1["ref"], 0)])
    # lineitem (l_commitdate < l_receiptdate is column-vs-column: not pushed down) probes and flags
    sl = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_commitdate", "l_receiptdate"]))
    late = column("map0::pred", "int1")
    mp = d.subop("map", streams=[sl["ref"]], computed=[{"computed": late, "expression": lt(L("l_commitdate"), L("l_receiptdate"))}])
    fl = d.subop("filter", streams=[mp["ref"]], semantic="all_true", columns=[late])
    h2 = column("hj1::hash", "index")
    m2 = d.subop("map", streams=[fl["ref"]], computed=[{"computed": h2, "expression": hash_(L("l_orderkey"))}])
    lst, ent = column("lookup0::list", "?"), column("lookup0::entryref", "?")
    lk = d.subop("lookup", streams=[m2["ref"]], accesses=[arg(1)], stateType="HashIndexedView", reference=lst)
    sli = d.subop("scan_list", accesses=[{"type": "nested_map_arg", "column": lst, "id": "pending"}], elem=ent)
    ga = d.subop("gather", streams=[sli["ref"]], reference=ent, mapping=[{"member": "o_orderkey$1", "column": O("o_orderkey")}])
    ct = d.subop("combine_tuple", streams=[ga["ref"]])
    pred = column("map_hj0::pred", "int1")
    mq = d.subop("map", streams=[ct["ref"]], computed=[{"computed": pred, "expression": eq(L("l_orderkey"), O("o_orderkey"))}])
    fq = d.subop("filter", streams=[mq["ref"]], semantic="all_true", columns=[pred])
    mark = column("marker1::marker", "int1")
    mb = d.subop("map", streams=[fq["ref"]], computed=[{"computed": mark, "expression": const(True, "int1")}])
    sca = d.subop("scatter", streams=[mb["ref"]], reference=ent, mapping=[{"member": "flag$2", "column": mark}])
    nm = d.subop("nested_map", streams=[lk["ref"]], inputs=[], subops=[sli, ga, ct, mq, fq, mb, sca])
    sli["accesses"][0]["id"] = nm["ref"] + "_0"
    d.step([sl, mp, fl, m2, lk, nm], inputs=[(tlty, tl_, 0), ("?", s_v1, 0)])
    # the flagged orders
    flag = column("materialized::marker", "int1")
    sb = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "o_orderpriority$1", "column": O("o_orderpriority")}, {"member": "flag$2", "column": flag}])
    ff = d.subop("filter", streams=[sb["ref"]], semantic="all_true", columns=[flag])
    q4_tail(d, ff["ref"], [sb, ff], [("Buffer[...]", s_b1, 0)], O)
    return d.write()


def q4_probe_side():
    """no reverseSides: the late lineitems are the build side, every probing order carries a marker state (anyTuple)"""
    d = Dump("tpch_q4_probe_side")
    O = lambda c: col("orders", c)
    L = lambda c: col("lineitem", c)
    to, toty = get_external(d, "orders", ORDERS_Q4)
    tl_, tlty = get_external(d, "lineitem", [])
    b1 = d.subop("generic_create")
    s_b1 = d.step([b1], results=[("Buffer[...]", b1["ref"], 0)])
    sl = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_commitdate", "l_receiptdate"]))
    late = column("map0::pred", "int1")
    mp = d.subop("map", streams=[sl["ref"]], computed=[{"computed": late, "expression": lt(L("l_commitdate"), L("l_receiptdate"))}])
    fl = d.subop("filter", streams=[mp["ref"]], semantic="all_true", columns=[late])
    h1 = column("hj0::hash", "index")
    m1 = d.subop("map", streams=[fl["ref"]], computed=[{"computed": h1, "expression": hash_(L("l_orderkey"))}])
    mt1 = d.subop("materialize", streams=[m1["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "hash$0", "column": h1}, {"member": "l_orderkey$1", "column": L("l_orderkey")}])
    d.step([sl, mp, fl, m1, mt1], inputs=[(tlty, tl_, 0), ("Buffer[...]", s_b1, 0)])
    v1 = d.subop("create_hash_indexed_view", accesses=[arg(0)])
    s_v1 = d.step([v1], inputs=[("Buffer[...]", s_b1, 0)], results=[("?", v1["ref"], 0)])
    so = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("orders", ["o_orderkey", "o_orderpriority"]))
    h2 = column("hj1::hash", "index")
    m2 = d.subop("map", streams=[so["ref"]], computed=[{"computed": h2, "expression": hash_(O("o_orderkey"))}])
    lst, ent = column("lookup0::list", "?"), column("lookup0::entryref", "?")
    lk = d.subop("lookup", streams=[m2["ref"]], accesses=[arg(1)], stateType="HashIndexedView", reference=lst)
    sli = d.subop("scan_list", accesses=[{"type": "nested_map_arg", "column": lst, "id": "pending"}], elem=ent)
    ga = d.subop("gather", streams=[sli["ref"]], reference=ent, mapping=[{"member": "l_orderkey$1", "column": L("l_orderkey")}])
    ct = d.subop("combine_tuple", streams=[ga["ref"]])
    pred = column("map_hj0::pred", "int1")
    mq = d.subop("map", streams=[ct["ref"]], computed=[{"computed": pred, "expression": eq(O("o_orderkey"), L("l_orderkey"))}])
    fq = d.subop("filter", streams=[mq["ref"]], semantic="all_true", columns=[pred])
    ms = d.subop("create_simple_state")  # createMarkerState: <[marker$0 : i1]> initial false (test/lit/RelAlg/lowering.mlir:99-110)
    bv = column("map_u_1::boolval", "int1")
    mb = d.subop("map", streams=[fq["ref"]], computed=[{"computed": bv, "expression": const(True, "int1")}])
    mref = column("lookup1::ref", "?")
    ml = d.subop("lookup", streams=[mb["ref"]], accesses=[node(ms["ref"])], stateType="SimpleState", reference=mref)
    sca = d.subop("scatter", streams=[ml["ref"]], reference=mref, mapping=[{"member": "marker$0", "column": bv}])
    mk = column("marker::marker", "int1")
    sm = d.subop("scan", accesses=[node(ms["ref"])], mapping=[{"member": "marker$0", "column": mk}])
    fm = d.subop("filter", streams=[sm["ref"]], semantic="all_true", columns=[mk])
    nm = d.subop("nested_map", streams=[lk["ref"]], inputs=[], subops=[sli, ga, ct, mq, fq, ms, mb, ml, sca, sm, fm])
    sli["accesses"][0]["id"] = nm["ref"] + "_0"
    q4_tail(d, nm["ref"], [so, m2, lk, nm], [(toty, to, 0), ("?", s_v1, 0)], O)
    return d.write()


def and_(*xs): return inner([""] + [" and "] * (len(xs) - 1) + [""], list(xs))


def hash_join_probe_multi(d, stream_ref, hiv_arg, pairs, extra_cols, n):
    """hash_join_probe for a composite key: `pairs` = [(probe column, (build member, build column))]; the equality is one
    db.and over the pairs (createVerifyEqFnForTuple, RelAlgToSubOp.cpp:1067-1095)"""
    lst = column("lookup%d::list" % n, "?")
    ent = column("lookup%d::entryref" % n, "?")
    lk = d.subop("lookup", streams=[stream_ref], accesses=[arg(hiv_arg)], stateType="HashIndexedView", reference=lst)
    sl = d.subop("scan_list", accesses=[{"type": "nested_map_arg", "column": lst, "id": "pending"}], elem=ent)
    ga = d.subop("gather", streams=[sl["ref"]], reference=ent, mapping=[{"member": m, "column": c} for _, (m, c) in pairs] + [{"member": m, "column": c} for m, c in extra_cols])
    ct = d.subop("combine_tuple", streams=[ga["ref"]])
    pred = column("map_hj%d::pred" % n, "int1")
    cond = and_(*[eq(p, b[1]) for p, b in pairs]) if len(pairs) > 1 else eq(pairs[0][0], pairs[0][1][1])
    mp = d.subop("map", streams=[ct["ref"]], computed=[{"computed": pred, "expression": cond}])
    fl = d.subop("filter", streams=[mp["ref"]], semantic="all_true", columns=[pred])
    nm = d.subop("nested_map", streams=[lk["ref"]], inputs=[], subops=[sl, ga, ct, mp, fl])
    sl["accesses"][0]["id"] = nm["ref"] + "_0"
    return [lk, nm], nm["ref"]


def build_side(d, pipeline, last_ref, step_inputs, key_cols, payload, n):
    """map hash(keys) → materialize into a fresh buffer → create_hash_indexed_view; returns the step ref of the view"""
    buf = d.subop("generic_create")
    s_buf = d.step([buf], results=[("Buffer[...]", buf["ref"], 0)])
    h = column("hj_b%d::hash" % n, "index")
    m = d.subop("map", streams=[last_ref], computed=[{"computed": h, "expression": hash_(*key_cols)}])
    mt = d.subop("materialize", streams=[m["ref"]], accesses=[arg(len(step_inputs))], stateType="Buffer",
                 mapping=[{"member": "hash$b%d" % n, "column": h}] + [{"member": mem, "column": c} for mem, c in payload])
    d.step(pipeline + [m, mt], inputs=step_inputs + [("Buffer[...]", s_buf, 0)])
    v = d.subop("create_hash_indexed_view", accesses=[arg(0)])
    return d.step([v], inputs=[("Buffer[...]", s_buf, 0)], results=[("?", v["ref"], 0)])


def q5():
    """six tables, five hash joins (one on a composite key), GROUP BY a string column, ORDER BY the aggregate"""
    d = Dump("tpch_q5")
    T = lambda t: (lambda c: col(t, c))
    C, O, L, S, N, R = T("customer"), T("orders"), T("lineitem"), T("supplier"), T("nation"), T("region")
    tr, trty = get_external(d, "region", [("r_name", "EQ", "ASIA")])
    tn, tnty = get_external(d, "nation", [])
    tc, tcty = get_external(d, "customer", [])
    to, toty = get_external(d, "orders", [("o_orderdate", "GTE", "1994-01-01"), ("o_orderdate", "LT", "1995-01-01")])
    tl_, tlty = get_external(d, "lineitem", [])
    ts, tsty = get_external(d, "supplier", [])
    # region → view on r_regionkey
    sr = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("region", ["r_regionkey"]))
    v_r = build_side(d, [sr], sr["ref"], [(trty, tr, 0)], [R("r_regionkey")], [("r_regionkey$b0", R("r_regionkey"))], 0)
    # nation ⋈ region → view on n_nationkey
    sn = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("nation", ["n_nationkey", "n_name", "n_regionkey"]))
    hn = column("hj_p0::hash", "index")
    mn = d.subop("map", streams=[sn["ref"]], computed=[{"computed": hn, "expression": hash_(N("n_regionkey"))}])
    p0, a0 = hash_join_probe_multi(d, mn["ref"], 1, [(N("n_regionkey"), ("r_regionkey$b0", R("r_regionkey")))], [], 10)
    v_n = build_side(d, [sn, mn] + p0, a0, [(tnty, tn, 0), ("?", v_r, 0)], [N("n_nationkey")], [("n_nationkey$b1", N("n_nationkey")), ("n_name$b1", N("n_name"))], 1)
    # customer ⋈ nation → view on c_custkey
    sc = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("customer", ["c_custkey", "c_nationkey"]))
    hc = column("hj_p1::hash", "index")
    mc = d.subop("map", streams=[sc["ref"]], computed=[{"computed": hc, "expression": hash_(C("c_nationkey"))}])
    p1, a1 = hash_join_probe_multi(d, mc["ref"], 1, [(C("c_nationkey"), ("n_nationkey$b1", N("n_nationkey")))], [("n_name$b1", N("n_name"))], 11)
    v_c = build_side(d, [sc, mc] + p1, a1, [(tcty, tc, 0), ("?", v_n, 0)], [C("c_custkey")],
                     [("c_custkey$b2", C("c_custkey")), ("c_nationkey$b2", C("c_nationkey")), ("n_name$b2", N("n_name"))], 2)
    # orders ⋈ customer → view on o_orderkey
    so = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("orders", ["o_orderkey", "o_custkey"]))
    ho = column("hj_p2::hash", "index")
    mo = d.subop("map", streams=[so["ref"]], computed=[{"computed": ho, "expression": hash_(O("o_custkey"))}])
    p2, a2 = hash_join_probe_multi(d, mo["ref"], 1, [(O("o_custkey"), ("c_custkey$b2", C("c_custkey")))], [("c_nationkey$b2", C("c_nationkey")), ("n_name$b2", N("n_name"))], 12)
    v_o = build_side(d, [so, mo] + p2, a2, [(toty, to, 0), ("?", v_c, 0)], [O("o_orderkey")],
                     [("o_orderkey$b3", O("o_orderkey")), ("c_nationkey$b3", C("c_nationkey")), ("n_name$b3", N("n_name"))], 3)
    # supplier → view on (s_suppkey, s_nationkey)
    ss = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("supplier", ["s_suppkey", "s_nationkey"]))
    v_s = build_side(d, [ss], ss["ref"], [(tsty, ts, 0)], [S("s_suppkey"), S("s_nationkey")], [("s_suppkey$b4", S("s_suppkey")), ("s_nationkey$b4", S("s_nationkey"))], 4)
    # lineitem ⋈ orders ⋈ supplier, aggregate per nation name
    hm = d.subop("generic_create")
    s_hm = d.step([hm], results=[("?", hm["ref"], 0)])
    sl = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]))
    hl = column("hj_p3::hash", "index")
    ml = d.subop("map", streams=[sl["ref"]], computed=[{"computed": hl, "expression": hash_(L("l_orderkey"))}])
    p3, a3 = hash_join_probe_multi(d, ml["ref"], 1, [(L("l_orderkey"), ("o_orderkey$b3", O("o_orderkey")))], [("c_nationkey$b3", C("c_nationkey")), ("n_name$b3", N("n_name"))], 13)
    hl2 = column("hj_p4::hash", "index")
    ml2 = d.subop("map", streams=[a3], computed=[{"computed": hl2, "expression": hash_(L("l_suppkey"), C("c_nationkey"))}])
    p4, a4 = hash_join_probe_multi(d, ml2["ref"], 2, [(L("l_suppkey"), ("s_suppkey$b4", S("s_suppkey"))), (C("c_nationkey"), ("s_nationkey$b4", S("s_nationkey")))], [], 14)
    rev_in = column("map0::tmp_attr0", "decimal(24,4)")
    m5 = d.subop("map", streams=[a4], computed=[{"computed": rev_in, "expression": mul(L("l_extendedprice"), sub(const(1, "decimal(12,2)"), L("l_discount")))}])
    ref = column("lookup20::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[m5["ref"]], accesses=[arg(3)], stateType="HashMap", reference=ref)
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=[{"member": "aggrVal$0", "expression": add(member("aggrVal$0"), rev_in)}])
    d.step([sl, ml] + p3 + [ml2] + p4 + [m5, lk, rd], inputs=[(tlty, tl_, 0), ("?", v_o, 0), ("?", v_s, 0), ("?", s_hm, 0)])
    # ORDER BY revenue DESC
    buf = d.subop("generic_create")
    s_buf = d.step([buf], results=[("Buffer[...]", buf["ref"], 0)])
    revenue = column("aggr0::tmp_attr0", "decimal(38,4)")
    sg = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": N("n_name")}, {"member": "aggrVal$0", "column": revenue}])
    outs = [("n_name", N("n_name")), ("revenue", revenue)]
    mat = d.subop("materialize", streams=[sg["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "%s$7" % n, "column": c} for n, c in outs])
    d.step([sg, mat], inputs=[("?", s_hm, 0), ("Buffer[...]", s_buf, 0)])
    sv = d.subop("create_sorted_view", accesses=[arg(0)], sortBy=[{"member": "revenue$7", "direction": "desc"}])  # EXT E4
    s_sv = d.step([sv], inputs=[("Buffer[...]", s_buf, 0)], results=[("SortedView Buffer[...]", sv["ref"], 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("sorted0::%s" % n, c["datatype"])) for n, c in outs]
    s3 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$7" % n, "column": c} for n, c in final])
    m3 = d.subop("materialize", streams=[s3["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$8" % n, "column": c} for n, c in final])
    d.step([s3, m3], inputs=[("SortedView Buffer[...]", s_sv, 0), ("ResultTable[...]", s_rt, 0)])
    return d.write()


def q12():
    """IN and range restrictions pushed into the scan, two column-vs-column filters, a non-unique build side, and the two
    conditional counts: the frontend lowers `sum(case when … then 1 else 0 end)` to a map computing the case (scf.if) and a
    plain SUM over it"""
    d = Dump("tpch_q12")
    O = lambda c: col("orders", c)
    L = lambda c: col("lineitem", c)
    tl_, tlty = get_external(d, "lineitem", [("l_shipmode", "IN", ["MAIL", "SHIP"]), ("l_receiptdate", "GTE", "1994-01-01"), ("l_receiptdate", "LT", "1995-01-01")])
    to, toty = get_external(d, "orders", [])
    sl = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_shipmode", "l_shipdate", "l_commitdate", "l_receiptdate"]))
    p1, p2 = column("map0::pred", "int1"), column("map1::pred", "int1")
    m1 = d.subop("map", streams=[sl["ref"]], computed=[{"computed": p1, "expression": lt(L("l_commitdate"), L("l_receiptdate"))}])
    f1 = d.subop("filter", streams=[m1["ref"]], semantic="all_true", columns=[p1])
    m2 = d.subop("map", streams=[f1["ref"]], computed=[{"computed": p2, "expression": lt(L("l_shipdate"), L("l_commitdate"))}])
    f2 = d.subop("filter", streams=[m2["ref"]], semantic="all_true", columns=[p2])
    v_l = build_side(d, [sl, m1, f1, m2, f2], f2["ref"], [(tlty, tl_, 0)], [L("l_orderkey")], [("l_orderkey$b0", L("l_orderkey")), ("l_shipmode$b0", L("l_shipmode"))], 0)
    hm = d.subop("generic_create")
    s_hm = d.step([hm], results=[("?", hm["ref"], 0)])
    so = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("orders", ["o_orderkey", "o_orderpriority"]))
    ho = column("hj_p0::hash", "index")
    mo = d.subop("map", streams=[so["ref"]], computed=[{"computed": ho, "expression": hash_(O("o_orderkey"))}])
    pr, after = hash_join_probe_multi(d, mo["ref"], 1, [(O("o_orderkey"), ("l_orderkey$b0", L("l_orderkey")))], [("l_shipmode$b0", L("l_shipmode"))], 10)
    hi, lo = column("map2::tmp_attr0", "int32"), column("map2::tmp_attr1", "int32")
    urgent, high = const("1-URGENT", "str"), const("2-HIGH", "str")
    one, zero = const(1, "int32"), const(0, "int32")
    mc = d.subop("map", streams=[after], computed=[
        {"computed": hi, "expression": if_(or_(eq(O("o_orderpriority"), urgent), eq(O("o_orderpriority"), high)), one, zero)},
        {"computed": lo, "expression": if_(and_(neq(O("o_orderpriority"), urgent), neq(O("o_orderpriority"), high)), one, zero)}])
    ref = column("lookup20::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[mc["ref"]], accesses=[arg(2)], stateType="HashMap", reference=ref)
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=[{"member": "aggrVal$0", "expression": add(member("aggrVal$0"), hi)},
                                                                        {"member": "aggrVal$1", "expression": add(member("aggrVal$1"), lo)}])
    d.step([so, mo] + pr + [mc, lk, rd], inputs=[(toty, to, 0), ("?", v_l, 0), ("?", s_hm, 0)])
    buf = d.subop("generic_create")
    s_buf = d.step([buf], results=[("Buffer[...]", buf["ref"], 0)])
    a0, a1 = column("aggr0::tmp_attr0", "int32"), column("aggr0::tmp_attr1", "int32")
    sg = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": L("l_shipmode")}, {"member": "aggrVal$0", "column": a0}, {"member": "aggrVal$1", "column": a1}])
    outs = [("l_shipmode", L("l_shipmode")), ("high_line_count", a0), ("low_line_count", a1)]
    mat = d.subop("materialize", streams=[sg["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "%s$7" % n, "column": c} for n, c in outs])
    d.step([sg, mat], inputs=[("?", s_hm, 0), ("Buffer[...]", s_buf, 0)])
    sv = d.subop("create_sorted_view", accesses=[arg(0)], sortBy=[{"member": "l_shipmode$7", "direction": "asc"}])  # EXT E4
    s_sv = d.step([sv], inputs=[("Buffer[...]", s_buf, 0)], results=[("SortedView Buffer[...]", sv["ref"], 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("sorted0::%s" % n, c["datatype"])) for n, c in outs]
    s3 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$7" % n, "column": c} for n, c in final])
    m3 = d.subop("materialize", streams=[s3["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$8" % n, "column": c} for n, c in final])
    d.step([s3, m3], inputs=[("SortedView Buffer[...]", s_sv, 0), ("ResultTable[...]", s_rt, 0)])
    return d.write()


def hash_semi_probe(d, stream_ref, hiv_arg, probe_key, build, n):
    """lookup + nested_map whose body ends in anyTuple's marker idiom (RelAlgToSubOp.cpp:1296-1305): the probe row survives if
    any entry of its bucket satisfies the equality"""
    lst, ent = column("lookup%d::list" % n, "?"), column("lookup%d::entryref" % n, "?")
    lk = d.subop("lookup", streams=[stream_ref], accesses=[arg(hiv_arg)], stateType="HashIndexedView", reference=lst)
    sli = d.subop("scan_list", accesses=[{"type": "nested_map_arg", "column": lst, "id": "pending"}], elem=ent)
    ga = d.subop("gather", streams=[sli["ref"]], reference=ent, mapping=[{"member": build[0], "column": build[1]}])
    ct = d.subop("combine_tuple", streams=[ga["ref"]])
    pred = column("map_hj%d::pred" % n, "int1")
    mq = d.subop("map", streams=[ct["ref"]], computed=[{"computed": pred, "expression": eq(probe_key, build[1])}])
    fq = d.subop("filter", streams=[mq["ref"]], semantic="all_true", columns=[pred])
    ms = d.subop("create_simple_state")
    bv = column("map_u_%d::boolval" % n, "int1")
    mb = d.subop("map", streams=[fq["ref"]], computed=[{"computed": bv, "expression": const(True, "int1")}])
    mref = column("lookup%d::ref" % (n + 100), "?")
    ml = d.subop("lookup", streams=[mb["ref"]], accesses=[node(ms["ref"])], stateType="SimpleState", reference=mref)
    sca = d.subop("scatter", streams=[ml["ref"]], reference=mref, mapping=[{"member": "marker$%d" % n, "column": bv}])
    mk = column("marker%d::marker" % n, "int1")
    sm = d.subop("scan", accesses=[node(ms["ref"])], mapping=[{"member": "marker$%d" % n, "column": mk}])
    fm = d.subop("filter", streams=[sm["ref"]], semantic="all_true", columns=[mk])
    nm = d.subop("nested_map", streams=[lk["ref"]], inputs=[], subops=[sli, ga, ct, mq, fq, ms, mb, ml, sca, sm, fm])
    sli["accesses"][0]["id"] = nm["ref"] + "_0"
    return [lk, nm], nm["ref"]


def q18():
    """IN (subquery with GROUP BY … HAVING) = aggregation → filter on the aggregate → the build side of a semi join; then two
    inner joins, a five-key GROUP BY with a string key and a top-100 heap"""
    d = Dump("tpch_q18")
    C, O, L = (lambda c: col("customer", c)), (lambda c: col("orders", c)), (lambda c: col("lineitem", c))
    tl_, tlty = get_external(d, "lineitem", [])
    to, toty = get_external(d, "orders", [])
    tc, tcty = get_external(d, "customer", [])
    # subquery: sum(l_quantity) per order
    hm = d.subop("generic_create")
    s_hm = d.step([hm], results=[("?", hm["ref"], 0)])
    s1 = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_quantity"]))
    ref = column("lookup0::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[s1["ref"]], accesses=[arg(1)], stateType="HashMap", reference=ref)
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=[{"member": "aggrVal$0", "expression": add(member("aggrVal$0"), L("l_quantity"))}])
    d.step([s1, lk, rd], inputs=[(tlty, tl_, 0), ("?", s_hm, 0)])
    # HAVING sum > 300 → build side of the semi join (the key column is re-defined by the scan of the map)
    sq = column("aggr0::tmp_attr0", "decimal(38,2)")
    sg = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": L("l_orderkey")}, {"member": "aggrVal$0", "column": sq}])
    hp = column("map0::pred", "int1")
    mh = d.subop("map", streams=[sg["ref"]], computed=[{"computed": hp, "expression": inner(["", ">", ""], [sq, const("300", "decimal(38,2)")])}])
    fh = d.subop("filter", streams=[mh["ref"]], semantic="all_true", columns=[hp])
    v_k = build_side(d, [sg, mh, fh], fh["ref"], [("?", s_hm, 0)], [L("l_orderkey")], [("l_orderkey$b0", L("l_orderkey"))], 0)
    # orders semi join the big orders → build on o_custkey
    so = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("orders", ["o_orderkey", "o_custkey", "o_orderdate", "o_totalprice"]))
    ho = column("hj_p0::hash", "index")
    mo = d.subop("map", streams=[so["ref"]], computed=[{"computed": ho, "expression": hash_(O("o_orderkey"))}])
    sp, after = hash_semi_probe(d, mo["ref"], 1, O("o_orderkey"), ("l_orderkey$b0", L("l_orderkey")), 10)
    v_o = build_side(d, [so, mo] + sp, after, [(toty, to, 0), ("?", v_k, 0)], [O("o_custkey")],
                     [("o_custkey$b1", O("o_custkey")), ("o_orderkey$b1", O("o_orderkey")), ("o_orderdate$b1", O("o_orderdate")), ("o_totalprice$b1", O("o_totalprice"))], 1)
    # customer ⋈ those orders → build on o_orderkey
    sc = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("customer", ["c_custkey", "c_name"]))
    hc = column("hj_p1::hash", "index")
    mc = d.subop("map", streams=[sc["ref"]], computed=[{"computed": hc, "expression": hash_(C("c_custkey"))}])
    p1, a1 = hash_join_probe_multi(d, mc["ref"], 1, [(C("c_custkey"), ("o_custkey$b1", O("o_custkey")))],
                                   [("o_orderkey$b1", O("o_orderkey")), ("o_orderdate$b1", O("o_orderdate")), ("o_totalprice$b1", O("o_totalprice"))], 11)
    v_co = build_side(d, [sc, mc] + p1, a1, [(tcty, tc, 0), ("?", v_o, 0)], [O("o_orderkey")],
                      [("o_orderkey$b2", O("o_orderkey")), ("c_name$b2", C("c_name")), ("c_custkey$b2", C("c_custkey")), ("o_orderdate$b2", O("o_orderdate")), ("o_totalprice$b2", O("o_totalprice"))], 2)
    # lineitem ⋈, outer aggregation
    hm2 = d.subop("generic_create")
    s_hm2 = d.step([hm2], results=[("?", hm2["ref"], 0)])
    s2 = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("lineitem", ["l_orderkey", "l_quantity"]))
    hl = column("hj_p2::hash", "index")
    ml = d.subop("map", streams=[s2["ref"]], computed=[{"computed": hl, "expression": hash_(L("l_orderkey"))}])
    p2, a2 = hash_join_probe_multi(d, ml["ref"], 1, [(L("l_orderkey"), ("o_orderkey$b2", O("o_orderkey")))],
                                   [("c_name$b2", C("c_name")), ("c_custkey$b2", C("c_custkey")), ("o_orderdate$b2", O("o_orderdate")), ("o_totalprice$b2", O("o_totalprice"))], 12)
    ref2 = column("lookup30::ref", "?")
    lk2 = d.subop("lookup_or_insert", streams=[a2], accesses=[arg(2)], stateType="HashMap", reference=ref2)
    rd2 = d.subop("reduce", streams=[lk2["ref"]], reference=ref2, updated=[{"member": "aggrVal$1", "expression": add(member("aggrVal$1"), L("l_quantity"))}])
    d.step([s2, ml] + p2 + [lk2, rd2], inputs=[(tlty, tl_, 0), ("?", v_co, 0), ("?", s_hm2, 0)])
    # ORDER BY o_totalprice DESC, o_orderdate LIMIT 100
    total = column("aggr1::tmp_attr0", "decimal(38,2)")
    hp_ = d.subop("create_heap", maxRows=100, sortBy=[{"member": "o_totalprice$9", "direction": "desc"}, {"member": "o_orderdate$9", "direction": "asc"}])  # EXT E4
    s_hp = d.step([hp_], results=[("?", hp_["ref"], 0)])
    keys = [("keyval$0", C("c_name")), ("keyval$1", C("c_custkey")), ("keyval$2", O("o_orderkey")), ("keyval$3", O("o_orderdate")), ("keyval$4", O("o_totalprice"))]
    sg2 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": m, "column": c} for m, c in keys] + [{"member": "aggrVal$1", "column": total}])
    outs = [("c_name", C("c_name")), ("c_custkey", C("c_custkey")), ("o_orderkey", O("o_orderkey")), ("o_orderdate", O("o_orderdate")), ("o_totalprice", O("o_totalprice")), ("sum_quantity", total)]
    mh2 = d.subop("materialize", streams=[sg2["ref"]], accesses=[arg(1)], stateType="Heap", mapping=[{"member": "%s$9" % n, "column": c} for n, c in outs])
    d.step([sg2, mh2], inputs=[("?", s_hm2, 0), ("?", s_hp, 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("heap0::%s" % n, c["datatype"])) for n, c in outs]
    sh = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$9" % n, "column": c} for n, c in final])
    mr = d.subop("materialize", streams=[sh["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$10" % n, "column": c} for n, c in final])
    d.step([sh, mr], inputs=[("?", s_hp, 0), ("ResultTable[...]", s_rt, 0)])
    return d.write()


def nl_band():
    """translateNLJ (RelAlgToSubOp.cpp:948-1033; test/lit/RelAlg/lowering.mlir:44-60): a join without an equality — the
    right side is materialised into a buffer, a nested_map scans it for every left tuple, a map + filter applies the
    predicate.  Not a TPC-H query: suppliers per nation with the key match written as the band
    `s_nationkey >= n_nationkey AND s_nationkey <= n_nationkey`, so that the answer is known."""
    d = Dump("nl_band")
    S, N = (lambda c: col("supplier", c)), (lambda c: col("nation", c))
    tn, tnty = get_external(d, "nation", [])
    ts, tsty = get_external(d, "supplier", [])
    buf = d.subop("generic_create")
    s_buf = d.step([buf], results=[("Buffer[...]", buf["ref"], 0)])
    sn = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("nation", ["n_nationkey", "n_name"]))
    mt = d.subop("materialize", streams=[sn["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "member$0", "column": N("n_nationkey")}, {"member": "member$1", "column": N("n_name")}])
    d.step([sn, mt], inputs=[(tnty, tn, 0), ("Buffer[...]", s_buf, 0)])
    hm = d.subop("generic_create")
    s_hm = d.step([hm], results=[("?", hm["ref"], 0)])
    ss = d.subop("scan", accesses=[arg(0)], mapping=scan_mapping("supplier", ["s_suppkey", "s_nationkey"]))
    sb = d.subop("scan", accesses=[arg(1)], mapping=[{"member": "member$0", "column": N("n_nationkey")}, {"member": "member$1", "column": N("n_name")}])
    ct = d.subop("combine_tuple", streams=[sb["ref"]])
    pred = column("map::pred", "int1")
    mp = d.subop("map", streams=[ct["ref"]], computed=[{"computed": pred, "expression": and_(inner(["", ">=", ""], [S("s_nationkey"), N("n_nationkey")]), inner(["", "<=", ""], [S("s_nationkey"), N("n_nationkey")]))}])
    fl = d.subop("filter", streams=[mp["ref"]], semantic="all_true", columns=[pred])
    nm = d.subop("nested_map", streams=[ss["ref"]], inputs=[], subops=[sb, ct, mp, fl])
    ref = column("lookup0::ref", "?")
    lk = d.subop("lookup_or_insert", streams=[nm["ref"]], accesses=[arg(2)], stateType="HashMap", reference=ref)
    rd = d.subop("reduce", streams=[lk["ref"]], reference=ref, updated=[{"member": "aggrVal$0", "expression": add(member("aggrVal$0"), const(1, "int64"))}])
    d.step([ss, nm, lk, rd], inputs=[(tsty, ts, 0), ("Buffer[...]", s_buf, 0), ("?", s_hm, 0)])
    b2 = d.subop("generic_create")
    s_b2 = d.step([b2], results=[("Buffer[...]", b2["ref"], 0)])
    cnt = column("aggr0::tmp_attr0", "int64")
    sg = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "keyval$0", "column": N("n_name")}, {"member": "aggrVal$0", "column": cnt}])
    outs = [("n_name", N("n_name")), ("suppliers", cnt)]
    mat = d.subop("materialize", streams=[sg["ref"]], accesses=[arg(1)], stateType="Buffer", mapping=[{"member": "%s$7" % n, "column": c} for n, c in outs])
    d.step([sg, mat], inputs=[("?", s_hm, 0), ("Buffer[...]", s_b2, 0)])
    sv = d.subop("create_sorted_view", accesses=[arg(0)], sortBy=[{"member": "n_name$7", "direction": "asc"}])  # EXT E4
    s_sv = d.step([sv], inputs=[("Buffer[...]", s_b2, 0)], results=[("SortedView Buffer[...]", sv["ref"], 0)])
    rt = d.subop("generic_create")
    s_rt = d.step([rt], results=[("ResultTable[...]", rt["ref"], 0)])
    final = [(n, column("sorted0::%s" % n, c["datatype"])) for n, c in outs]
    s3 = d.subop("scan", accesses=[arg(0)], mapping=[{"member": "%s$7" % n, "column": c} for n, c in final])
    m3 = d.subop("materialize", streams=[s3["ref"]], accesses=[arg(1)], stateType="ResultTable", mapping=[{"member": "%s$8" % n, "column": c} for n, c in final])
    d.step([s3, m3], inputs=[("SortedView Buffer[...]", s_sv, 0), ("ResultTable[...]", s_rt, 0)])
    return d.write()


if __name__ == "__main__":
    for f in (q6, q1, q3, q4, q4_probe_side, q5, q12, q18, nl_band):
        print(f())
