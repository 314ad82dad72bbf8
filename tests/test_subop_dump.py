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


@pytest.mark.parametrize("q", [6, 1, 3, 4, "4_probe_side", 5, 12, 18])
def test_dump_translates_to_a_checked_plan(q):
    text, report = api.translate_subop_dump(dump(q), "tpch_q%s" % q)
    plan = json.loads(text)
    assert all(r["target"] == "gpu" for r in report) and len(report) == len(json.loads(dump(q)))
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
        for step in d:
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
    assert got[0]["preds"] == want[0]["preds"] and got[1]["keys"] == want[1]["keys"] and got[1]["unique"] is False
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
    assert rep[d[0]["ref"]]["target"] == "gpu" and rep[d[-1]["ref"]]["target"] == "cpu"  # what follows cannot be placed either
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


def test_dumps_are_what_the_generator_writes(tmp_path):
    import subprocess
    import sys

    qs = (6, 1, 3, 4, "4_probe_side", 5, 12, 18)
    before = {q: dump(q) for q in qs}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "write_subop_dumps.py")], stdout=subprocess.DEVNULL)
    assert {q: dump(q) for q in qs} == before


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
