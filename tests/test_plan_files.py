"""The plan files shipped under lingo-db_amd/plans/tpch (CPU-only checks of the data itself)."""
import json
import os

import tpch_plans

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP_OPS = {"scan", "filter", "filter_dnf", "join_build", "join_probe", "groupby", "map", "sort", "topk", "materialize"}


def test_every_plan_file_is_listed():
    """one JSON file per TPC-H query, each naming its reference SQL and only inputs the runner provides"""
    for q in range(1, 23):
        with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "q%d.json" % q)) as f:
            plan = json.load(f)
        assert plan["ref"] == "resources/sql/tpch/%d.sql" % q
        assert sorted(plan["inputs"]) == sorted(tpch_plans.JSON_PLANS[q])
        outs = [s["out"] for s in plan["steps"]]
        assert len(outs) == len(set(outs)) and plan["result"] in outs
        assert {s["op"] for s in plan["steps"]} <= STEP_OPS


def _check(text, inputs):
    import ctypes as C

    from lingodb_amd import capi

    lib = capi.host_lib()
    arr = (C.c_char_p * max(len(inputs), 1))(*[n.encode() for n in inputs])
    st = lib.ldb_plan_json_check(text.encode(), arr, len(inputs))
    return st, lib.ldb_plan_json_last_error().decode(errors="replace")


def test_plan_files_pass_the_interpreters_structure_check():
    """the interpreter's own parser and step table accept every shipped plan (no device needed):
    values are defined before use, never twice, and the result is produced by a step"""
    for q in range(1, 23):
        with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "q%d.json" % q)) as f:
            text = f.read()
        st, err = _check(text, sorted(tpch_plans.JSON_PLANS[q]))
        assert st == 0, (q, err)


def test_structure_check_names_what_is_wrong():
    ok = '{"steps": [{"op": "filter", "in": "t", "out": "a", "preds": []}, {"op": "materialize", "in": "a", "cols": [], "out": "r"}], "result": "r"}'
    assert _check(ok, ["t"])[0] == 0
    cases = [
        ('{"steps": [', "plan JSON"),
        ('{"steps": [{"op": "frobnicate", "out": "x"}], "result": "x"}', "unknown step"),
        ('{"steps": [{"op": "filter", "in": "nope", "out": "a", "preds": []}], "result": "a"}', "before it exists"),
        ('{"steps": [{"op": "filter", "in": "t", "out": "a"}], "result": "a"}', "preds"),
        ('{"steps": [{"op": "filter", "in": "t", "out": "a", "preds": []}, {"op": "sort", "in": "a", "by": [], "out": "a"}], "result": "a"}', "defined twice"),
        ('{"steps": [{"op": "filter", "in": "t", "out": "a", "preds": []}], "result": "t"}', "not produced"),
        ('{"inputs": ["u"], "steps": [], "result": "u"}', "not provided"),
        ('{"steps": [{"op": "map", "in": "t", "as": "c", "out": "m"}], "result": "m"}', "expr"),
        ('{"steps": ' + "[" * 100 + "]" * 100 + ', "result": "x"}', "nesting"),
        ('{"steps": [], "result": "x\\', "unterminated escape"),  # a text ending inside an escape must not be read past its end
        ('{"steps": [], "result": "\\u12G4"}', "four hex digits"),
        ('{"steps": [], "result": "\\u12', "four hex digits"),
        ('{"steps": [], "result": "x", "n": -}', "digit expected"),  # (was: std::stoll's own exception text)
        ('{"steps": [], "result": "x", "n": 99999999999999999999999}', "out of range"),
        ('{"steps": [], "result": "\\ud83d"}', "surrogate"),  # a lone high surrogate
        ('{"steps": [], "result": "\\udc00x"}', "surrogate"),  # a lone low surrogate
    ]
    for text, needle in cases:
        st, err = _check(text, ["t"])
        assert st != 0 and needle in err, (text, err)
    # \r \b \f decode to the control characters and a surrogate pair to ONE code point (4 bytes of UTF-8): the value named
    # "a\r\U0001F600" below is found again under exactly that name
    good = '{"steps": [{"op": "filter", "in": "t", "out": "a\\r\\ud83d\\ude00", "preds": []}, {"op": "materialize", "in": "a\\u000d\U0001F600", "cols": [], "out": "r"}], "result": "r"}'
    st, err = _check(good, ["t"])
    assert st == 0, err


def test_database_generates_every_column_a_plan_names():
    """bench.py's Database generates only the columns the selected queries touch: every TPC-H column a plan
    file names must be among them, per query (checked with a recording stand-in for the device context)"""
    import re

    import tpch_data

    class Recorder:
        def __init__(self):
            self.tables = {}

        def tpch_generate(self, table_id, n_orders, part=0, n_parts=1, cols=None, narrow_decimals=False):
            names = [tpch_data.SCHEMAS[table_id][c][0] for c in (cols if cols is not None else range(len(tpch_data.SCHEMAS[table_id])))]
            self.tables.setdefault(table_id, set()).update(names)
            return ("table", table_id, tuple(names))

    all_cols = {name: tid for tid, fields in tpch_data.SCHEMAS.items() for name, _ in fields}
    for q in range(1, 23):
        rec = Recorder()
        db = tpch_plans.Database(rec, 1500, 0, 1, [q], False)
        with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "q%d.json" % q)) as f:
            text = f.read()
        provided = {name: getattr(db, attr) for name, attr in tpch_plans.JSON_PLANS[q].items()}
        assert all(v is not None for v in provided.values()), (q, provided)
        for col in set(re.findall(r'"(?:\d:)?((?:l|o|c|p|ps|s|n|r)_[a-z]+)"', text)):
            if col not in all_cols:
                continue  # a name the plan itself introduces (`as`, key_names)
            table = {"l": "lineitem", "o": "orders", "c": "customer", "p": "part", "ps": "partsupp", "s": "supplier", "n": "nation", "r": "region"}[col.split("_")[0]]
            if table not in provided:
                continue
            assert col in provided[table][2], (q, col, sorted(provided[table][2]))


# ---------------------------------------------------------------- the sharded plans (plans/tpch/dist)
DIST_OPS = STEP_OPS | {"allgather", "shuffle"}


def _dist_plan(q):
    with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "dist", "q%d.json" % q)) as f:
        text = f.read()
    return text, json.loads(text)


def test_every_query_has_a_sharded_plan_with_an_exchange():
    """all 22 queries are sharded as DATA: each plan has at least one exchange step, only reads inputs the runner
    provides (the query's tables + the replicated dimension tables it declares), and passes the structure check"""
    for q in range(1, 23):
        text, plan = _dist_plan(q)
        assert plan["ref"] == "resources/sql/tpch/%d.sql" % q
        ops = [s["op"] for s in plan["steps"]]
        assert set(ops) <= DIST_OPS and ("allgather" in ops or "shuffle" in ops), (q, ops)
        rep = plan.get("replicated_inputs", {})
        assert set(plan["inputs"]) <= set(tpch_plans.JSON_PLANS[q]) | set(rep), (q, plan["inputs"])
        for name, spec in rep.items():
            assert spec["table"] in tpch_plans.JSON_PLANS[q], (q, name, spec)
        st, err = _check(text, sorted(plan["inputs"]))
        assert st == 0, (q, err)


def test_sharded_plans_are_what_the_generator_writes():
    """tools/write_tpch_dist_plans.py derives the sharded plans from the single-GPU ones: the committed files are its output"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("write_tpch_dist_plans", os.path.join(ROOT, "tools", "write_tpch_dist_plans.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for q in range(1, 23):
        _, plan = _dist_plan(q)
        assert plan["steps"] == json.loads(json.dumps(mod.PLANS[q]["steps"])), q


def test_exchange_steps_are_checked():
    ok = '{"steps": [{"op": "materialize", "in": "t", "cols": ["a"], "out": "m"}, {"op": "allgather", "in": "m", "out": "g"}, {"op": "shuffle", "in": "g", "keys": ["a"], "cols": ["a"], "out": "r"}], "result": "r"}'
    assert _check(ok, ["t"])[0] == 0
    st, err = _check('{"steps": [{"op": "shuffle", "in": "t", "cols": ["a"], "out": "r"}], "result": "r"}', ["t"])
    assert st != 0 and "keys" in err
    st, err = _check('{"steps": [{"op": "allgather", "in": "nope", "out": "r"}], "result": "r"}', ["t"])
    assert st != 0 and "before it exists" in err


def test_bench_dry_run_checks_budgets_without_a_device():
    """`bench.py --dry-run`: per-rank rows, resident bytes and shuffle volume of BASELINE configs[4] (SF300 Q9 on 8 GPUs) against
    the HBM and uint32 row-id budgets — and a configuration that cannot fit says so"""
    import json
    import subprocess
    import sys

    def run(*args):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *args], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    d = run("--gpus", "8", "--sf", "300", "--queries", "9")
    assert d["checks"] == {"row_ids_fit_uint32": True, "hbm_fits": True}
    assert len(d["per_rank"]) == 8
    total_li = sum(r["rows"]["lineitem"] for r in d["per_rank"])
    n = 450_000_000
    assert total_li == (n // 7) * 28 + [0, 4, 5, 12, 15, 21, 23, 28][n % 7]
    assert sum(r["rows"]["orders"] for r in d["per_rank"]) == n
    sh = {s["input"]: s for s in d["shuffles_of_base_tables"]}
    assert set(sh) == {"lpy", "ps1"} and sh["lpy"]["rows_descend_from"] == "lineitem"
    # ≈ 11.8 GB egress per GPU over 7 links ≈ 11 ms when every lineitem row travels
    assert 10e9 < sh["lpy"]["bytes_out_per_rank_upper_bound"] < 14e9 and 9 < sh["lpy"]["ms_at_xgmi_link_rate_upper_bound"] < 13
    one = run("--gpus", "1", "--sf", "1000", "--queries", "9")
    assert one["checks"]["row_ids_fit_uint32"] is False and one["checks"]["hbm_fits"] is False  # 6 G lineitem rows on one GPU
    # the same arithmetic guards a real run: a configuration that cannot fit is refused before torch is imported or anything is generated
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sf", "300"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "does not fit" in out.stderr and "hbm_fits" in out.stderr, out.stderr[-500:]


def test_loop_and_nested_map_plans_pass_the_structure_check():
    """plans/subop/: the reference's loop.mlir counter and a k-means after kmeans.mlir, written with the `loop` / `nested_map` steps"""
    import json

    for name, inputs in (("loop_counter.json", ["ctr0"]), ("kmeans.json", ["points", "initial"])):
        with open(os.path.join(ROOT, "lingo-db_amd", "plans", "subop", name)) as f:
            text = f.read()
        st, err = _check(text, inputs)
        assert st == 0, (name, err)
    with open(os.path.join(ROOT, "lingo-db_amd", "plans", "subop", "loop_counter.json")) as f:
        good = json.load(f)

    def broken(edit):
        p = json.loads(json.dumps(good))
        edit(p["steps"][0])
        return json.dumps(p)

    cases = [(lambda l: l["vars"][0].update(init="nope"), "before it exists"),
             (lambda l: l["next"][0].update(var="other"), "unknown variable"),
             (lambda l: l["next"][0].update(**{"from": "ctr0"}), "not produced by the body"),
             (lambda l: l["continue"].update(**{"from": "later"}), "before it exists"),
             (lambda l: l["body"].append({"op": "materialize", "in": "s2", "cols": ["x"], "out": "cond"}), "defined twice"),
             (lambda l: l["results"][0].update(name="ctr0"), "defined twice"),
             (lambda l: l.pop("body"), "body")]
    for edit, needle in cases:
        st, err = _check(broken(edit), ["ctr0"])
        assert st != 0 and needle in err, (needle, err)
    # a value defined inside the body is not visible after the loop; the loop's result is
    after = json.loads(json.dumps(good))
    after["steps"][1]["in"] = "newCounter"
    st, err = _check(json.dumps(after), ["ctr0"])
    assert st != 0 and "before it exists" in err


def test_oracle_q6_in_slices_equals_the_whole_table():
    """bench.py's spot check at its own scale (checks.oracle_q6_at_bench_scale) runs the oracle's Q6 leg over host-generated SLICES of lineitem and adds
    the partial sums: the same number as the leg over the whole table (small scale here; the generator is counter-based, a slice is a row range)"""
    import sys

    sys.path[:0] = [os.path.join(ROOT, "oracle"), ROOT]
    import tpch_legs
    import tpch_plans

    n_orders = 60_000
    whole = tpch_legs.Legs(n_orders, queries=[6]).q6()
    got, secs = tpch_plans.oracle_q6_at_scale(n_orders, n_parts=7, threads=3)
    assert whole == [(got,)] and got is not None and got > 0 and secs >= 0


def test_oracle_q1_and_q3_in_slices_equal_the_whole_tables():
    """bench.py's checks.oracle_q1_at_bench_scale / oracle_q3_at_bench_scale (BASELINE configs[1] / [2] at the bench's own scale): Q1's partial sums
    and counts add up over lineitem slices; Q3 runs per order-range slice (all customers, the slice's orders and lineitems — orders and lineitem are
    cut at the same order boundaries) and the slices' ten best rows merge into the query's ten best"""
    import sys

    sys.path[:0] = [os.path.join(ROOT, "oracle"), ROOT]
    import tpch_legs
    import tpch_plans

    n_orders = 90_000
    whole1 = tpch_legs.Legs(n_orders, queries=[1]).q1()
    got1, _ = tpch_plans.oracle_q1_at_scale(n_orders, n_parts=7, threads=3)
    assert got1 == whole1 and len(got1) == 4
    whole3 = tpch_legs.Legs(n_orders, queries=[3]).q3()
    got3, _ = tpch_plans.oracle_q3_at_scale(n_orders, n_parts=7, threads=3)
    assert len(whole3) > 50 and got3[:10] == whole3[:10]
    assert tpch_plans.matches_legs(3, whole3[:10], got3) and not tpch_plans.matches_legs(3, whole3[1:11], got3)
    assert tpch_plans.matches_legs(1, whole1, got1) and not tpch_plans.matches_legs(1, whole1[:3], got1) and not tpch_plans.matches_legs(6, [], [])


def test_oracle_q9_and_q18_in_slices_equal_the_whole_tables():
    """bench.py's checks.oracle_q9_at_bench_scale / oracle_q18_at_bench_scale (round 6: BASELINE configs[4]'s query and the query with the round-5
    kernel paths): Q9's partial sums per (nation, year) add up over order-range slices joined against the green parts and the partsupp rows of those
    parts; Q18's big orders never cross a slice and meet the customers at the end"""
    import sys

    sys.path[:0] = [os.path.join(ROOT, "oracle"), ROOT]
    import tpch_legs
    import tpch_plans

    n_orders = 120_000
    whole9 = tpch_legs.Legs(n_orders, queries=[9]).q9()
    got9, _ = tpch_plans.oracle_q9_at_scale(n_orders, n_parts=7, threads=3, dim_parts=3)
    assert got9 == whole9 and len(got9) == 175
    assert tpch_plans.matches_legs(9, whole9, got9) and not tpch_plans.matches_legs(9, whole9[:-1], got9)
    whole18 = tpch_legs.Legs(n_orders, queries=[18]).q18()
    got18, _ = tpch_plans.oracle_q18_at_scale(n_orders, n_parts=7, threads=3)
    assert got18 == whole18 and len(got18) >= 3
    assert tpch_plans.matches_legs(18, whole18[:100], got18) and not tpch_plans.matches_legs(18, whole18[1:], got18)


def test_set_op_and_window_steps_are_checked():
    """the steps the dump consumer emits for set operations and windows (round 4): both inputs of a set_op must exist and it needs its kind and the
    two column lists; a window needs its functions"""
    ok = ('{"steps": [{"op": "set_op", "kind": "except_all", "left": "t", "left_cols": ["a"], "right": "u", "right_cols": ["b"], "out": "s"},'
          ' {"op": "window", "in": "s", "partition_by": ["a"], "order_by": [{"col": "a", "desc": true}], "frame_from": "unbounded_preceding", "frame_to": 0,'
          ' "fns": [{"fn": "rank", "as": "r"}, {"fn": "sum", "col": "a", "as": "x"}], "out": "w"},'
          ' {"op": "join_build", "in": "u", "keys": ["b"], "out": "h"}, {"op": "join_probe", "ht": "h", "in": "w", "keys": ["a"], "kind": "mark", "mark_as": "m", "out": "j"}], "result": "j"}')
    assert _check(ok, ["t", "u"])[0] == 0
    for text, needle in (
            ('{"steps": [{"op": "set_op", "kind": "union", "left": "t", "left_cols": ["a"], "right": "nope", "right_cols": ["b"], "out": "s"}], "result": "s"}', "before it exists"),
            ('{"steps": [{"op": "set_op", "left": "t", "left_cols": ["a"], "right": "u", "right_cols": ["b"], "out": "s"}], "result": "s"}', "kind"),
            ('{"steps": [{"op": "set_op", "kind": "union", "left": "t", "right": "u", "right_cols": ["b"], "out": "s"}], "result": "s"}', "left_cols"),
            ('{"steps": [{"op": "window", "in": "t", "order_by": ["a"], "out": "w"}], "result": "w"}', "fns")):
        st, err = _check(text, ["t", "u"])
        assert st != 0 and needle in err, (text, err)


def test_no_base_table_probes_a_non_unique_table_of_its_own_primary_key():
    """round 6: four hand plans kept a NON-unique hash table of a reduced lineitem side that every order probed with its primary key — pair
    counting, pairs and an expansion where the side with the unique key could have been the table (Q7 8.7 → 5.4 ms, Q8 4.9 → 3.4, Q12 5.1 → 3.4, Q21
    8.4 → 7.4).  No plan, single-GPU or sharded, does that any more: a relation that still carries a base table's primary key (the table itself, or a
    filter / semi join of it) never probes a non-unique table with that key.  (The two that remain probe with the CUSTOMER key into a handful of rows:
    Q10's 20 winners and Q18's few large orders.)"""
    primary = {"orders": "o_orderkey", "customer": "c_custkey", "part": "p_partkey", "supplier": "s_suppkey", "nation": "n_nationkey", "region": "r_regionkey"}
    allowed = {(10, "c_custkey"), (18, "c_custkey")}
    for sub in ("", "dist"):
        for q in range(1, 23):
            with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", sub, "q%d.json" % q)) as f:
                plan = json.load(f)
            base = {t: t for t in primary}  # relation name → the base table whose rows (a subset of them) it holds
            builds = {}
            for s in plan["steps"]:
                if s["op"] in ("filter", "filter_dnf") and s["in"] in base:
                    base[s["out"]] = base[s["in"]]
                elif s["op"] == "join_build":
                    builds[s["out"]] = s
                elif s["op"] == "join_probe":
                    if s["kind"] in ("semi", "anti") and s["in"] in base:
                        base[s["out"]] = base[s["in"]]
                    b = builds[s["ht"]]
                    if not b.get("unique", False) and s["in"] in base and s["keys"] == [primary[base[s["in"]]]]:
                        assert (q, s["keys"][0]) in allowed, (sub or "single", q, s)
