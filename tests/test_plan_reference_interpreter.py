"""The plan files and the plans `ldb_subop_translate` makes of the sub-operator dumps, run WITHOUT a device by a plain-Python reading of the plan
language (tests/plan_ref.py: test infrastructure, typed by the host library's own decimal rules) over a small generated database, against the oracle
legs (oracle/tpch_legs.py) — the CPU half of what tests/test_gpu_sf1_oracle.py checks on the GPU at SF1: both step lists of a query say what the
oracle computes."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), ROOT]

import plan_ref  # noqa: E402
import tpch_data as T  # noqa: E402
import tpch_legs  # noqa: E402
from lingodb_amd import api  # noqa: E402

N_ORDERS = 30_000  # SF 0.02: 120 000 lineitems
TABLES = {"lineitem": T.LINEITEM, "orders": T.ORDERS, "customer": T.CUSTOMER, "part": T.PART, "supplier": T.SUPPLIER, "partsupp": T.PARTSUPP, "nation": T.NATION, "region": T.REGION}
LIMITS = {3: (10, lambda r: (-r[1], r[2])), 10: (20, lambda r: -r[2]), 18: (100, lambda r: (-r[4], r[3])), 2: (100, None), 21: (100, None)}


@pytest.fixture(scope="module")
def world():
    tables = {name: plan_ref.table_from_arrow(T.host_table(tid, N_ORDERS)) for name, tid in TABLES.items()}
    return tables, tpch_legs.Legs(N_ORDERS)


def run(tables, plan):
    return plan_ref.rows(plan_ref.Interp({n: tables[n] for n in plan["inputs"]}).run(plan))


def hand(q):
    with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "q%d.json" % q)) as f:
        return json.load(f)


def translated(q):
    with open(os.path.join(ROOT, "tests", "golden", "subop_tpch_q%s.json" % q)) as f:
        return json.loads(api.translate_subop_dump(f.read(), "tpch_q%s" % q)[0])


def same(q, got, want):
    if q in (3, 10, 18):
        k, key = LIMITS[q]
        assert len(got) == min(k, len(want)) and [key(r) for r in got] == [key(r) for r in want[: len(got)]] and set(got) <= set(want)
    elif q in (5, 11):  # ORDER BY one aggregate: equal values may swap
        assert [r[1] for r in got] == [r[1] for r in want] and sorted(got) == sorted(want)
    else:
        assert got == want


@pytest.mark.parametrize("q", list(range(1, 23)))
def test_plan_file_and_translated_dump_say_what_the_oracle_computes(world, q):
    tables, legs = world
    want = legs.run(q)
    assert want, "empty oracle result: the check would be vacuous"
    same(q, run(tables, hand(q)), want)
    same(q, run(tables, translated(q)), want)


def test_the_second_semi_join_form_of_q4(world):
    tables, legs = world
    same(4, run(tables, translated("4_probe_side")), legs.run(4))


def test_pattern_dumps_say_what_plain_python_computes(world):
    """the lowering patterns TPC-H does not exercise (tests/golden/subop_pat_*.json): translated and run here, held against the same plain-Python
    expectations tests/test_gpu_sf1_oracle.py::test_pattern_dumps_match_numpy uses on the GPU"""
    import collections
    import decimal

    tables, _ = world
    arrow = {name: T.host_table(tid, N_ORDERS) for name, tid in (("supplier", T.SUPPLIER), ("nation", T.NATION), ("customer", T.CUSTOMER))}
    col = lambda t, c: arrow[t].column(c).to_pylist()
    s_key, s_nat, s_bal = col("supplier", "s_suppkey"), col("supplier", "s_nationkey"), col("supplier", "s_acctbal")
    n_key, n_reg, n_name = col("nation", "n_nationkey"), col("nation", "n_regionkey"), dict(zip(col("nation", "n_nationkey"), col("nation", "n_name")))
    c_nat, c_bal = col("customer", "c_nationkey"), col("customer", "c_acctbal")
    rich, richest = decimal.Decimal("9000.00"), decimal.Decimal("9990.00")

    def pat(name):
        with open(os.path.join(ROOT, "tests", "golden", "subop_pat_%s.json" % name)) as f:
            return run(tables, json.loads(api.translate_subop_dump(f.read(), "pat_" + name)[0]))

    reg_of = dict(zip(n_key, n_reg))
    assert pat("between") == [(k, k - reg_of[n]) for k, n in sorted(zip(s_key, s_nat)) if 10 <= k < 20] and len(pat("between")) == 10
    region1 = {k for k, r in zip(n_key, n_reg) if r == 1}
    assert pat("mark") == [(k,) for k in sorted(k for k, n, b in zip(s_key, s_nat, s_bal) if n in region1 or b > rich)]
    per_nation, total = collections.Counter(), collections.Counter()
    for n, b in zip(s_nat, s_bal):
        if b > rich:
            per_nation[n] += 1
            total[n] += int(b.scaleb(2))
    assert pat("right_outer") == [(k, per_nation.get(k, 0)) for k in sorted(n_key)]
    assert pat("groupjoin") == [(k, n_name[k], per_nation[k], total[k]) for k in sorted(per_nation)]
    per9990, tot9990 = collections.Counter(), collections.Counter()
    for n, b in zip(s_nat, s_bal):
        if b > richest:
            per9990[n] += 1
            tot9990[n] += int(b.scaleb(2))
    assert pat("groupjoin_outer") == [(k, n_name[k], per9990.get(k, 0), tot9990[k] if k in tot9990 else None) for k in sorted(n_key)]
    region2 = {k for k, r in zip(n_key, n_reg) if r == 2}
    richest_s = [(k, n) for k, n, b in zip(s_key, s_nat, s_bal) if b > richest]
    m = [(k, n) for k, n in richest_s if n in region2]
    ls = [(k, n) for k, n in richest_s if n not in region2]
    ln = [n for n in region2 if n not in {x for _, x in richest_s}]
    sum_or_none = lambda xs: sum(xs) if xs else None
    assert pat("full_outer") == [(len(m) + len(ls) + len(ln), len(m) + len(ls), len(m) + len(ln), sum_or_none([k for k, _ in m + ls]), sum_or_none([n for _, n in m] + ln))]
    order = sorted(range(len(s_key)), key=lambda i: s_key[i])
    bal = [int(s_bal[i].scaleb(2)) for i in order]
    assert pat("window") == [(s_key[i], min(p, 2) + 1, sum(bal[max(0, p - 2): p + 1]), min(p, 2) + 1) for p, i in enumerate(order)]
    seen, running, want = collections.Counter(), collections.Counter(), []
    for p, i in enumerate(order):
        seen[s_nat[i]] += 1
        running[s_nat[i]] += bal[p]
        want.append((s_key[i], seen[s_nat[i]], running[s_nat[i]], seen[s_nat[i]]))
    assert pat("window_part") == want
    nat_total, nat_rows = collections.Counter(), collections.Counter(s_nat)
    for p, i in enumerate(order):
        nat_total[s_nat[i]] += bal[p]
    assert pat("window_total") == [(s_key[i], sum(bal), len(order)) for i in order]
    assert pat("window_total_part") == [(s_key[i], nat_total[s_nat[i]], nat_rows[s_nat[i]]) for i in order]
    left = [n for n, b in zip(c_nat, c_bal) if b > rich]
    right = [n for n, b in zip(s_nat, s_bal) if b > richest]
    cl, cr = collections.Counter(left), collections.Counter(right)
    expected = {"union_all": sorted(left + right), "union": sorted(set(left) | set(right)), "intersect": sorted(set(left) & set(right)), "except": sorted(set(left) - set(right)),
                "intersect_all": sorted((cl & cr).elements()), "except_all": sorted((cl - cr).elements())}
    for kind, rows_ in expected.items():
        assert pat(kind) == [(k,) for k in rows_], kind


@pytest.mark.parametrize("world_size", [2, 3, 8])
def test_sharded_plans_on_every_rank_say_what_the_oracle_computes(world, world_size):
    """§8(e) without devices: the sharded step lists (plans/tpch/dist/qN.json) run in lockstep on `world_size` simulated ranks over the shards the
    generator gives each rank (orders / lineitem co-located by order ranges, the other tables by row ranges, nation / region replicated), their
    allgather / shuffle steps executed between the ranks — EVERY rank ends with the oracle's answer for the whole database"""
    _, legs = world
    shards = [{name: plan_ref.table_from_arrow(T.host_table(tid, N_ORDERS, part=r, n_parts=world_size)) for name, tid in TABLES.items()} for r in range(world_size)]
    for q in range(1, 23):
        with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", "dist", "q%d.json" % q)) as f:
            plan = json.load(f)
        inputs = []
        for r in range(world_size):
            t = dict(shards[r])
            for name, spec in plan.get("replicated_inputs", {}).items():  # a static dimension table the host program all-gathers once per database
                cols = spec.get("cols") or list(shards[0][spec["table"]].sides[0])
                t[name] = plan_ref.Rel([{c: (shards[0][spec["table"]].sides[0][c][0], [x for sh in shards for x in sh[spec["table"]].sides[0][c][1]]) for c in cols}],
                                       sum(sh[spec["table"]].n for sh in shards))
            inputs.append({n: t[n] for n in plan["inputs"]})
        want = legs.run(q)
        for r, res in enumerate(plan_ref.run_sharded(plan, inputs)):
            same(q, plan_ref.rows(res), want)


def test_loop_and_nested_map_plans_of_the_reference_lit_tests():
    """plans/subop/loop_counter.json (test/lit/SubOp/loop.mlir, CHECK: 6) and kmeans.json (test/lit/SubOp/kmeans.mlir in fixed point x 1000: the lit
    test's CHECK values truncated to three digits) — what tests/test_gpu_f4.py pins on the device, read here without one"""
    import pyarrow as pa

    def plan(name):
        with open(os.path.join(ROOT, "lingo-db_amd", "plans", "subop", name)) as f:
            return json.load(f)

    ctr = plan_ref.table_from_arrow(pa.table({"ctr": pa.array([0], pa.int32())}))
    assert plan_ref.rows(plan_ref.Interp({"ctr0": ctr}).run(plan("loop_counter.json"))) == [(6,)]
    pts = [(x * 1000, y * 1000) for x, y in [(1, 1), (1, 2), (2, 1), (2, 4), (2, 5), (3, 2), (3, 5), (6, 3), (6, 5), (8, 4)]]  # kmeans.mlir's ten points
    points = plan_ref.table_from_arrow(pa.table({"px": pa.array([p[0] for p in pts], pa.int64()), "py": pa.array([p[1] for p in pts], pa.int64())}))
    initial = plan_ref.table_from_arrow(pa.table({"cx": pa.array([pts[i][0] for i in range(3)], pa.int64()), "cy": pa.array([pts[i][1] for i in range(3)], pa.int64()),
                                                  "cid": pa.array([0, 1, 2], pa.int64())}))
    assert plan_ref.rows(plan_ref.Interp({"points": points, "initial": initial}).run(plan("kmeans.json"))) == [(0, 1750, 1500), (1, 2333, 4666), (2, 6666, 4000)]
