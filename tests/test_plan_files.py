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
