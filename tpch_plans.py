"""bench.py support: device-resident synthetic TPC-H database, query runner over the C++ plan
layer (libldb_host.so → C-ABI), the join-probe micro-benchmark, and the CPU-baseline leg.

Only cpu_baseline() touches the oracle (allowed: reported baseline, never the measured path)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

LINEITEM, ORDERS, CUSTOMER, PART, SUPPLIER, PARTSUPP, NATION = 0, 1, 2, 3, 4, 5, 6
# column ids of include/ldb_tpchgen.h
L_ORDERKEY, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_RETURNFLAG, L_LINESTATUS, L_SHIPDATE = 0, 4, 5, 6, 7, 8, 9, 10
L_COMMITDATE, L_RECEIPTDATE, L_SHIPMODE = 11, 12, 14
O_ORDERKEY, O_CUSTKEY, O_TOTALPRICE, O_ORDERDATE, O_ORDERPRIORITY, O_SHIPPRIORITY = 0, 1, 3, 4, 5, 6
C_CUSTKEY, C_MKTSEGMENT, C_NAME = 0, 3, 4


# Every single-GPU plan is data: lingo-db_amd/plans/tpch/qN.json, interpreted by libldb_host.so
# (ldb_plan_run_json).  Per query: plan input name → Database attribute.  Queries that show supplier
# strings read the wider supplier table.
_T = {n: n for n in ("lineitem", "orders", "customer", "part", "partsupp", "supplier", "nation", "region")}


def _inputs(*names, **over):
    d = {n: _T[n] for n in names}
    d.update(over)
    return d


JSON_PLANS = {
    1: _inputs("lineitem"), 6: _inputs("lineitem"), 3: _inputs("customer", "orders", "lineitem"), 4: _inputs("orders", "lineitem"), 12: _inputs("orders", "lineitem"),
    18: _inputs("customer", "orders", "lineitem"), 9: _inputs("part", "supplier", "lineitem", "partsupp", "orders", "nation"),
    5: _inputs("customer", "orders", "lineitem", "supplier", "nation", "region"), 7: _inputs("customer", "orders", "lineitem", "supplier", "nation"),
    8: _inputs("part", "supplier", "lineitem", "orders", "customer", "nation", "region"), 10: _inputs("customer", "orders", "lineitem", "nation"),
    11: _inputs("partsupp", "supplier", "nation"), 14: _inputs("part", "lineitem"), 15: _inputs("supplier", "lineitem"),
    2: _inputs("part", "partsupp", "nation", "region", supplier="supplier_full"), 13: _inputs("customer", "orders"), 16: _inputs("part", "partsupp", supplier="supplier_full"),
    17: _inputs("lineitem", "part"), 19: _inputs("lineitem", "part"), 20: _inputs("lineitem", "part", "partsupp", "nation", supplier="supplier_full"),
    21: _inputs("lineitem", "orders", "nation", supplier="supplier_full"), 22: _inputs("customer", "orders")}


class Database:
    """Slice rank/world of the SF database, generated straight into HBM (only the columns the
    selected queries touch — the reference likewise scans only referenced columns)."""

    @staticmethod
    def tables_for(queries):
        """[(Database attribute, generator table id, sorted column ids)] of everything the given queries touch — also what
        `bench.py --dry-run` sizes a rank's HBM from, without a device"""
        gen = []
        have = set()
        lcols = set()
        if 1 in queries:
            lcols |= {L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_RETURNFLAG, L_LINESTATUS, L_SHIPDATE}
        if 6 in queries:
            lcols |= {L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_SHIPDATE}
        if 3 in queries:
            lcols |= {L_ORDERKEY, L_EXTENDEDPRICE, L_DISCOUNT, L_SHIPDATE}
        if 4 in queries:
            lcols |= {L_ORDERKEY, L_COMMITDATE, L_RECEIPTDATE}
        if 12 in queries:
            lcols |= {L_ORDERKEY, L_SHIPDATE, L_COMMITDATE, L_RECEIPTDATE, L_SHIPMODE}
        if 18 in queries:
            lcols |= {L_ORDERKEY, L_QUANTITY}
        if 10 in queries:
            lcols |= {L_ORDERKEY, L_EXTENDEDPRICE, L_DISCOUNT, L_RETURNFLAG}
        if 15 in queries:
            lcols |= {2, L_EXTENDEDPRICE, L_DISCOUNT, L_SHIPDATE}  # + l_suppkey
        if 9 in queries:
            lcols |= {L_ORDERKEY, 1, 2, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT}  # + l_partkey, l_suppkey
        if 5 in queries or 7 in queries:
            lcols |= {L_ORDERKEY, 2, L_EXTENDEDPRICE, L_DISCOUNT, L_SHIPDATE}  # + l_suppkey
        if 14 in queries:
            lcols |= {1, L_EXTENDEDPRICE, L_DISCOUNT, L_SHIPDATE}  # + l_partkey
        if 8 in queries:
            lcols |= {L_ORDERKEY, 1, 2, L_EXTENDEDPRICE, L_DISCOUNT}  # + l_partkey, l_suppkey
        if 17 in queries:
            lcols |= {1, L_QUANTITY, L_EXTENDEDPRICE}
        if 19 in queries:
            lcols |= {1, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, 13, L_SHIPMODE}  # + l_partkey, l_shipinstruct
        if 20 in queries:
            lcols |= {1, 2, L_QUANTITY, L_SHIPDATE}
        if 21 in queries:
            lcols |= {L_ORDERKEY, 2, L_COMMITDATE, L_RECEIPTDATE}
        lcols |= {L_EXTENDEDPRICE, L_SHIPDATE}  # hbm_ceiling() calibration scans
        gen.append(("lineitem", LINEITEM, sorted(lcols))); have.add("lineitem")
        ocols, ccols = set(), set()
        if 3 in queries:
            ocols |= {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_SHIPPRIORITY}
            ccols |= {C_CUSTKEY, C_MKTSEGMENT}
        if 4 in queries:
            ocols |= {O_ORDERKEY, O_ORDERDATE, O_ORDERPRIORITY}
        if 12 in queries:
            ocols |= {O_ORDERKEY, O_ORDERPRIORITY}
        if 18 in queries:
            ocols |= {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_TOTALPRICE}
            ccols |= {C_CUSTKEY, C_NAME}
        if 10 in queries:
            ocols |= {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE}
            ccols |= {C_CUSTKEY, 1, 2, C_NAME}  # + c_nationkey, c_acctbal
        if 13 in queries:
            ocols |= {O_CUSTKEY, 7}  # + o_comment
            ccols |= {C_CUSTKEY}
        if 21 in queries:
            ocols |= {O_ORDERKEY, 2}  # + o_orderstatus
        if 22 in queries:
            ocols |= {O_CUSTKEY}
            ccols |= {C_CUSTKEY, 2, 5}  # + c_acctbal, c_phone
        if 9 in queries:
            ocols |= {O_ORDERKEY, O_ORDERDATE}
        pcols = set()
        if any(q in queries for q in (8, 9, 14)):  # p_partkey, [p_name,] [p_type]
            pcols |= {0} | ({3} if 9 in queries else set()) | ({4} if 14 in queries or 8 in queries else set())
        for q, cs in ((2, {0, 1, 4, 7}), (16, {0, 1, 4, 5}), (17, {0, 5, 6}), (19, {0, 1, 5, 6}), (20, {0, 3})):
            if q in queries:
                pcols |= cs
        if pcols:
            gen.append(("part", PART, sorted(pcols))); have.add("part")
        pscols = set()
        if 9 in queries or 11 in queries:  # ps_partkey, ps_suppkey, [ps_availqty,] ps_supplycost
            pscols |= {0, 1, 2, 3} if 11 in queries else {0, 1, 3}
        for q, cs in ((2, {0, 1, 3}), (16, {0, 1}), (20, {0, 1, 2})):
            if q in queries:
                pscols |= cs
        if pscols:
            gen.append(("partsupp", PARTSUPP, sorted(pscols))); have.add("partsupp")
        if 5 in queries or 8 in queries:
            ocols |= {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE}
            ccols |= {C_CUSTKEY, 1}  # + c_nationkey
            gen.append(("region", 7, [0, 1])); have.add("region")  # r_regionkey, r_name
        if 7 in queries:
            ocols |= {O_ORDERKEY, O_CUSTKEY}
            ccols |= {C_CUSTKEY, 1}  # + c_nationkey
        if 10 in queries and not any(q in queries for q in (5, 7, 8, 9, 11)):
            gen.append(("nation", NATION, [0, 1, 2])); have.add("nation")
        if 15 in queries and not any(q in queries for q in (5, 7, 8, 9, 11)):
            gen.append(("supplier", SUPPLIER, [0, 1])); have.add("supplier")
        if any(q in queries for q in (5, 7, 8, 9, 11)):
            gen.append(("supplier", SUPPLIER, [0, 1])); have.add("supplier")  # s_suppkey, s_nationkey
            gen.append(("nation", NATION, [0, 1, 2])); have.add("nation")  # n_nationkey, n_regionkey, n_name
        scols = set()
        for q, cs in ((2, {0, 1, 2, 3, 4, 5, 6}), (16, {0, 6}), (20, {0, 1, 3, 4}), (21, {0, 1, 3})):
            if q in queries:
                scols |= cs
        if scols:  # the wider supplier table of the queries that show supplier strings (s_name, s_address, …)
            gen.append(("supplier_full", SUPPLIER, sorted(scols | {0, 1}))); have.add("supplier_full")
            if "nation" not in have:
                gen.append(("nation", NATION, [0, 1, 2])); have.add("nation")
        if 2 in queries and "region" not in have:
            gen.append(("region", 7, [0, 1])); have.add("region")
        if ocols:
            gen.append(("orders", ORDERS, sorted(ocols))); have.add("orders")
        if ccols:
            gen.append(("customer", CUSTOMER, sorted(ccols))); have.add("customer")

        return gen

    def __init__(self, ctx, n_orders, rank, world, queries, narrow):
        self.ctx, self.n_orders, self.rank, self.world = ctx, n_orders, rank, world
        self.lineitem = self.orders = self.customer = self.part = self.supplier = self.partsupp = self.nation = self.region = None
        for attr, table_id, cols in self.tables_for(queries):
            setattr(self, attr, ctx.tpch_generate(table_id, n_orders, rank, world, cols, narrow))
        self.n_lineitem_total = (n_orders // 7) * 28 + [0, 4, 5, 12, 15, 21, 23, 28][n_orders % 7]


class Runner:
    """Runs TPC-H query q as data: the single-GPU plan lingo-db_amd/plans/tpch/qN.json, or — with a communicator
    (world > 1) or dist=True — the sharded plan plans/tpch/dist/qN.json, whose allgather / shuffle steps go
    through the library's exchange (ldb_plan_run_json_comm)."""

    def __init__(self, ctx, db, world, dist, torch, comm=None, force_dist=False, plans="files"):
        self.ctx, self.db, self.world, self.dist, self.torch = ctx, db, world, dist, torch
        self.plan_source = plans  # "files": the plan files; "subop": the reference-schema dumps (tests/golden/subop_tpch_qN.json) through ldb_subop_translate
        self.comm = comm  # api.Comm or None
        self.force_dist = force_dist
        self.last = {}
        self.cache = {}  # replicated dimension tables of the sharded plans: all-gathered once per database
        self.plans = {}
        self.meta = {}
        self.prepared = {}  # q -> api.PreparedPlan (ldb_plan_prepare: parsed once, later executions replay their read-back trace)
        self.prepared_on = os.environ.get("LDB_BENCH_PREPARED", "1") != "0"

    def sharded(self):
        return self.world > 1 or self.force_dist

    def run(self, q):
        if q not in JSON_PLANS:
            raise ValueError(f"TPC-H Q{q} has no plan")
        if self.world > 1 and self.comm is None:
            raise RuntimeError("world > 1 needs a communicator (api.Comm)")
        if self.prepared_on:
            if q not in self.prepared:
                self.prepared[q] = self.ctx.prepare_plan(self.plan_text(q))
            res = self.prepared[q].execute(self.plan_inputs(q), comm=self.comm)
        else:
            res = self.ctx.run_plan(self.plan_text(q), self.plan_inputs(q), comm=self.comm)
        self.last[q] = res
        return res

    def prepared_stats(self):
        """executions / replays / mis-speculations of the prepared plans + the context's descriptor-cache counters"""
        tot = {"executions": 0, "replays": 0, "misses": 0, "misses_by_query": {}, "readbacks_per_pass": 0, "host_issue_ms_per_replay": {}, "host_wait_ms_per_replay": {}}
        for q, p in sorted(self.prepared.items()):
            st = p.stats()
            tot["executions"] += st["executions"]
            tot["replays"] += st["replays"]
            tot["misses"] += st["misses"]
            if st["misses"]:
                tot["misses_by_query"]["Q%d" % q] = st["misses"]
            tot["readbacks_per_pass"] += st["readbacks"]
            if st["replays"]:  # how long the host needs to issue the plan vs how long it then waits for the device
                tot["host_issue_ms_per_replay"]["Q%d" % q] = round(st["issue_ms"] / st["replays"], 3)
                tot["host_wait_ms_per_replay"]["Q%d" % q] = round(st["wait_ms"] / st["replays"], 3)
        tot["descriptor_cache"] = self.ctx.desc_cache_stats()
        tot["order_dependent_misses"] = int(self.ctx.lib.ldb_gpu_order_dependent_misses())  # of `misses`: on the open-addressing build's run-length flag
        return tot

    def plan_inputs(self, q):
        inputs = {name: getattr(self.db, attr) for name, attr in JSON_PLANS[q].items()}
        if self.sharded():
            self.plan_text(q)
            for name, spec in self.meta[q].get("replicated_inputs", {}).items():
                key = (name, spec["table"], tuple(spec.get("cols", ())))
                if key not in self.cache:  # a static dimension table: one all-gather per database, itself a (one- or two-step) plan
                    steps = []
                    src = "t"
                    if spec.get("cols"):
                        steps.append({"op": "materialize", "in": "t", "cols": spec["cols"], "out": "m"})
                        src = "m"
                    steps.append({"op": "allgather", "in": src, "out": "result"})
                    import json

                    self.cache[key] = self.ctx.run_plan(json.dumps({"name": "replicate_" + name, "inputs": ["t"], "steps": steps, "result": "result"}),
                                                        {"t": inputs[spec["table"]]}, comm=self.comm)
                inputs[name] = self.cache[key]
            inputs = {n: t for n, t in inputs.items() if n in self.meta[q]["inputs"]}
        return inputs

    def plan_text(self, q):
        if q not in self.plans:
            import json

            if self.plan_source == "subop":  # f1: what a LingoDB with the GPU step handler would hand over — its own sub-operator dump
                if self.sharded():
                    raise RuntimeError("translated sub-operator dumps are single-GPU plans")
                from lingodb_amd import api

                self.plans[q] = api.translate_subop_dump(os.path.join(ROOT, "tests", "golden", "subop_tpch_q%d.json" % q), "tpch_q%d" % q)[0]
                self.meta[q] = json.loads(self.plans[q])
                return self.plans[q]
            sub = ("dist", "q%d.json" % q) if self.sharded() else ("q%d.json" % q,)
            with open(os.path.join(ROOT, "lingo-db_amd", "plans", "tpch", *sub)) as f:
                self.plans[q] = f.read()

            self.meta[q] = json.loads(self.plans[q])
        return self.plans[q]

    def probe_microbench(self, reps=3):
        """FK probe of l_orderkey into a table built on o_orderkey (100 % match), SURVEY §8(d).  With asynchronous specialisation (round 6) the
        first launches of a new kernel shape run the generic kernels: one untimed pass first, then wait for the compiler — the bench lines of
        round 6 up to run 25 timed the generic kernel here (8.7 ms where the specialised one takes 0.6)."""
        self._probe_microbench_once(1)
        _jit_wait()
        return self._probe_microbench_once(reps)

    def _probe_microbench_once(self, reps):
        from lingodb_amd import api, capi

        ctx, db = self.ctx, self.db
        orel, lrel = db.orders.rel(), db.lineitem.rel()
        ok, lk = db.orders.col("o_orderkey"), db.lineitem.col("l_orderkey")
        ctx.prof_reset()
        ht = orel.join_build([(0, ok)], unique=True)
        matches = 0
        for _ in range(reps):
            matches = ht.probe_count(lrel, [(0, lk)])
        prof = ctx.prof_all()
        n_b, ms_b = prof.get("k_join_build", (0, 0.0))
        n_p, ms_p = prof.get("k_join_probe_count", (0, 0.0))
        rows = db.lineitem.rows
        out = {"probe_rows": rows, "build_rows": db.orders.rows, "matches": matches, "table_slots": ht.slots, "table_bytes": ht.table_bytes,
               "bytes_per_key_value": round(ht.table_bytes / max(ht.slots, 1), 3)}  # 0.25 = rank bitmap, 4 = direct words, 8 x slots/keys = open addressing
        if n_p:
            avg = ms_p / n_p
            out["probe_ms"] = round(avg, 4)
            out["probe_grows_per_s"] = round(rows / (avg * 1e-3) / 1e9, 3)
            # byte model of SURVEY §8(d): key 4 B + one table word per probe (the sector-granular figure is the PMC traffic)
            out["survey_model_gbs"] = round(rows * 12 / (avg * 1e-3) / 1e9, 1)
        if n_b:
            out["build_ms"] = round(ms_b / n_b, 4)
            out["build_grows_per_s"] = round(db.orders.rows / (ms_b / n_b * 1e-3) / 1e9, 3)
        # unclustered control: the keys of uniformly random orders (same build side, 100 % match) — no
        # locality between consecutive probe rows, every probe is a random access into the table
        pk = ctx.tpch_generate(8, db.n_orders, db.rank, db.world, [0])
        ctx.prof_reset()
        for _ in range(reps):
            m2 = ht.probe_count(pk.rel(), [(0, 0)])
        pr = ctx.prof_all()
        n_p, ms_p = pr.get("k_join_probe_count", (0, 0.0))
        if n_p:
            # since round 4 the library partitions such a probe side itself (join_radix = -1: write-combining two-pass partition into
            # LDS-sized table slices + LDS-staged probe): the time of a probe is histogram + scatter + probe kernels together
            parts = {k: round(pr[k][1] / reps, 4) for k in ("k_radix_hist", "k_radix_scatter", "k_join_probe_count") if k in pr}
            avg = sum(parts.values())
            out["unclustered"] = {"probe_rows": pk.rows, "matches": m2, "probe_ms": round(avg, 4), "probe_grows_per_s": round(pk.rows / (avg * 1e-3) / 1e9, 3),
                                  "survey_model_gbs": round(pk.rows * 12 / (avg * 1e-3) / 1e9, 1), "kernels_ms": parts,
                                  "radix_partitioned": "k_radix_scatter" in parts}
        pk.release()
        ht.release()
        # selective variant: build side filtered to ≈10 % of orders (o_orderdate < 1992-09-01)
        ctx.prof_reset()
        sel = orel.scan_filter([api.pred((0, db.orders.col("o_orderdate")), capi.F_LT, 8279)])
        ht = sel.join_build([(0, ok)], unique=True)
        for _ in range(reps):
            matches = ht.probe_count(lrel, [(0, lk)])
        n_p, ms_p = ctx.prof_all().get("k_join_probe_count", (0, 0.0))
        if n_p:
            avg = ms_p / n_p
            out["selective"] = {"build_rows": sel.rows, "matches": matches, "table_slots": ht.slots, "table_bytes": ht.table_bytes, "probe_ms": round(avg, 4),
                                "probe_grows_per_s": round(rows / (avg * 1e-3) / 1e9, 3)}
        ht.release()
        sel.release()
        ctx.prof_reset()
        return out

    def hbm_ceiling(self, reps=5):
        """What this GPU's HBM delivers to simple streaming kernels, measured in the same run
        (SURVEY §8(d)): a device-to-device copy (read + write) and the library's own count-only
        scan over one 16-byte column (read only)."""
        torch, ctx, db = self.torch, self.ctx, self.db
        out = {}
        n = 1 << 30
        a = torch.empty(n, dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        a.fill_(1)
        b.copy_(a)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            b.copy_(a)
        ev1.record()
        torch.cuda.synchronize()
        out["copy_gbs"] = round(2 * 4 * n * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
        del a, b
        from lingodb_amd import api, capi

        col = db.lineitem.col("l_extendedprice")
        col4 = db.lineitem.col("l_shipdate")
        lrel = db.lineitem.rel()
        for c in (col, col4):  # (asynchronous specialisation: the first launch of a shape runs the generic kernel — one untimed call each, then wait)
            lrel.scan_count([api.pred((0, c), capi.F_GTE, 0)])
        _jit_wait()
        ctx.prof_reset()
        for _ in range(reps):
            lrel.scan_count([api.pred((0, col), capi.F_GTE, 0)])
        n_s, ms_s = ctx.prof_all().get("k_scan_count", (0, 0.0))
        if n_s:
            width = db.lineitem.col_width(col)
            out["scan_count_gbs"] = round(db.lineitem.rows * width / (ms_s / n_s * 1e-3) / 1e9, 1)
            out["scan_count_ms"] = round(ms_s / n_s, 4)
        # the same over a 4-byte column (second PMC calibration point: dword loads)
        ctx.prof_reset()
        for _ in range(reps):
            lrel.scan_count([api.pred((0, col4), capi.F_GTE, 0)])
        n_s, ms_s = ctx.prof_all().get("k_scan_count", (0, 0.0))
        if n_s:
            out["scan_count_4B_gbs"] = round(db.lineitem.rows * 4 / (ms_s / n_s * 1e-3) / 1e9, 1)
        ctx.prof_reset()
        return out


# ---------------------------------------------------------------- CPU baseline (oracle = reported, non-target)
def _canon(table):
    """GPU result rows in the oracle legs' conventions (decimals unscaled, dates as days, char(1) as the int32 of its 4 bytes)"""
    import pyarrow as pa
    from test_gpu_tpch_new import result_rows

    rows = result_rows(table)
    fsb = [i for i in range(table.num_columns) if pa.types.is_fixed_size_binary(table.schema.field(i).type)]
    if fsb:
        rows = [tuple(int.from_bytes(v, "little", signed=True) if i in fsb else v for i, v in enumerate(r)) for r in rows]
    return rows


# LIMIT queries whose ORDER BY keys do not decide every tie: (limit, key of a result row); compared on the key sequence + membership
_LIMITS = {3: (10, lambda r: (-r[1], r[2])), 10: (20, lambda r: -r[2]), 18: (100, lambda r: (-r[4], r[3]))}


def matches_legs(q, got, want, allow_empty=False):
    """GPU result rows of TPC-H Q`q` (in the legs' conventions, _canon) against the oracle leg's rows: bit-exact row for row; LIMIT queries on
    their ORDER BY keys + membership (ties beyond the keys are unspecified in the reference too); Q5 / Q11 (ORDER BY one aggregate) up to
    swaps of equal values.  An empty oracle result counts as a match only where the caller says the query may be empty (the check would be vacuous)."""
    if not want:
        return allow_empty and not got
    if q in _LIMITS:
        k, key = _LIMITS[q]
        return bool(len(got) == min(k, len(want)) and [key(r) for r in got] == [key(r) for r in want[: len(got)]] and set(got) <= set(want))
    if q in (5, 11):
        return bool([r[1] for r in got] == [r[1] for r in want] and sorted(got) == sorted(want))
    return bool(got == want)


def _cgroup_cpus():
    """CPUs the container may use (cgroup v2 cpu.max = quota / period), or None when unlimited / unknown: os.cpu_count() reports the machine's
    hardware threads, which a quota can be far below — the legs then stop scaling at the quota (round 5: 32 of 256 on the GPU box)"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        return None


def cpu_reference_legs(ctx, db, n_orders, results, checks, runs="3+10", budget_s=150.0):
    """TPC-H Q1 / Q6 / Q3 at the BENCH'S OWN scale as compiled morsel loops around the reference's real runtime objects (oracle/ref_baseline.py,
    oracle/ref_build/ref_glue.cpp over oracle/_ref: Restrictions::applyFilters, PreAggregationHashtableFragment + merge, GrowingBuffer, HashIndexedView::build,
    the scheduler interface) — SURVEY §8(d)'s CPU baseline, kind "reference".  The columns are the device tables' own bytes, copied back to the host
    once (outside every timing); `cores` = the container's CPU quota; the reference's benchmark.py protocol (3 warm-up + 10 measured, median and
    min).  The legs' results are compared with the rows the GPU returned in the timed region (checks.reference_objects_q*_at_bench_scale).  None when
    oracle/_ref is not there."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import math
    import statistics

    import numpy as np

    import ref_baseline
    import tpch_data as T
    import tpch_legs

    if not ref_baseline.available() or db.lineitem is None:
        return None
    t_all = time.perf_counter()
    quota = _cgroup_cpus()
    threads = max(1, int(quota) if quota else (os.cpu_count() or 1))
    warm, measured = (int(x) for x in runs.split("+"))

    def host(table, names):
        out = {}
        for nme in names:
            i = table.col(nme)
            raw = table.read_fixed(i)
            out[nme] = raw.view(np.dtype("V16")) if table.col_width(i) == 16 else raw.view(np.int32)
        return out

    t0 = time.perf_counter()
    want_li = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount"] + (["l_tax", "l_returnflag", "l_linestatus"] if 1 in results else []) + (["l_orderkey"] if 3 in results else [])
    li = host(db.lineitem, want_li)
    od = cu = None
    if 3 in results and db.orders is not None and db.customer is not None:
        od = host(db.orders, ["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
        cu = host(db.customer, ["c_custkey"])
        cu["c_segment4"] = ref_baseline.segment4(T.host_column(T.CUSTOMER, 3, n_orders))  # (the device keeps c_mktsegment as utf8: its first four bytes from the same generator on the host)
    copy_s = time.perf_counter() - t0
    med, mn, note = {}, {}, {}
    with ref_baseline.Session(threads) as s:
        legs = [(q, fn) for q, fn in ((1, lambda: s.q1(li)), (6, lambda: s.q6(li)), (3, lambda: s.q3(cu, od, li) if od is not None else None)) if q in results and (q != 3 or od is not None)]
        for q, fn in legs:
            ts, res = [], None
            for r in range(warm + measured):
                if time.perf_counter() - t_all > budget_s and len(ts) >= 3:
                    note["Q%d" % q] = "%d measured runs (time budget)" % len(ts)
                    break
                ms, res = fn()
                if r >= warm:
                    ts.append(ms)
            med[q], mn[q] = statistics.median(ts), min(ts)
            try:
                if q == 1:
                    ok = matches_legs(1, _canon(results[1]), tpch_legs.Legs.q1_finish(res))
                elif q == 6:
                    got = results[6].column(0)[0].as_py()
                    ok = (None if got is None else int(got.scaleb(results[6].schema.field(0).type.scale))) == res
                else:
                    ok = matches_legs(3, _canon(results[3]), res)
                checks["reference_objects_q%d_at_bench_scale" % q] = bool(ok)
            except Exception as e:
                checks["reference_objects_q%d_at_bench_scale" % q] = "%s: %s" % (type(e).__name__, e)
    done = sorted(med)
    if not done:
        return None
    gm = math.exp(sum(math.log(max(med[q], 1e-9)) for q in done) / len(done))
    return {"value": round(gm, 3), "unit": "ms", "cores": threads, "kind": "reference",
            "sample": "TPC-H SF%g %s at the bench's own scale over the device tables' bytes copied back to the host: compiled (-O2) morsel loops (20 000 rows) around the reference's real "
                      "runtime objects — Restrictions::applyFilters, PreAggregationHashtableFragment + PreAggregationHashtable::merge, GrowingBuffer, HashIndexedView::build, the scheduler "
                      "interface — built from /root/reference into oracle/_ref; the JIT-generated per-tuple loops restated after the lowerings (oracle/ref_build/ref_glue.cpp); %d threads (the "
                      "container's CPU quota); %d warm-up + %d measured runs, geomean of the medians" % (n_orders / 1_500_000, "+".join("Q%d" % q for q in done), threads, warm, measured),
            "per_query_median_ms": {"Q%d" % q: round(med[q], 3) for q in done}, "per_query_min_ms": {"Q%d" % q: round(mn[q], 3) for q in done}, "sample_sf": n_orders / 1_500_000,
            "device_to_host_copy_s": round(copy_s, 2), "hardware_threads": os.cpu_count(), "cgroup_cpu_limit": quota, **({"shortened": note} if note else {}),
            "seconds": round(time.perf_counter() - t_all, 1)}


def _jit_wait(timeout_ms=600_000):
    import ctypes as C

    from lingodb_amd import capi

    pend = C.c_int64()
    capi.gpu_lib().ldb_gpu_jit_wait(timeout_ms, C.byref(pend))
    return pend.value


def cpu_baseline(queries, sample_sf, runs="1+3", budget_s=100.0, ctx=None, narrow=False, checks=None):
    """The oracle legs (oracle/tpch_legs.py: the C restatement of the reference's CPU path — morsels of
    20 000 rows, HashIndexedView / PreAggregationHashtable restated — plus numpy for the few rows after
    the first aggregation; kind "port") timed on this host's cores over a bounded sample: the SAME
    generator as the GPU leg at `sample_sf`, `runs` = warm-up+measured passes per query (the reference's
    tools/scripts/benchmark.py uses 3+10), median and min per query.  No new leg starts after `budget_s`
    seconds; the line names the queries that were measured.  With a device context the GPU runs the same
    sample (same plans, same generator, same SF) so the two geomeans stand beside each other, and the GPU's
    Q1 / Q6 / Q3 results of that sample are compared bit-exactly with the legs' (`checks`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import math
    import statistics

    import tpch_legs

    n_orders = int(round(sample_sf * 1_500_000))
    legs = tpch_legs.Legs(n_orders, queries=list(queries))
    if os.environ.get("LDB_CPU_BASELINE_RESULTS"):  # tests: the legs' results instead of their times
        return {q: legs.run(q) for q in queries}
    warm, measured = (int(x) for x in runs.split("+"))
    t_gen = time.perf_counter()
    for tid in tpch_legs.Legs.NEED:  # host generation is registration time, not query time
        if any(q in tpch_legs.Legs.NEED[tid] for q in queries):
            legs.table(tid)
    gen_s = time.perf_counter() - t_gen
    # how many threads the legs are worth on this host: the C oracle starts `threads` workers per phase over morsels of 20 000 rows; beyond the
    # physical cores of one socket more threads only add start-up and memory-system contention (round 4: 256 threads ran Q1 at SF10 in 300 – 500 ms).
    # One probe query (Q1 when selected: the longest scan + aggregation leg), one run per candidate after a warm-up, the fastest count is used for
    # every leg and reported as `cores`
    calib = {}
    probe_q = 1 if 1 in queries else queries[0]
    all_threads = legs.threads
    legs.run(probe_q)
    for cand in sorted({all_threads, max(1, all_threads // 2), max(1, all_threads // 4), max(1, all_threads // 8), max(1, all_threads // 16), max(1, all_threads // 32)}, reverse=True):
        legs.threads = cand
        t0 = time.perf_counter()
        legs.run(probe_q)
        calib[cand] = (time.perf_counter() - t0) * 1000.0
    legs.threads = min(calib, key=calib.get)
    med, mn, leg_rows = {}, {}, {}
    t_start = time.perf_counter()
    order = [q for q in (1, 6, 3) if q in queries] + [q for q in queries if q not in (1, 6, 3)]  # the three configs[1..2] queries first: they carry the oracle check
    for q in order:
        if time.perf_counter() - t_start > budget_s and q not in (1, 6, 3):
            continue
        ts = []
        # the three queries of BASELINE configs[1..2] follow the reference's own protocol (tools/scripts/benchmark.py:32-33: 3 warm-up + 10
        # measured) whatever `runs` says; the others use `runs` inside the time budget
        w_q, m_q = (max(warm, 3), max(measured, 10)) if q in (1, 6, 3) else (warm, measured)
        for r in range(w_q + m_q):
            t0 = time.perf_counter()
            rows = legs.run(q)
            if r >= w_q:
                ts.append((time.perf_counter() - t0) * 1000.0)
        leg_rows[q] = rows
        med[q], mn[q] = statistics.median(ts), min(ts)
    done = [q for q in queries if q in med]
    gm = math.exp(sum(math.log(max(med[q], 1e-9)) for q in done) / max(len(done), 1))
    out = {"value": round(gm, 3), "unit": "ms", "cores": legs.threads, "kind": "port",
           "sample": "SF%g (%d orders; same generator and seed as the GPU leg): oracle restatement of the reference CPU path (reference binary not buildable offline), "
                     "%d threads, morsel 20000, %d warm-up + %d measured passes per query (Q1 / Q6 / Q3: 3 + 10, the reference's benchmark.py protocol); geomean of the per-query medians over %s%s" % (
                         sample_sf, n_orders, legs.threads, warm, measured, "+".join("Q%d" % q for q in done),
                         "" if len(done) == len(queries) else " (time budget %g s: %s not measured)" % (budget_s, "+".join("Q%d" % q for q in queries if q not in med))),
           "per_query_median_ms": {"Q%d" % q: round(med[q], 3) for q in done}, "per_query_min_ms": {"Q%d" % q: round(mn[q], 3) for q in done}, "sample_sf": sample_sf,
           "host_generation_s": round(gen_s, 2), "hardware_threads": all_threads, "cgroup_cpu_limit": _cgroup_cpus(),
           "thread_calibration_ms": {"Q%d with %d threads" % (probe_q, k): round(v, 1) for k, v in sorted(calib.items())}}
    if ctx is not None:
        # the GPU on the SAME sample: same plans, generator, seed and SF — the number to read beside `value`
        sdb = Database(ctx, n_orders, 0, 1, list(queries), narrow)
        srun = Runner(ctx, sdb, 1, None, None)
        g_med = {}
        got = {}
        for q in done:
            ts = []
            for r in range(2 + 3):
                t0 = time.perf_counter()
                got[q] = srun.run(q).to_arrow()
                ctx.sync()
                if r == 0:  # specialisations this scale asks for beyond the bench's own compile on worker threads: wait, then one more untimed run loads them
                    _jit_wait()
                if r >= 2:
                    ts.append((time.perf_counter() - t0) * 1000.0)
            g_med[q] = statistics.median(ts)
        out["gpu_same_sample"] = {"value": round(math.exp(sum(math.log(max(g_med[q], 1e-9)) for q in done) / max(len(done), 1)), 3), "unit": "ms",
                                  "per_query_median_ms": {"Q%d" % q: round(g_med[q], 3) for q in done}, "protocol": "2 warm-up + 3 measured, host wall clock around plan + result hand-over"}
        if checks is not None:  # every measured query at the sample scale, bit-exact against the oracle legs, in this run (BASELINE configs[1] / [2] are Q1 / Q3)
            ver = {}
            for q in done:
                if q in got and q in leg_rows:
                    # Q11's HAVING fraction is the constant 0.0001 of the reference's resources/sql/tpch/11.sql (the TPC-H text scales it by 1 / SF): from
                    # about SF 3 on no part reaches it and the query returns no row — on both sides
                    ver["Q%d" % q] = matches_legs(q, _canon(got[q]), leg_rows[q], allow_empty=(q == 11))
                    if not leg_rows[q]:
                        checks.setdefault("empty_results_at_sample", []).append("Q%d" % q)
            checks["oracle_bit_exact_at_sample_sf%g" % sample_sf] = ver
            checks["oracle_bit_exact_at_sample_all"] = bool(ver) and all(ver.values())
    return out


def _oracle_q6_slice(job):
    """one slice of oracle_q6_at_scale (no device, no torch)"""
    n_orders, part, n_parts = job
    import sys

    for sub in ("oracle", "tests", "lingo-db_amd"):
        path = os.path.join(ROOT, sub)
        if path not in sys.path:
            sys.path.insert(0, path)
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    leg = tpch_legs.Legs(n_orders, threads=1, queries=[6])
    leg._tables[T.LINEITEM] = oracle_bind.HostTable(T.host_table(T.LINEITEM, n_orders, part=part, n_parts=n_parts, cols=tpch_legs.Legs.NEED[T.LINEITEM][6]))
    (v,), = leg.q6()
    return v


def oracle_q6_at_scale(n_orders, n_parts=256, threads=None):
    """TPC-H Q6 by the ORACLE over the bench's own scale (verdict item 9: a spot check where the conservation laws are not enough): the four
    lineitem columns are generated on the host in `n_parts` slices of the same counter-based generator the device uses
    (tests/tpch_data.host_table with part / n_parts), each slice runs the oracle's Q6 leg (oracle/tpch_legs.py: the C restatement of the reference's scan
    + restrictions + key-less SUM), the partial sums add up in Python integers.  Slices run on a thread pool (the generator and the oracle are C calls
    that release the GIL: 19 M rows/s on 8 cores here, SF100 = 600 M rows).  Returns (unscaled sum or None, seconds)."""
    import concurrent.futures
    import time

    t0 = time.time()
    workers = threads or min(64, os.cpu_count() or 8)
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        parts = list(pool.map(_oracle_q6_slice, [(n_orders, part, n_parts) for part in range(n_parts)]))
    vals = [v for v in parts if v is not None]
    return (sum(vals) if vals else None), time.time() - t0


def _slice_paths():
    import sys

    for sub in ("oracle", "tests", "lingo-db_amd"):
        path = os.path.join(ROOT, sub)
        if path not in sys.path:
            sys.path.insert(0, path)


def _oracle_q1_slice(job):
    """one slice of oracle_q1_at_scale: the oracle's Q1 partial sums and counts per (returnflag, linestatus)"""
    n_orders, part, n_parts = job
    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    leg = tpch_legs.Legs(n_orders, threads=1, queries=[1])
    leg._tables[T.LINEITEM] = oracle_bind.HostTable(T.host_table(T.LINEITEM, n_orders, part=part, n_parts=n_parts, cols=tpch_legs.Legs.NEED[T.LINEITEM][1]))
    return leg.q1_partials()


def oracle_q1_at_scale(n_orders, n_parts=256, threads=None):
    """TPC-H Q1 (BASELINE configs[1]) by the ORACLE over the bench's own scale: lineitem generated on the host slice by slice (the counter-based
    generator the device uses), each slice through the oracle's Q1 leg up to its SUMs and COUNT (oracle/tpch_legs.py q1_partials: the C restatement of
    scan + restriction + PreAggregationHashtable), the partials added in Python integers, the averages taken at the end (q1_finish).
    Returns (rows in the legs' conventions, seconds)."""
    import concurrent.futures
    import time

    _slice_paths()
    import tpch_legs

    t0 = time.time()
    workers = threads or min(64, os.cpu_count() or 8)
    total = {}
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        for partial in pool.map(_oracle_q1_slice, [(n_orders, part, n_parts) for part in range(n_parts)]):
            for key, vals in partial.items():
                acc = total.setdefault(key, [0] * len(vals))
                for i, v in enumerate(vals):
                    acc[i] += v
    return tpch_legs.Legs.q1_finish(total), time.time() - t0


def _oracle_q3_slice(job):
    """one order-range slice of oracle_q3_at_scale: the oracle's Q3 over (all customers, the slice's orders, the slice's lineitems) — orders and
    lineitem are cut at the same order boundaries, so every group (l_orderkey) lies inside one slice; the slice's best rows by (revenue desc,
    orderdate), with everything that ties with the tenth"""
    n_orders, part, n_parts, customer, keep = job
    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    leg = tpch_legs.Legs(n_orders, threads=1, queries=[3])
    leg._tables[T.CUSTOMER] = customer
    for tid in (T.ORDERS, T.LINEITEM):
        leg._tables[tid] = oracle_bind.HostTable(T.host_table(tid, n_orders, part=part, n_parts=n_parts, cols=tpch_legs.Legs.NEED[tid][3]))
    rows = leg.q3()  # sorted by (-revenue, orderdate)
    if len(rows) > keep:
        last = (rows[keep - 1][1], rows[keep - 1][2])
        cut = keep
        while cut < len(rows) and (rows[cut][1], rows[cut][2]) == last:
            cut += 1
        rows = rows[:cut]
    return rows


def oracle_q3_at_scale(n_orders, n_parts=256, threads=None, keep=10):
    """TPC-H Q3 (BASELINE configs[2]) by the ORACLE over the bench's own scale: customer generated whole (15 M rows at SF100, two columns), orders and
    lineitem slice by slice at the same order boundaries; every slice runs the oracle's Q3 leg (C restatement of the scans, both hash joins and the
    aggregation) and keeps its ten best rows (+ ties); the merged list sorted by the query's ORDER BY is what the caller compares the GPU's ten rows
    with (matches_legs: key sequence + membership).  Returns (rows, seconds)."""
    import concurrent.futures
    import time

    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    t0 = time.time()
    customer = oracle_bind.HostTable(T.host_table(T.CUSTOMER, n_orders, cols=tpch_legs.Legs.NEED[T.CUSTOMER][3]))
    workers = threads or min(64, os.cpu_count() or 8)
    rows = []
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        for part_rows in pool.map(_oracle_q3_slice, [(n_orders, part, n_parts, customer, keep) for part in range(n_parts)]):
            rows.extend(part_rows)
    rows.sort(key=lambda r: (-r[1], r[2]))
    return rows, time.time() - t0


def _host_memory_room():
    """bytes this process may still allocate: MemAvailable, capped by the cgroup's limit minus its usage; None when unknown"""
    room = None
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    room = int(line.split()[1]) * 1024
        with open("/sys/fs/cgroup/memory.max") as f:
            lim = f.read().strip()
        if lim != "max":
            with open("/sys/fs/cgroup/memory.current") as f:
                cur = int(f.read().strip())
            room = min(room, int(lim) - cur) if room is not None else int(lim) - cur
    except Exception:
        pass
    return room


def _oracle_q9_slice(job):
    """one order-range slice of oracle_q9_at_scale: the oracle's Q9 joins over (the green parts, their partsupp rows, all suppliers, the slice's
    orders and lineitems) up to the partial sums per (nation, year)"""
    n_orders, part, n_parts, shared = job
    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    leg = tpch_legs.Legs(n_orders, threads=1, queries=[9])
    leg._tables.update(shared)
    for tid in (T.ORDERS, T.LINEITEM):
        leg._tables[tid] = oracle_bind.HostTable(T.host_table(tid, n_orders, part=part, n_parts=n_parts, cols=tpch_legs.Legs.NEED[tid][9]))
    return leg.q9_partials()


def oracle_q9_at_scale(n_orders, n_parts=None, threads=None, dim_parts=None):
    """TPC-H Q9 (the query BASELINE configs[4] names) by the ORACLE over the bench's own scale.  The dimension side once: part generated in `dim_parts`
    slices, each through the oracle's LIKE '%green%' restriction, the survivors kept (5.4 %); partsupp slice by slice through the oracle's semi join
    against those parts (the semi-join reduction: the rows the (l_partkey, l_suppkey) probe can reach); supplier and nation whole.  Then orders and
    lineitem slice by slice at the same order boundaries, every slice through the oracle's four hash joins (tpch_legs.q9_partials), the partial sums per
    (nation, year) added in Python integers.  Every slice builds its own hash table on the green partsupp rows (4.3 M at SF100), so the slices are
    few and large: two per worker, at most ≈ 30 M lineitems each (≈ 2 GB of host columns per worker).  Returns (rows in the legs' conventions, seconds)."""
    import concurrent.futures
    import time

    import pyarrow as pa

    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    t0 = time.time()
    workers = threads or min(32, os.cpu_count() or 8)
    n_parts = n_parts or max(2 * workers, -(-n_orders * 4 // 30_000_000))
    # a worker holds one slice's columns (≈ 64 B per lineitem + 12 B per order) and about as much again in row ids and gathered values: never more
    # workers than a third of the memory this process may still take
    room = _host_memory_room()
    if room is not None:
        per_worker = 3 * (n_orders * 4 // n_parts) * 80
        workers = max(1, min(workers, int(room / 3 // max(per_worker, 1))))
    dim_parts = dim_parts or max(1, min(64, n_orders // 4_000_000))
    need = tpch_legs.Legs.NEED

    def green_part(p):
        leg = tpch_legs.Legs(n_orders, threads=1, queries=[9])
        t = oracle_bind.HostTable(T.host_table(T.PART, n_orders, part=p, n_parts=dim_parts, cols=need[T.PART][9]))
        fr = tpch_legs.Frame(leg, [(t, None)]).where(("p_name", "LIKE", "%green%"))
        return t.arrow.take(pa.array(fr.rel.phys(0)))

    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        green = oracle_bind.HostTable(pa.concat_tables(list(pool.map(green_part, range(dim_parts)))).combine_chunks())

        def green_partsupp(p):
            leg = tpch_legs.Legs(n_orders, threads=1, queries=[9])
            t = oracle_bind.HostTable(T.host_table(T.PARTSUPP, n_orders, part=p, n_parts=dim_parts, cols=need[T.PARTSUPP][9]))
            fr = tpch_legs.Frame(leg, [(t, None)]).join(tpch_legs.Frame(leg, [(green, None)]), [("ps_partkey", "p_partkey")], "semi")
            return t.arrow.take(pa.array(fr.rel.phys(0)))

        shared = {T.PART: green, T.PARTSUPP: oracle_bind.HostTable(pa.concat_tables(list(pool.map(green_partsupp, range(dim_parts)))).combine_chunks()),
                  T.SUPPLIER: oracle_bind.HostTable(T.host_table(T.SUPPLIER, n_orders, cols=need[T.SUPPLIER][9])),
                  T.NATION: oracle_bind.HostTable(T.host_table(T.NATION, n_orders, cols=need[T.NATION][9]))}
        total = {}
        for partial in pool.map(_oracle_q9_slice, [(n_orders, part, n_parts, shared) for part in range(n_parts)]):
            for key, v in partial.items():
                total[key] = total.get(key, 0) + v
    fin = tpch_legs.Legs(n_orders, threads=1, queries=[9])
    fin._tables.update(shared)
    return fin.q9_finish(total), time.time() - t0


def _oracle_q18_slice(job):
    """one order-range slice of oracle_q18_at_scale: the oracle's group-by of the slice's lineitems by order and the semi join of its orders"""
    n_orders, part, n_parts = job
    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    leg = tpch_legs.Legs(n_orders, threads=1, queries=[18])
    for tid in (T.ORDERS, T.LINEITEM):
        leg._tables[tid] = oracle_bind.HostTable(T.host_table(tid, n_orders, part=part, n_parts=n_parts, cols=tpch_legs.Legs.NEED[tid][18]))
    return leg.q18_big()


def oracle_q18_at_scale(n_orders, n_parts=256, threads=None):
    """TPC-H Q18 by the ORACLE over the bench's own scale: orders and lineitem slice by slice at the same order boundaries (a group — one order's
    lines — never crosses a slice), every slice through the oracle's PreAggregationHashtable restatement and the semi join (tpch_legs.q18_big); the few
    orders above 300 units are joined with the whole customer table at the end (q18_finish) and sorted by the query's ORDER BY.  The caller compares
    the GPU's hundred rows with matches_legs (key sequence + membership).  Returns (rows, seconds)."""
    import concurrent.futures
    import time

    _slice_paths()
    import oracle_bind
    import tpch_data as T
    import tpch_legs

    t0 = time.time()
    workers = threads or min(64, os.cpu_count() or 8)
    big = []
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        for rows in pool.map(_oracle_q18_slice, [(n_orders, part, n_parts) for part in range(n_parts)]):
            big.extend(rows)
    fin = tpch_legs.Legs(n_orders, queries=[18])
    fin._tables[T.CUSTOMER] = oracle_bind.HostTable(T.host_table(T.CUSTOMER, n_orders, cols=tpch_legs.Legs.NEED[T.CUSTOMER][18]))
    return fin.q18_finish(big), time.time() - t0
