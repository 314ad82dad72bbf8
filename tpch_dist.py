"""Multi-GPU TPC-H (bench.py, world > 1): one process per GPU, database sharded by order ranges
(lineitem co-partitioned with orders, customer by row ranges), shard-local plans + RCCL exchange
of the small partial relations (replicate small build sides / partial aggregates).

  Q1, Q6: local partial aggregation → all-gather of the (<= 6 row) partial tables → merge.
  Q3    : all-gather of the filtered customer keys (replicated build side) → local joins and
          group-by (order keys are disjoint across shards) → all-gather of the shard top-10s →
          final top-10.
Every rank ends with the full result (the merge is replicated; it is microseconds of work).
"""
import ctypes as C
import os

import torch

from lingodb_amd import capi, dist as ldist
from lingodb_amd.api import Table
from lingodb_amd.capi import ColType, check, check_plan


def _plan(ctx, name, *tables):
    t = C.c_void_p()
    check_plan(getattr(capi.host_lib(), name)(ctx.h, *[x.h for x in tables], C.byref(t)))
    return Table(ctx, t)


def table_to_tensors(ctx, table):
    """fixed-width device table → list of uint8 CUDA tensors (device-to-device copies)"""
    cols, widths = [], []
    n = table.rows
    for c in range(table.n_cols):
        w = table.col_width(c)
        if w <= 0:
            raise ValueError("only fixed-width columns are exchanged")
        values, _, validity, _ = table.col_ptrs(c)
        t = torch.empty(max(n * w, 1), dtype=torch.uint8, device="cuda")
        check(ctx.lib.ldb_gpu_memcpy_d2d(ctx.h, C.c_void_p(t.data_ptr()), C.c_void_p(values), n * w))
        cols.append(t)
        widths.append(w)
    ctx.sync()
    return cols, widths


def tensors_to_table(ctx, like, cols, n_rows, name):
    """uint8 CUDA tensors → a new device table with the column types/names of `like`"""
    nc = like.n_cols
    types = (ColType * nc)(*[like.coltype(c) for c in range(nc)])
    names = [like.col_name(c).encode() for c in range(nc)]
    name_arr = (C.c_char_p * nc)(*names)
    h = C.c_void_p()
    narrow = 1 if any(like.col_width(c) == 8 and like.coltype(c).type == capi.T_DECIMAL128 for c in range(nc)) else 0
    check(ctx.lib.ldb_gpu_table_alloc(ctx.h, name.encode(), nc, types, name_arr, n_rows, None, narrow, C.byref(h)))
    out = Table(ctx, h)
    torch.cuda.synchronize()
    for c in range(nc):
        values, _, _, _ = out.col_ptrs(c)
        check(ctx.lib.ldb_gpu_memcpy_d2d(ctx.h, C.c_void_p(values), C.c_void_p(cols[c].data_ptr()), n_rows * out.col_width(c)))
    ctx.sync()
    return out


def _d2d(ctx, dst_ptr, src_ptr, nbytes):
    if nbytes:
        check(ctx.lib.ldb_gpu_memcpy_d2d(ctx.h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes))


def replicate(runner, table, name):
    """all-gather a small table: every rank gets the concatenation in rank order.  Fixed-width
    columns travel as they are; a utf8 column travels as its lengths (int64 per row) plus one
    separate gather of its bytes, and the offsets are rebuilt on arrival.
    NULLs: a column that has a validity bitmap on ANY rank travels with one validity byte per row
    (ldb_gpu_table_validity_bytes → gather → ldb_gpu_table_set_validity_bytes): a keyless partial
    SUM of a shard without qualifying rows stays NULL and the merge ignores it."""
    if getattr(runner, "comm", None) is not None:  # RCCL inside the library (ldb_gpu_allgather): no torch staging, validity travels along
        return runner.comm.allgather(table, name)
    ctx, n, nc = runner.ctx, table.rows, table.n_cols
    staged = runner.dist.get_backend() == "gloo"  # functional testing of the N>1 path on one GPU
    cols, widths, blobs = [], [], {}
    for c in range(nc):
        values, offsets, _, _ = table.col_ptrs(c)
        if table.coltype(c).type == capi.T_UTF8:
            off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()  # the fill runs on torch's stream, the copy below on the ctx stream
            _d2d(ctx, off.data_ptr(), offsets, 8 * (n + 1) if n else 0)
            ctx.sync()
            first, total = (int(off[0].item()), int((off[n] - off[0]).item())) if n else (0, 0)
            blob = torch.empty(max(total, 1), dtype=torch.uint8, device="cuda")
            _d2d(ctx, blob.data_ptr(), values + first, total)
            cols.append((off[1:] - off[:-1]).contiguous().view(torch.uint8) if n else torch.empty(1, dtype=torch.uint8, device="cuda"))
            widths.append(8)
            blobs[c] = (blob, total)
        else:
            w = table.col_width(c)
            t = torch.empty(max(n * w, 1), dtype=torch.uint8, device="cuda")
            _d2d(ctx, t.data_ptr(), values, n * w)
            cols.append(t)
            widths.append(w)
    # which columns carry NULLs anywhere: agreed by a MAX over the ranks' flags (one tiny collective)
    flags = torch.tensor([1 if table.col_ptrs(c)[2] else 0 for c in range(nc)] or [0], dtype=torch.int32, device="cpu" if staged else "cuda")
    runner.dist.all_reduce(flags, op=runner.dist.ReduceOp.MAX)
    nullable = [c for c in range(nc) if int(flags[c].item())] if nc else []
    for c in nullable:
        vb = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
        check(ctx.lib.ldb_gpu_table_validity_bytes(ctx.h, table.h, c, C.c_void_p(vb.data_ptr())))
        cols.append(vb)
        widths.append(1)
    ctx.sync()

    def gather(tensors, ws, rows):
        if staged:
            tensors = [t.cpu() for t in tensors]
        out, counts = ldist.allgather_columns(runner.dist, tensors, ws, rows)
        return ([t.cuda() for t in out] if staged else out), counts

    out, counts = gather(cols, widths, n)
    n_all = sum(counts)
    data = {c: gather([blob], [1], total) for c, (blob, total) in blobs.items()}
    torch.cuda.synchronize()
    types = (ColType * nc)(*[table.coltype(c) for c in range(nc)])
    names = (C.c_char_p * nc)(*[table.col_name(c).encode() for c in range(nc)])
    data_bytes = (C.c_int64 * nc)(*[(sum(data[c][1]) if c in data else 0) for c in range(nc)])
    narrow = 1 if any(table.col_width(c) == 8 and table.coltype(c).type == capi.T_DECIMAL128 for c in range(nc)) else 0
    h = C.c_void_p()
    check(ctx.lib.ldb_gpu_table_alloc(ctx.h, name.encode(), nc, types, names, n_all, data_bytes, narrow, C.byref(h)))
    res = Table(ctx, h)
    keep = []
    for c in range(nc):
        values, offsets, _, _ = res.col_ptrs(c)
        if c in data:
            lens = out[c].view(torch.int64) if n_all else torch.zeros(0, dtype=torch.int64, device="cuda")
            off = torch.zeros(n_all + 1, dtype=torch.int64, device="cuda")
            off[1:] = torch.cumsum(lens, 0)
            keep.append(off)
            torch.cuda.synchronize()
            _d2d(ctx, offsets, off.data_ptr(), 8 * (n_all + 1))
            _d2d(ctx, values, data[c][0][0].data_ptr(), sum(data[c][1]))
        else:
            _d2d(ctx, values, out[c].data_ptr(), n_all * res.col_width(c))
    for k, c in enumerate(nullable):
        check(ctx.lib.ldb_gpu_table_set_validity_bytes(ctx.h, res.h, c, C.c_void_p(out[nc + k].data_ptr())))
    ctx.sync()
    return res


def _plan_partitioned(ctx, name, world, *tables):
    """a plan piece that ends in ldb_gpu_partition: returns (table with the rows grouped by
    destination rank, rows per destination)"""
    t = C.c_void_p()
    counts = (C.c_int64 * world)()
    check_plan(getattr(capi.host_lib(), name)(ctx.h, *[x.h for x in tables], world, C.byref(t), counts))
    return Table(ctx, t), [int(c) for c in counts]


def shuffle(runner, table, send_counts, name):
    """the all-to-all of the hash-radix shuffle: `table` holds send_counts[j] rows for
    rank j, in rank order (ldb_gpu_partition's layout); returns the rows this rank receives from
    every peer.  One grouped point-to-point exchange for the whole table (one message per peer pair
    carrying all columns; each peer pair has its own xGMI link); fixed-width columns only."""
    if getattr(runner, "comm", None) is not None:  # RCCL inside the library (ldb_gpu_alltoall)
        return runner.comm.alltoall(table, send_counts, name)
    cols, widths = table_to_tensors(runner.ctx, table)
    staged = runner.dist.get_backend() == "gloo"
    if staged:
        cols = [c.cpu() for c in cols]
    out, recv_counts = ldist.alltoall_columns(runner.dist, cols, widths, send_counts)
    if staged:
        out = [c.cuda() for c in out]
    torch.cuda.synchronize()
    return tensors_to_table(runner.ctx, table, out, sum(int(c) for c in recv_counts), name)


def run_query(runner, q):
    ctx, db = runner.ctx, runner.db
    if q == 1:
        part = _plan(ctx, "ldb_plan_tpch_q1_partial", db.lineitem)
        allp = replicate(runner, part, "q1_partials")
        return _plan(ctx, "ldb_plan_tpch_q1_final", allp)
    if q == 6:
        part = ctx.plan_q6(db.lineitem)
        # a shard where nothing passes yields a NULL sum: exchange it as 0 (SUM identity)
        allp = replicate(runner, part, "q6_partials")
        return _plan(ctx, "ldb_plan_tpch_q6_final", allp)
    if q == 3:
        keys = _plan(ctx, "ldb_plan_tpch_q3_customers", db.customer)
        allkeys = replicate(runner, keys, "q3_custkeys")
        top = _plan(ctx, "ldb_plan_tpch_q3_local", allkeys, db.orders, db.lineitem)
        tops = replicate(runner, top, "q3_tops")
        return _plan(ctx, "ldb_plan_tpch_q3_final", tops)
    if q == 4:  # orders and their lineitems are co-located: local plan, merge the per-rank counts
        part = ctx.plan_q4(db.orders, db.lineitem)
        return _plan(ctx, "ldb_plan_tpch_q4_final", replicate(runner, part, "q4_partials"))
    if q == 12:
        part = ctx.plan_q12(db.orders, db.lineitem)
        return _plan(ctx, "ldb_plan_tpch_q12_final", replicate(runner, part, "q12_partials"))
    if q == 18:
        top = _plan(ctx, "ldb_plan_tpch_q18_local", db.orders, db.lineitem)
        top100 = _plan(ctx, "ldb_plan_tpch_q18_mid", replicate(runner, top, "q18_tops"))
        named = _plan(ctx, "ldb_plan_tpch_q18_names", top100, db.customer)  # customers are sharded by rows
        allnamed = replicate(runner, named, "q18_named")
        if os.environ.get("LDB_DIST_DEBUG"):
            print(f"[rank {runner.dist.get_rank()}] named={named.to_arrow().to_pylist()}\n   gathered={allnamed.to_arrow().to_pylist()}", flush=True)
        return _plan(ctx, "ldb_plan_tpch_q18_final", allnamed)
    if q == 5:  # the two reduced dimension tables are all-gathered, the rest is shard-local
        custs = replicate(runner, _plan(ctx, "ldb_plan_tpch_q5_customers", db.customer, db.nation, db.region), "q5_customers")
        supps = replicate(runner, _plan(ctx, "ldb_plan_tpch_q5_suppliers", db.supplier, db.nation, db.region), "q5_suppliers")
        part = _plan(ctx, "ldb_plan_tpch_q5_local", custs, supps, db.orders, db.lineitem)
        return _plan(ctx, "ldb_plan_tpch_q5_final", replicate(runner, part, "q5_partials"), db.nation)
    if q == 7:  # like Q5: all-gather the customers / suppliers of the two nations, then shard-local
        custs = replicate(runner, _plan(ctx, "ldb_plan_tpch_q7_customers", db.customer, db.nation), "q7_customers")
        supps = replicate(runner, _plan(ctx, "ldb_plan_tpch_q7_suppliers", db.supplier, db.nation), "q7_suppliers")
        part = _plan(ctx, "ldb_plan_tpch_q7_local", custs, supps, db.orders, db.lineitem)
        return _plan(ctx, "ldb_plan_tpch_q7_final", replicate(runner, part, "q7_partials"), db.nation)
    if q == 9:
        world = runner.world
        if "supplier_all" not in runner.cache:  # a static dimension table: replicated once
            runner.cache["supplier_all"] = replicate(runner, db.supplier, "supplier_all")
        green = replicate(runner, _plan(ctx, "ldb_plan_tpch_q9_green", db.part), "q9_green")
        lside, lcounts = _plan_partitioned(ctx, "ldb_plan_tpch_q9_lineitem_side", world, green, db.lineitem, db.orders)
        pside, pcounts = _plan_partitioned(ctx, "ldb_plan_tpch_q9_partsupp_side", world, green, db.partsupp)
        lrows = shuffle(runner, lside, lcounts, "q9_lrows")  # co-partition both sides on the part key
        psrows = shuffle(runner, pside, pcounts, "q9_psrows")
        part = _plan(ctx, "ldb_plan_tpch_q9_join", lrows, psrows, runner.cache["supplier_all"], db.nation)
        return _plan(ctx, "ldb_plan_tpch_q9_final", replicate(runner, part, "q9_partials"), db.nation)
    if q == 11:
        # partsupp is sharded by rows, so a part's rows may straddle two shards: the shard-local
        # groups are re-partitioned on the hash of ps_partkey (the high-cardinality group-by exchange
        # of SURVEY §8(e)) and merged; the scalar subquery's total is the sum of the ranks' totals
        supps = replicate(runner, _plan(ctx, "ldb_plan_tpch_q11_suppliers", db.supplier, db.nation), "q11_suppliers")
        local = _plan(ctx, "ldb_plan_tpch_q11_groups", supps, db.partsupp)
        parts, counts = _plan_partitioned(ctx, "ldb_plan_tpch_q11_partition", runner.world, local)
        groups = _plan(ctx, "ldb_plan_tpch_q11_merge", shuffle(runner, parts, counts, "q11_rows"))
        totals = replicate(runner, _plan(ctx, "ldb_plan_tpch_q11_total", groups), "q11_totals")
        kept = _plan(ctx, "ldb_plan_tpch_q11_filter", groups, totals)
        return _plan(ctx, "ldb_plan_tpch_q11_sort", replicate(runner, kept, "q11_kept"))
    if q == 10:
        # orders and their lineitems are co-located but a customer's orders are spread over the
        # shards: the shard-local (o_custkey, revenue) groups are re-partitioned on the hash of the
        # key and merged (Q11's exchange); every rank's 20 best merged groups are all-gathered, the
        # global 20 looked up in the row-sharded customer table, and the gathered rows ordered
        local = _plan(ctx, "ldb_plan_tpch_q10_local", db.orders, db.lineitem)
        parts, counts = _plan_partitioned(ctx, "ldb_plan_tpch_q10_partition", runner.world, local)
        groups = _plan(ctx, "ldb_plan_tpch_q10_merge", shuffle(runner, parts, counts, "q10_rows"))
        tops = replicate(runner, _plan(ctx, "ldb_plan_tpch_q10_top", groups), "q10_tops")
        top20 = _plan(ctx, "ldb_plan_tpch_q10_top", tops)
        named = _plan(ctx, "ldb_plan_tpch_q10_names", top20, db.customer, db.nation)
        return _plan(ctx, "ldb_plan_tpch_q10_final", replicate(runner, named, "q10_named"))
    if q == 15:
        # a supplier's lineitems are spread over the shards: exchange + merge of the groups as in Q10;
        # the maximum of the ranks' best groups is the view's maximum; every rank keeps its groups
        # that reach it, the gathered winners are joined with the (once replicated) supplier table
        if "supplier_all" not in runner.cache:
            runner.cache["supplier_all"] = replicate(runner, db.supplier, "supplier_all")
        local = _plan(ctx, "ldb_plan_tpch_q15_local", db.lineitem)
        parts, counts = _plan_partitioned(ctx, "ldb_plan_tpch_q15_partition", runner.world, local)
        groups = _plan(ctx, "ldb_plan_tpch_q15_merge", shuffle(runner, parts, counts, "q15_rows"))
        best = _plan(ctx, "ldb_plan_tpch_q15_max", replicate(runner, _plan(ctx, "ldb_plan_tpch_q15_max", groups), "q15_best"))
        winners = replicate(runner, _plan(ctx, "ldb_plan_tpch_q15_winners", groups, best), "q15_winners")
        return _plan(ctx, "ldb_plan_tpch_q15_final", winners, runner.cache["supplier_all"])
    if q == 14:
        # lineitem is sharded by orders, part by rows: the PROMO part keys (1/6 of part) are all-gathered
        # per query, the part key column (the inner join's build side) once; the two partial sums are
        # added before the ratio is formed
        if "part_keys_all" not in runner.cache:
            kc = [c for c in range(db.part.n_cols) if db.part.col_name(c) == "p_partkey"][0]
            runner.cache["part_keys_all"] = replicate(runner, db.part.rel().materialize([(0, kc)]), "part_keys_all")
        promo = replicate(runner, _plan(ctx, "ldb_plan_tpch_q14_promo", db.part), "q14_promo")
        part = _plan(ctx, "ldb_plan_tpch_q14_local", promo, runner.cache["part_keys_all"], db.lineitem)
        return _plan(ctx, "ldb_plan_tpch_q14_final", replicate(runner, part, "q14_partials"))
    if q == 8:
        # the part keys of the type (1/150 of part) and the region's customers are all-gathered, the
        # supplier table once; lineitem and orders are co-located, so the joins and the partial sums
        # per year are shard-local
        if "supplier_all" not in runner.cache:
            runner.cache["supplier_all"] = replicate(runner, db.supplier, "supplier_all")
        parts = replicate(runner, _plan(ctx, "ldb_plan_tpch_q8_parts", db.part), "q8_parts")
        custs = replicate(runner, _plan(ctx, "ldb_plan_tpch_q8_customers", db.customer, db.nation, db.region), "q8_customers")
        part = _plan(ctx, "ldb_plan_tpch_q8_local", parts, custs, runner.cache["supplier_all"], db.orders, db.lineitem, db.nation)
        return _plan(ctx, "ldb_plan_tpch_q8_final", replicate(runner, part, "q8_partials"))
    raise ValueError(f"TPC-H Q{q} has no multi-GPU plan yet")
