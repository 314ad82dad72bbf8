// ldb_setops.hip — the remaining sub-operator state types of SURVEY §8(f).4 on the device:
//   * set operations          UnionAllLowering / UnionDistinctLowering / CountingSetOperationLowering
//                             (src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp:622-930): a map keyed by ALL
//                             columns with one counter per input, then per key 1 row (distinct) or
//                             min(c1, c2) / max(c1 - c2, 0) rows (INTERSECT ALL / EXCEPT ALL);
//   * window functions        WindowLowering (:2193-2553) over a partitioned, sorted "continuous view":
//                             frames are ROW offsets clamped to the partition (OffsetReferenceByLowering,
//                             SubOpToControlFlow.cpp:3860-3885), rank = entries between the frame begin and the
//                             current row + 1 (RankWindowFunc :2043-2058), aggregates over the frame through a
//                             SegmentTreeView (include/lingodb/runtime/SegmentTreeView.h:11-43,
//                             src/runtime/SegmentTreeView.cpp:18-95).
// (The third piece — outer joins that also keep the BUILD side's unmatched rows, the reference's HashMultiMap with
//  per-entry markers, RelAlgToSubOp.cpp:1217-1294 — lives with the other join kinds in ldb_join.hip.)
//
// MI355X design.  Set operations reuse the group-by operator: both inputs are concatenated with a side flag, ONE
// hash aggregation over all columns carries the two conditional counters (NULL = NULL, as the reference's `isa`
// compare block), the multiplicity of every group is computed on the device, scanned, and the result rows are
// produced by a balanced binary search over the scan (no per-group loops, no host round trip per group).
// The segment tree is the implicit array form (node k covers nodes 2k and 2k + 1, leaves at [n, 2n)): built level by
// level with plain stores — no pointers, no recursion — and queried bottom-up in O(log n) per row by one lane; one
// tree over the whole sorted input serves every partition because frames are clamped to their partition.
#include "ldb_internal.h"
#include "ldb_keys.h"
#include <climits>
#include <memory>
#include <vector>

int32_t ldb_rel_select(ldb_ctx* ctx, ldb_rel* in, uint32_t* sel, int64_t n_sel, ldb_rel** out);

namespace {
struct TableGuard {
   ldb_ctx* ctx;
   ldb_table* t = nullptr;
   explicit TableGuard(ldb_ctx* c) : ctx(c) {}
   ~TableGuard() {
      if (t) ldb_gpu_table_release(ctx, t);
   }
   ldb_table* release() {
      ldb_table* r = t;
      t = nullptr;
      return r;
   }
};
struct RelGuard {
   ldb_ctx* ctx;
   ldb_rel* r = nullptr;
   explicit RelGuard(ldb_ctx* c) : ctx(c) {}
   ~RelGuard() {
      if (r) ldb_gpu_rel_release(ctx, r);
   }
};
struct Bufs {
   ldb_ctx* ctx;
   std::vector<void*> ptrs;
   explicit Bufs(ldb_ctx* c) : ctx(c) {}
   ~Bufs() {
      for (void* p : ptrs) ldb_dev_free(ctx, p);
   }
   template <typename T>
   int32_t alloc(T** out, size_t bytes) {
      void* p = nullptr;
      LDB_TRY(ldb_dev_alloc(ctx, &p, bytes ? bytes : 8));
      ptrs.push_back(p);
      *out = (T*) p;
      return LDB_OK;
   }
   void forget(void* p) { // ownership moved elsewhere
      for (auto& q : ptrs)
         if (q == p) q = nullptr;
   }
};
} // namespace

// ================================================================== concatenation of two tables (same schema)
__global__ void k_concat_offsets(const int64_t* __restrict__ a, uint64_t na, const int64_t* __restrict__ b, uint64_t nb, int64_t* __restrict__ out) {
   const uint64_t n = na + nb;
   const int64_t a0 = na ? a[0] : 0, bytes_a = na ? a[na] - a0 : 0, b0 = nb ? b[0] : 0;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i <= n; i += (uint64_t) gridDim.x * blockDim.x)
      out[i] = i <= na ? (na ? a[i] - a0 : 0) : bytes_a + (b[i - na] - b0);
}
__global__ void k_concat_validity(const uint8_t* __restrict__ a, uint64_t na, const uint8_t* __restrict__ b, uint64_t nb, uint8_t* __restrict__ out) {
   const uint64_t n = na + nb, n_bytes = (n + 7) / 8;
   for (uint64_t w = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; w < n_bytes; w += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++) {
         const uint64_t i = w * 8 + k;
         if (i >= n) break;
         bool v;
         if (i < na) v = a ? (a[i >> 3] >> (i & 7)) & 1 : true;
         else v = b ? (b[(i - na) >> 3] >> ((i - na) & 7)) & 1 : true;
         if (v) m |= (uint8_t) (1u << k);
      }
      out[w] = m;
   }
}
__global__ void k_fill_side(int32_t* __restrict__ out, uint64_t na, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = i < na ? 0 : 1;
}

// rows of a followed by the rows of b (+ an int32 `side` column, 0 / 1, when with_side); schema = a's
static int32_t table_concat(ldb_ctx* ctx, const ldb_table* a, const ldb_table* b, bool with_side, const char* name, ldb_table** out) {
   const int nc = (int) a->cols.size();
   if ((int) b->cols.size() != nc) LDB_FAIL(LDB_ERR_INVALID, "set_op: the inputs have %d and %zu columns", nc, b->cols.size());
   const int64_t na = a->n_rows, nb = b->n_rows, n = na + nb;
   if (n >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "set_op: %ld rows exceed uint32 row ids", (long) n);
   for (const ldb_table* t : {a, b}) // the concatenation copies string bytes: dictionary-coded (lazy) columns are written out first
      for (auto& c : t->cols)
         if (ldb_column_is_lazy(c)) LDB_TRY(ldb_column_strings(ctx, c, t->n_rows));
   std::vector<ldb_coltype> types;
   std::vector<const char*> names;
   std::vector<int64_t> data_bytes;
   for (int k = 0; k < nc; k++) {
      const ldb_column &ca = a->cols[(size_t) k], &cb = b->cols[(size_t) k];
      if (ca.type.type != cb.type.type || ca.width != cb.width || (ca.type.type == LDB_T_DECIMAL128 && (ca.type.precision != cb.type.precision || ca.type.scale != cb.type.scale)))
         LDB_FAIL(LDB_ERR_INVALID, "set_op: column %d has different types on the two sides (cast to a common type first, as the frontend does)", k);
      types.push_back(ca.type);
      names.push_back(ca.name.c_str());
      data_bytes.push_back(ca.type.type == LDB_T_UTF8 ? ca.value_bytes + cb.value_bytes : 0);
   }
   if (with_side) {
      types.push_back({LDB_T_INT32, 0, 0, 0});
      names.push_back("set_side");
      data_bytes.push_back(0);
   }
   TableGuard res(ctx);
   LDB_TRY(ldb_gpu_table_alloc(ctx, name, (int32_t) types.size(), types.data(), names.data(), n, data_bytes.data(), 0, &res.t));
   const int grid = ldb_grid_for(ctx, n + 1, 256, 8);
   for (int k = 0; k < nc; k++) {
      const ldb_column &ca = a->cols[(size_t) k], &cb = b->cols[(size_t) k];
      ldb_column& dst = res.t->cols[(size_t) k];
      if (ca.type.type == LDB_T_UTF8) {
         // (a table's string bytes start at offsets[0], which is 0 for every table this library builds; a sliced import keeps its base)
         int64_t a_first = 0, a_bytes = 0, b_first = 0, b_bytes = 0;
         if (na) {
            int64_t e[2];
            LDB_TRY(LDB_READBACK(ctx, &e[0], ca.offsets, 8));
            LDB_TRY(LDB_READBACK(ctx, &e[1], ca.offsets + na, 8));
            a_first = e[0], a_bytes = e[1] - e[0];
         }
         if (nb) {
            int64_t e[2];
            LDB_TRY(LDB_READBACK(ctx, &e[0], cb.offsets, 8));
            LDB_TRY(LDB_READBACK(ctx, &e[1], cb.offsets + nb, 8));
            b_first = e[0], b_bytes = e[1] - e[0];
         }
         if (a_bytes) LDB_HIP(hipMemcpyAsync(dst.values, (const char*) ca.values + a_first, (size_t) a_bytes, hipMemcpyDeviceToDevice, ctx->stream));
         if (b_bytes) LDB_HIP(hipMemcpyAsync((char*) dst.values + a_bytes, (const char*) cb.values + b_first, (size_t) b_bytes, hipMemcpyDeviceToDevice, ctx->stream));
         hipLaunchKernelGGL(k_concat_offsets, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t*) ca.offsets, (uint64_t) na, (const int64_t*) cb.offsets, (uint64_t) nb, dst.offsets);
         dst.value_bytes = a_bytes + b_bytes;
      } else {
         const size_t w = (size_t) ca.width;
         if (dst.width != ca.width) { // (narrowed decimal inputs: keep the source width)
            ldb_dev_free(ctx, dst.values);
            dst.values = nullptr;
            dst.width = ca.width;
            dst.value_bytes = n * (int64_t) w;
            LDB_TRY(ldb_dev_alloc(ctx, &dst.values, (size_t) (dst.value_bytes ? dst.value_bytes : 8)));
         }
         if (na) LDB_HIP(hipMemcpyAsync(dst.values, ca.values, (size_t) na * w, hipMemcpyDeviceToDevice, ctx->stream));
         if (nb) LDB_HIP(hipMemcpyAsync((char*) dst.values + (size_t) na * w, cb.values, (size_t) nb * w, hipMemcpyDeviceToDevice, ctx->stream));
      }
      if (ca.validity || cb.validity) {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &dst.validity, (size_t) ((n + 7) / 8 + 1)));
         hipLaunchKernelGGL(k_concat_validity, dim3(ldb_grid_for(ctx, (n + 7) / 8, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) ca.validity, (uint64_t) na,
                            (const uint8_t*) cb.validity, (uint64_t) nb, dst.validity);
         dst.null_count = ca.null_count + cb.null_count;
         dst.type.nullable = 1;
      }
   }
   if (with_side && n) hipLaunchKernelGGL(k_fill_side, dim3(grid), dim3(256), 0, ctx->stream, (int32_t*) res.t->cols[(size_t) nc].values, (uint64_t) na, (uint64_t) n);
   LDB_HIP(hipGetLastError());
   *out = res.release();
   return LDB_OK;
}

// ================================================================== set operations
// rows every group contributes to the result (CountingSetOperationLowering, RelAlgToSubOp.cpp:870-905)
__global__ void k_setop_multiplicity(const int64_t* __restrict__ c1, const int64_t* __restrict__ c2, uint64_t n, int op, uint32_t* __restrict__ m) {
   for (uint64_t g = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; g < n; g += (uint64_t) gridDim.x * blockDim.x) {
      const int64_t l = c1[g], r = c2[g];
      int64_t k;
      switch (op) {
         case LDB_SET_UNION: k = 1; break;
         case LDB_SET_INTERSECT: k = (l > 0 && r > 0) ? 1 : 0; break;
         case LDB_SET_EXCEPT: k = (l > 0 && r == 0) ? 1 : 0; break;
         case LDB_SET_INTERSECT_ALL: k = l > r ? r : l; break;
         default: k = l - r < 0 ? 0 : l - r; break; // EXCEPT ALL
      }
      m[g] = (uint32_t) k;
   }
}
// output row i belongs to the group g with off[g] <= i < off[g + 1]: binary search, every lane one row
__global__ void k_expand_groups(const uint32_t* __restrict__ off, uint64_t n_groups, uint64_t total, uint32_t* __restrict__ out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < total; i += (uint64_t) gridDim.x * blockDim.x) {
      uint64_t lo = 0, hi = n_groups; // largest g with off[g] <= i
      while (hi - lo > 1) {
         const uint64_t mid = (lo + hi) >> 1;
         if ((uint64_t) off[mid] <= i) lo = mid;
         else hi = mid;
      }
      out[i] = (uint32_t) lo;
   }
}

extern "C" int32_t ldb_gpu_set_op(ldb_ctx* ctx, ldb_rel* left, const ldb_colref* left_cols, ldb_rel* right, const ldb_colref* right_cols, int32_t n_cols, int32_t op, ldb_table** out) {
   if (!ctx || !left || !right || !left_cols || !right_cols || !out || n_cols < 1 || n_cols > LDB_MAX_KEYS) LDB_FAIL(LDB_ERR_INVALID, "set_op: bad argument (1..%d columns)", LDB_MAX_KEYS);
   if (op < LDB_SET_UNION_ALL || op > LDB_SET_EXCEPT_ALL) LDB_FAIL(LDB_ERR_INVALID, "set_op: bad operation %d", op);
   TableGuard a(ctx), b(ctx), u(ctx), g(ctx);
   LDB_TRY(ldb_gpu_materialize(ctx, left, left_cols, n_cols, &a.t));
   LDB_TRY(ldb_gpu_materialize(ctx, right, right_cols, n_cols, &b.t));
   LDB_TRY(table_concat(ctx, a.t, b.t, op != LDB_SET_UNION_ALL, "set_op_input", &u.t));
   if (op == LDB_SET_UNION_ALL) { // subop.union of the two streams: nothing else to do
      *out = u.release();
      return LDB_OK;
   }
   // ONE aggregation over all columns with a counter per input side (the map with two i64 counters)
   RelGuard ur(ctx);
   LDB_TRY(ldb_gpu_rel_from_table(ctx, u.t, &ur.r));
   std::vector<ldb_colref> keys;
   for (int32_t k = 0; k < n_cols; k++) keys.push_back({0, k});
   ldb_agg_spec aggs[2];
   memset(aggs, 0, sizeof(aggs));
   for (int s = 0; s < 2; s++) {
      aggs[s].fn = LDB_AGG_COUNT_STAR;
      aggs[s].out_type = LDB_T_INT64;
      aggs[s].n_preds = 1;
      aggs[s].preds[0].col = {0, n_cols};
      aggs[s].preds[0].op = LDB_F_EQ;
      aggs[s].preds[0].rhs_kind = LDB_RHS_INT;
      aggs[s].preds[0].value_lo = (uint64_t) s;
   }
   LDB_TRY(ldb_gpu_groupby(ctx, ur.r, nullptr, 0, keys.data(), n_cols, aggs, 2, std::max<int64_t>(16, u.t->n_rows), &g.t));
   const int64_t ng = g.t->n_rows;
   Bufs bufs(ctx);
   uint32_t *mult, *off, *sel;
   LDB_TRY(bufs.alloc(&mult, 4 * (size_t) (ng + 1)));
   LDB_TRY(bufs.alloc(&off, 4 * (size_t) (ng + 1)));
   uint64_t* d_total;
   LDB_TRY(ldb_counters(ctx, 1, &d_total));
   uint64_t total = 0;
   if (ng) {
      hipLaunchKernelGGL(k_setop_multiplicity, dim3(ldb_grid_for(ctx, ng, 256, 8)), dim3(256), 0, ctx->stream, (const int64_t*) g.t->cols[(size_t) n_cols].values,
                         (const int64_t*) g.t->cols[(size_t) n_cols + 1].values, (uint64_t) ng, op, mult);
      LDB_TRY(ldb_exclusive_scan_u32(ctx, mult, off, ng, d_total));
      LDB_TRY(ldb_read_u64(ctx, d_total, &total));
   }
   if (total >= (uint64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "set_op: %llu result rows exceed uint32 row ids", (unsigned long long) total);
   LDB_TRY(bufs.alloc(&sel, 4 * (size_t) (total ? total : 1)));
   if (total) hipLaunchKernelGGL(k_expand_groups, dim3(ldb_grid_for(ctx, (int64_t) total, 256, 8)), dim3(256), 0, ctx->stream, (const uint32_t*) off, (uint64_t) ng, total, sel);
   LDB_HIP(hipGetLastError());
   RelGuard gr(ctx), picked(ctx);
   LDB_TRY(ldb_gpu_rel_from_table(ctx, g.t, &gr.r));
   bufs.forget(sel); // ldb_rel_select takes the selection vector over
   LDB_TRY(ldb_rel_select(ctx, gr.r, sel, (int64_t) total, &picked.r));
   LDB_TRY(ldb_gpu_materialize(ctx, picked.r, keys.data(), n_cols, out));
   return LDB_OK;
}

// ================================================================== window functions
#define LDB_WIN_MAX_FNS 8
struct DWindow {
   uint64_t n;
   int64_t from, to; // frame offsets in rows; INT64_MIN / INT64_MAX = unbounded
   uint64_t seg; // const uint32_t*: partition number of every sorted row
   uint64_t starts; // const uint32_t*: first sorted row of every partition (+ one sentinel = n)
   int32_t n_fns;
   int32_t pad;
   int32_t fn[LDB_WIN_MAX_FNS];
   DCol col[LDB_WIN_MAX_FNS]; // argument column read through the SORTED relation
   uint64_t tree_val[LDB_WIN_MAX_FNS]; // i128[2n]: the implicit segment tree of that function's states
   uint64_t tree_ok[LDB_WIN_MAX_FNS]; // uint8_t[2n]: state is non-NULL
   uint64_t out_val[LDB_WIN_MAX_FNS]; // result column values
   uint64_t out_ok[LDB_WIN_MAX_FNS]; // uint8_t[n] (one byte per row; packed afterwards) or 0
   int32_t out_width[LDB_WIN_MAX_FNS];
};
__device__ __forceinline__ void d_win_combine(int fn, i128& v, bool& ok, i128 v2, bool ok2) {
   // SumAggrFunc / MinAggrFunc / MaxAggrFunc / CountAggrFunc::combine (RelAlgToSubOp.cpp:1843-2027): a NULL state is the
   // identity, the result is NULL only when both are
   if (!ok2) return;
   if (!ok) {
      v = v2;
      ok = true;
      return;
   }
   switch (fn) {
      case LDB_WIN_MIN: v = v2 < v ? v2 : v; break;
      case LDB_WIN_MAX: v = v2 > v ? v2 : v; break;
      default: v = (i128) ((u128) v + (u128) v2); break; // SUM, COUNT (wrapping like the generated code)
   }
}
__global__ void k_win_heads(DKeys pk, uint64_t n, uint32_t* __restrict__ head) {
   const KV keys(pk);
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      head[i] = (i == 0 || (pk.n_keys > 0 && !d_keys_equal(keys, i - 1, keys, i, true))) ? 1u : 0u; // PARTITION BY groups NULLs together
}
__global__ void k_win_segments(const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ seg, uint32_t* __restrict__ starts, uint32_t n_seg) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t s = pos[i] + head[i] - 1u;
      seg[i] = s;
      if (head[i]) starts[s] = (uint32_t) i;
      if (i == 0) starts[n_seg] = (uint32_t) n;
   }
}
// leaves: createInitialStateFn(entry) of every function (SegmentTreeView::buildRecursively, leaf case)
__global__ void k_segtree_leaves(const DWindow* __restrict__ d) {
   const uint64_t n = d->n;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      for (int f = 0; f < d->n_fns; f++) {
         const int fn = d->fn[f];
         if (fn == LDB_WIN_RANK || fn == LDB_WIN_COUNT_STAR) continue; // need no tree
         const CV c(d->col[f]);
         const uint32_t row = d_phys_row(c, i);
         const bool ok = d_valid(c, row);
         i128 v = 0;
         if (fn == LDB_WIN_COUNT) v = ok ? 1 : 0;
         else if (ok) v = d_load_i128(c, row);
         gptr_mut<i128>(d->tree_val[f])[n + i] = v;
         gptr_mut<uint8_t>(d->tree_ok[f])[n + i] = (fn == LDB_WIN_COUNT || ok) ? 1 : 0;
      }
   }
}
// inner nodes [lo, hi): combineStatesFn(left, right); a level only reads nodes of larger index, built in earlier launches
__global__ void k_segtree_level(const DWindow* __restrict__ d, uint64_t lo, uint64_t hi) {
   for (uint64_t k = lo + blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; k < hi; k += (uint64_t) gridDim.x * blockDim.x) {
      for (int f = 0; f < d->n_fns; f++) {
         const int fn = d->fn[f];
         if (fn == LDB_WIN_RANK || fn == LDB_WIN_COUNT_STAR) continue;
         i128* tv = gptr_mut<i128>(d->tree_val[f]);
         uint8_t* to = gptr_mut<uint8_t>(d->tree_ok[f]);
         i128 v = tv[2 * k];
         bool ok = to[2 * k] != 0;
         d_win_combine(fn, v, ok, tv[2 * k + 1], to[2 * k + 1] != 0);
         tv[k] = v;
         to[k] = ok ? 1 : 0;
      }
   }
}
__device__ __forceinline__ void d_win_store(const DWindow* __restrict__ d, int f, uint64_t i, i128 v, bool ok) {
   if (d->out_ok[f]) gptr_mut<uint8_t>(d->out_ok[f])[i] = ok ? 1 : 0;
   switch (d->out_width[f]) {
      case 16: gptr_mut<i128>(d->out_val[f])[i] = v; break;
      case 8: gptr_mut<int64_t>(d->out_val[f])[i] = (int64_t) v; break;
      default: gptr_mut<int32_t>(d->out_val[f])[i] = (int32_t) v; break;
   }
}
// per sorted row: the frame [lo, hi] (both inclusive, clamped to the row's partition) and SegmentTreeView::lookup(lo, hi)
__global__ void k_win_lookup(const DWindow* __restrict__ d) {
   const uint64_t n = d->n;
   const uint32_t* seg = gptr<uint32_t>(d->seg);
   const uint32_t* starts = gptr<uint32_t>(d->starts);
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t s = seg[i];
      const int64_t st = starts[s], last = (int64_t) starts[s + 1] - 1, cur = (int64_t) i;
      // OffsetReferenceBy: max(0, cur + off) then min(len - 1, …) inside the partition's buffer; unbounded = begin / end reference
      int64_t lo = d->from == INT64_MIN ? st : (d->from == 0 ? cur : cur + d->from);
      int64_t hi = d->to == INT64_MAX ? last : (d->to == 0 ? cur : cur + d->to);
      lo = lo < st ? st : (lo > last ? last : lo);
      hi = hi < st ? st : (hi > last ? last : hi);
      for (int f = 0; f < d->n_fns; f++) {
         const int fn = d->fn[f];
         if (fn == LDB_WIN_RANK) { // entries between the frame begin and the current row, + 1
            d_win_store(d, f, i, (i128) (cur - lo + 1), true);
            continue;
         }
         if (fn == LDB_WIN_COUNT_STAR) {
            d_win_store(d, f, i, (i128) (hi >= lo ? hi - lo + 1 : 0), true);
            continue;
         }
         const i128* tv = gptr<i128>(d->tree_val[f]);
         const uint8_t* to = gptr<uint8_t>(d->tree_ok[f]);
         i128 v = 0;
         bool ok = false;
         if (hi >= lo) { // (the reference throws "from must be <= to"; an inverted frame yields the empty state here)
            uint64_t l = (uint64_t) lo + n, r = (uint64_t) hi + n + 1;
            while (l < r) {
               if (l & 1) {
                  d_win_combine(fn, v, ok, tv[l], to[l] != 0);
                  l++;
               }
               if (r & 1) {
                  r--;
                  d_win_combine(fn, v, ok, tv[r], to[r] != 0);
               }
               l >>= 1;
               r >>= 1;
            }
         }
         d_win_store(d, f, i, v, fn == LDB_WIN_COUNT ? true : ok);
      }
   }
}
__global__ void k_win_pack_validity(const uint8_t* __restrict__ bytes, uint64_t n, uint8_t* __restrict__ bitmap, unsigned long long* __restrict__ nulls) {
   const uint64_t nb = (n + 7) / 8;
   unsigned long long c = 0;
   for (uint64_t b = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; b < nb; b += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++)
         if (b * 8 + k < n) {
            if (bytes[b * 8 + k]) m |= (uint8_t) (1u << k);
            else c++;
         }
      bitmap[b] = m;
   }
   for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
   if ((threadIdx.x & 63) == 0 && c) atomicAdd(nulls, c);
}

__global__ void k_window_iota(uint32_t* out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = (uint32_t) i;
}
extern "C" int32_t ldb_gpu_window(ldb_ctx* ctx, ldb_rel* in, const ldb_colref* part_keys, int32_t n_part, const ldb_sort_spec* order, int32_t n_order, int64_t frame_from,
                                  int64_t frame_to, const ldb_window_fn* fns, int32_t n_fns, ldb_rel** out_rel, ldb_table** out_cols) {
   if (!ctx || !in || !out_rel || !out_cols || n_part < 0 || n_order < 0 || n_fns < 1 || n_fns > LDB_WIN_MAX_FNS || !fns || n_part > LDB_MAX_KEYS)
      LDB_FAIL(LDB_ERR_INVALID, "window: bad argument (1..%d functions, <= %d partition keys)", LDB_WIN_MAX_FNS, LDB_MAX_KEYS);
   LDB_TRY(ldb_rel_force(ctx, in));
   // 1. the continuous view: rows ordered by (partition keys, ORDER BY keys) — the reference hash-partitions and sorts every
   //    partition's buffer; one sort with the partition keys in front yields the same per-partition order
   std::vector<ldb_sort_spec> specs;
   for (int32_t k = 0; k < n_part; k++) specs.push_back({part_keys[k], 0, 0});
   for (int32_t k = 0; k < n_order; k++) specs.push_back(order[k]);
   RelGuard sorted(ctx);
   if (!specs.empty()) {
      LDB_TRY(ldb_gpu_sort(ctx, in, specs.data(), (int32_t) specs.size(), &sorted.r));
   } else { // no PARTITION BY, no ORDER BY: one partition in input order
      uint32_t* iota;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &iota, 4 * (size_t) (in->n_rows ? in->n_rows : 1)));
      if (in->n_rows) hipLaunchKernelGGL(k_window_iota, dim3(ldb_grid_for(ctx, in->n_rows, 256, 8)), dim3(256), 0, ctx->stream, iota, (uint64_t) in->n_rows);
      LDB_HIP(hipGetLastError());
      LDB_TRY(ldb_rel_select(ctx, in, iota, in->n_rows, &sorted.r));
   }
   const int64_t n = sorted.r->n_rows;
   // 2. result table (one column per function)
   std::vector<ldb_coltype> types((size_t) n_fns);
   std::vector<std::string> name_store((size_t) n_fns);
   std::vector<const char*> names((size_t) n_fns);
   auto d = std::make_unique<DWindow>();
   memset(d.get(), 0, sizeof(DWindow));
   d->n = (uint64_t) n;
   d->from = frame_from;
   d->to = frame_to;
   d->n_fns = n_fns;
   for (int32_t f = 0; f < n_fns; f++) {
      const int fn = fns[f].fn;
      if (fn < LDB_WIN_RANK || fn > LDB_WIN_COUNT_STAR) LDB_FAIL(LDB_ERR_INVALID, "window: unknown function %d", fn);
      d->fn[f] = fn;
      static const char* base[] = {"rank", "sum", "min", "max", "count", "count_star"};
      name_store[(size_t) f] = std::string(base[fn]) + "_" + std::to_string(f);
      names[(size_t) f] = name_store[(size_t) f].c_str();
      if (fn == LDB_WIN_RANK || fn == LDB_WIN_COUNT || fn == LDB_WIN_COUNT_STAR) {
         types[(size_t) f] = {LDB_T_INT64, 0, 0, 0};
      } else {
         LDB_TRY(ldb_make_dcol(sorted.r, fns[f].col, &d->col[f]));
         const DCol& c = d->col[f];
         switch (c.type) {
            case LDB_T_DECIMAL128: types[(size_t) f] = {LDB_T_DECIMAL128, c.precision, c.scale, 1}; break;
            case LDB_T_INT64: types[(size_t) f] = {LDB_T_INT64, 0, 0, 1}; break;
            case LDB_T_INT32: types[(size_t) f] = {fn == LDB_WIN_SUM ? LDB_T_INT64 : LDB_T_INT32, 0, 0, 1}; break; // (SUM over int32 accumulates in 64 bit here)
            case LDB_T_DATE32:
               if (fn == LDB_WIN_SUM) LDB_FAIL(LDB_ERR_INVALID, "window: SUM over a date column");
               types[(size_t) f] = {LDB_T_DATE32, 0, 0, 1};
               break;
            default: LDB_FAIL(LDB_ERR_UNSUPPORTED, "window: function %d over column type %d (integer / decimal / date columns)", fn, c.type);
         }
      }
      if (fn == LDB_WIN_COUNT) LDB_TRY(ldb_make_dcol(sorted.r, fns[f].col, &d->col[f]));
   }
   TableGuard res(ctx);
   LDB_TRY(ldb_gpu_table_alloc(ctx, "window", n_fns, types.data(), names.data(), n, nullptr, 0, &res.t));
   if (n == 0) {
      *out_rel = sorted.r;
      sorted.r = nullptr;
      *out_cols = res.release();
      return LDB_OK;
   }
   Bufs bufs(ctx);
   // 3. partitions: head flags → partition numbers and start rows
   uint32_t *head, *pos, *seg, *starts;
   LDB_TRY(bufs.alloc(&head, 4 * (size_t) n));
   LDB_TRY(bufs.alloc(&pos, 4 * (size_t) n));
   LDB_TRY(bufs.alloc(&seg, 4 * (size_t) n));
   auto pk = std::make_unique<DKeys>();
   memset(pk.get(), 0, sizeof(DKeys));
   if (n_part) LDB_TRY(ldb_make_dkeys(sorted.r, part_keys, n_part, pk.get()));
   const int grid = ldb_grid_for(ctx, n, 256, 8);
   hipLaunchKernelGGL(k_win_heads, dim3(grid), dim3(256), 0, ctx->stream, *pk, (uint64_t) n, head);
   uint64_t* d_total;
   LDB_TRY(ldb_counters(ctx, 1, &d_total));
   LDB_TRY(ldb_exclusive_scan_u32(ctx, head, pos, n, d_total));
   uint64_t n_seg = 0;
   LDB_TRY(ldb_read_u64(ctx, d_total, &n_seg));
   LDB_TRY(bufs.alloc(&starts, 4 * (size_t) (n_seg + 1)));
   hipLaunchKernelGGL(k_win_segments, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*) head, (const uint32_t*) pos, (uint64_t) n, seg, starts, (uint32_t) n_seg);
   d->seg = (uint64_t) seg;
   d->starts = (uint64_t) starts;
   // 4. the segment trees + result buffers
   std::vector<uint8_t*> ok_bytes((size_t) n_fns, nullptr);
   for (int32_t f = 0; f < n_fns; f++) {
      const int fn = d->fn[f];
      ldb_column& oc = res.t->cols[(size_t) f];
      d->out_val[f] = (uint64_t) oc.values;
      d->out_width[f] = oc.width;
      if (fn == LDB_WIN_SUM || fn == LDB_WIN_MIN || fn == LDB_WIN_MAX) {
         LDB_TRY(bufs.alloc(&ok_bytes[(size_t) f], (size_t) n));
         d->out_ok[f] = (uint64_t) ok_bytes[(size_t) f];
      }
      if (fn != LDB_WIN_RANK && fn != LDB_WIN_COUNT_STAR) {
         i128* tv;
         uint8_t* to;
         LDB_TRY(bufs.alloc(&tv, 16 * 2 * (size_t) n));
         LDB_TRY(bufs.alloc(&to, 2 * (size_t) n));
         d->tree_val[f] = (uint64_t) tv;
         d->tree_ok[f] = (uint64_t) to;
      }
   }
   LdbDesc<DWindow> dd_desc(ctx);
   LDB_TRY(dd_desc.upload(d.get(), sizeof(DWindow)));
   DWindow* dd = dd_desc.p;
   {
      LdbProf prof_(ctx, "k_segtree_build");
      hipLaunchKernelGGL(k_segtree_leaves, dim3(grid), dim3(256), 0, ctx->stream, (const DWindow*) dd);
      // inner nodes level by level: [ceil(n / 2), n), then [ceil(n / 4), ceil(n / 2)), … down to node 1
      for (uint64_t hi = (uint64_t) n; hi > 1;) {
         const uint64_t lo = (hi + 1) / 2;
         hipLaunchKernelGGL(k_segtree_level, dim3(ldb_grid_for(ctx, (int64_t) (hi - lo), 256, 8)), dim3(256), 0, ctx->stream, (const DWindow*) dd, lo, hi);
         hi = lo;
      }
   }
   {
      LdbProf prof_(ctx, "k_win_lookup");
      hipLaunchKernelGGL(k_win_lookup, dim3(grid), dim3(256), 0, ctx->stream, (const DWindow*) dd);
   }
   LDB_HIP(hipGetLastError());
   // 5. validity bitmaps of the nullable results (SUM / MIN / MAX over a frame that holds only NULLs)
   std::vector<int> vf;
   for (int32_t f = 0; f < n_fns; f++)
      if (ok_bytes[(size_t) f]) vf.push_back(f);
   if (!vf.empty()) {
      unsigned long long* d_nulls;
      LDB_TRY(bufs.alloc(&d_nulls, 8 * vf.size()));
      LDB_HIP(hipMemsetAsync(d_nulls, 0, 8 * vf.size(), ctx->stream));
      for (size_t v = 0; v < vf.size(); v++) {
         ldb_column& oc = res.t->cols[(size_t) vf[v]];
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &oc.validity, (size_t) ((n + 7) / 8 + 1)));
         hipLaunchKernelGGL(k_win_pack_validity, dim3(ldb_grid_for(ctx, (n + 7) / 8, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) ok_bytes[(size_t) vf[v]], (uint64_t) n, oc.validity,
                            d_nulls + v);
      }
      std::vector<unsigned long long> nulls(vf.size(), 0);
      LDB_TRY(LDB_READBACK(ctx, nulls.data(), d_nulls, 8 * vf.size()));
      for (size_t v = 0; v < vf.size(); v++) {
         ldb_column& oc = res.t->cols[(size_t) vf[v]];
         oc.null_count = (int64_t) nulls[v];
         if (!nulls[v]) { // no NULL after all: drop the bitmap
            ldb_dev_free(ctx, oc.validity);
            oc.validity = nullptr;
         }
      }
   }
   LDB_HIP(hipGetLastError());
   *out_rel = sorted.r;
   sorted.r = nullptr;
   *out_cols = res.release();
   return LDB_OK;
}
