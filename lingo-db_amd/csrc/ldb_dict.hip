// ldb_dict.hip — utf8 dictionary encoding at registration (second half).
//
// The reference stores char(n) / varchar columns as Arrow utf8 (src/runtime/storage/LingoDBTable.cpp:184-191) and
// evaluates string predicates, group keys and sort keys byte-wise on VarLen32 values (StringRuntime.cpp:242-256,
// LowerToStd.cpp:398-466).  On the GPU a 600 M-row `l_shipmode IN ('MAIL', 'SHIP')` then streams 7.4 GB of offsets and
// bytes to compare seven distinct values, and `GROUP BY p_brand, p_type` hashes and compares strings per row.
//
// MI355X design: a utf8 column with at most LDB_DICT_MAX (1024) distinct values gets, when its table is registered, an
// ORDER-PRESERVING dictionary BESIDE its strings (the Arrow buffers stay what they are — results, joins and the exchange
// still see utf8):
//     dict        the distinct non-NULL strings in bytewise ascending order (= std::string_view order), a device table;
//     dict_codes  one uint32 per row: the rank of the row's string in `dict` (0xFFFFFFFF for NULL).
// It is built with the library's own operators — GROUP BY the column (distinct values), ORDER BY (the dictionary order),
// a SINGLE join of every row against the sorted distinct values (the build row of the match IS the code) — so its
// semantics are the operators', not a second implementation.  Uses:
//   * a predicate `col OP constant` (=, <>, <, <=, >, >=, IN, LIKE, NOT LIKE) is evaluated ONCE on the dictionary's rows
//     with the ordinary scan kernel; the accepted codes become a 1024-bit set in the predicate descriptor and the row
//     predicate is a test of the row's 4-byte code (LDB_F_CODESET; a single accepted code becomes a plain `code = k`,
//     which takes the branch-free batched path).  Sets are cached per column and predicate text;
//   * GROUP BY / ORDER BY keys read the codes (ldb_make_dkeys_dict / ldb_make_dcol_dict): hashing and comparing 4 bytes
//     instead of strings, and because the dictionary preserves order, sorting codes sorts strings.  Output key columns
//     are still gathered from the strings;
//   * a column GATHERED from a dictionary-encoded one (ldb_gpu_materialize, the key columns of a group-by result) inherits
//     the dictionary: its codes are gathered alongside and the dictionary table is shared — the second-level GROUP BY and
//     the ORDER BY over Q16's 28 k result groups run on codes too.
// Joins, db.hash (ldb_gpu_hash_keys) and the exchange's hash partitioning keep the strings: their hashes must agree with
// the other side / the other ranks, whose dictionaries differ.
#include "ldb_internal.h"
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

void ldb_column_dict_release(ldb_ctx* ctx, ldb_column& c) {
   ldb_dev_free(ctx, c.dict_codes);
   c.dict_codes = nullptr;
   if (c.dict && --c.dict->dict_refs == 0) ldb_gpu_table_release(ctx, c.dict);
   c.dict = nullptr;
   c.dict_size = 0;
   delete c.dict_pred_cache;
   c.dict_pred_cache = nullptr;
}

namespace {
struct Rel {
   ldb_ctx* ctx;
   ldb_rel* r = nullptr;
   explicit Rel(ldb_ctx* c) : ctx(c) {}
   ~Rel() {
      if (r) ldb_gpu_rel_release(ctx, r);
   }
};
struct Tab {
   ldb_ctx* ctx;
   ldb_table* t = nullptr;
   explicit Tab(ldb_ctx* c) : ctx(c) {}
   ~Tab() {
      if (t) ldb_gpu_table_release(ctx, t);
   }
};
struct Ht {
   ldb_ctx* ctx;
   ldb_hashtable* h = nullptr;
   explicit Ht(ldb_ctx* c) : ctx(c) {}
   ~Ht() {
      if (h) ldb_gpu_hashtable_release(ctx, h);
   }
};
// a one-column view of the first `rows` rows of column `col` (buffers shared, nothing owned)
std::unique_ptr<ldb_table> view_of(const ldb_table* t, int32_t col, int64_t rows) {
   auto v = std::make_unique<ldb_table>();
   v->ctx = t->ctx;
   v->name = t->name + "#dictview";
   v->n_rows = rows;
   (void) ldb_column_strings(t->ctx, t->cols[(size_t) col], t->n_rows); // (a view shares the bytes: a lazy column writes them out first)
   v->cols.push_back(t->cols[(size_t) col]);
   ldb_column& c = v->cols[0];
   c.owned = false;
   c.dict_codes = nullptr;
   c.dict = nullptr;
   c.dict_size = 0;
   c.dict_pred_cache = nullptr;
   c.has_range = false;
   c.sorted_state = -1;
   return v;
}
// distinct non-NULL values of column 0 of `v` → *out (one utf8 column + the count); LDB_OK with *n_distinct > limit when there are too many
int32_t distinct_of(ldb_ctx* ctx, const ldb_table* v, int64_t limit, ldb_table** out, int64_t* n_distinct) {
   Rel r(ctx);
   LDB_TRY(ldb_gpu_rel_from_table(ctx, v, &r.r));
   ldb_filter_desc nn;
   memset(&nn, 0, sizeof(nn));
   nn.col = {0, 0};
   nn.op = LDB_F_NOTNULL;
   ldb_agg_spec cnt;
   memset(&cnt, 0, sizeof(cnt));
   cnt.fn = LDB_AGG_COUNT_STAR;
   cnt.out_type = LDB_T_INT64;
   const ldb_colref key = {0, 0};
   LDB_TRY(ldb_gpu_groupby(ctx, r.r, v->cols[0].validity ? &nn : nullptr, v->cols[0].validity ? 1 : 0, &key, 1, &cnt, 1, limit * 2 + 16, out));
   *n_distinct = (*out)->n_rows;
   return LDB_OK;
}
} // namespace

int32_t ldb_table_dict_encode(ldb_ctx* ctx, ldb_table* t, int32_t col, int32_t max_distinct) {
   if (!ctx || !t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "dict_encode: bad argument");
   ldb_column& c = t->cols[(size_t) col];
   if (c.type.type != LDB_T_UTF8 || c.dict_codes || t->n_rows == 0) return LDB_OK;
   if (max_distinct > LDB_DICT_MAX) max_distinct = LDB_DICT_MAX;
   // 1. cheap cardinality check on a prefix: a column of comments or names is dropped here, before anything of its size runs
   const int64_t sample = std::min<int64_t>(t->n_rows, 1 << 16);
   {
      auto v = view_of(t, col, sample);
      Tab d(ctx);
      int64_t nd = 0;
      LDB_TRY(distinct_of(ctx, v.get(), max_distinct, &d.t, &nd));
      if (nd > max_distinct || (sample < t->n_rows && nd * 2 > max_distinct && nd * 8 > sample)) return LDB_OK;
   }
   // 2. the distinct values of the whole column, in string order
   auto v = view_of(t, col, t->n_rows);
   Tab distinct(ctx), dict(ctx);
   int64_t nd = 0;
   LDB_TRY(distinct_of(ctx, v.get(), max_distinct, &distinct.t, &nd));
   if (nd > max_distinct || nd == 0) return LDB_OK;
   {
      Rel dr(ctx), sorted(ctx);
      LDB_TRY(ldb_gpu_rel_from_table(ctx, distinct.t, &dr.r));
      const ldb_sort_spec by = {{0, 0}, 0, 0};
      LDB_TRY(ldb_gpu_sort(ctx, dr.r, &by, 1, &sorted.r));
      const ldb_colref k = {0, 0};
      LDB_TRY(ldb_gpu_materialize(ctx, sorted.r, &k, 1, &dict.t));
   }
   // 3. every row's code = the dictionary row its string matches (SINGLE join: one output row per probe row, in order)
   Rel dictrel(ctx), rows(ctx), joined(ctx);
   Ht ht(ctx);
   LDB_TRY(ldb_gpu_rel_from_table(ctx, dict.t, &dictrel.r));
   LDB_TRY(ldb_gpu_rel_from_table(ctx, v.get(), &rows.r));
   const ldb_colref k = {0, 0};
   LDB_TRY(ldb_gpu_join_build(ctx, dictrel.r, &k, 1, 1, &ht.h));
   LDB_TRY(ldb_gpu_join_probe(ctx, ht.h, rows.r, &k, 1, LDB_JOIN_SINGLE, &joined.r, nullptr));
   if (joined.r->n_rows != t->n_rows || joined.r->sides.size() != 2 || !joined.r->sides[1].rowids) LDB_FAIL(LDB_ERR_INVALID, "dict_encode: unexpected join shape");
   uint32_t* codes = nullptr;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &codes, 4 * (size_t) t->n_rows));
   LDB_HIP(hipMemcpyAsync(codes, joined.r->sides[1].rowids, 4 * (size_t) t->n_rows, hipMemcpyDeviceToDevice, ctx->stream));
   LDB_HIP(hipStreamSynchronize(ctx->stream)); // (the views and relations above die with this scope)
   c.dict_codes = codes;
   c.dict = dict.t;
   dict.t = nullptr;
   c.dict_size = (int32_t) nd;
   c.dict_pred_cache = new std::unordered_map<std::string, std::string>();
   return LDB_OK;
}

int32_t ldb_table_dict_encode_all(ldb_ctx* ctx, ldb_table* t) {
   if (ldb_option("dict_encode", 1) == 0) return LDB_OK;
   if (t->n_rows < ldb_option("dict_min_rows", 4096)) return LDB_OK; // (a table this small gains nothing)
   for (size_t c = 0; c < t->cols.size(); c++)
      if (t->cols[c].type.type == LDB_T_UTF8) LDB_TRY(ldb_table_dict_encode(ctx, t, (int32_t) c, LDB_DICT_MAX));
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_table_dict_encode(ldb_ctx* ctx, ldb_table* t, int32_t col, int32_t* n_distinct) {
   if (!ctx || !t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "table_dict_encode: bad argument");
   LDB_TRY(ldb_table_dict_encode(ctx, t, col, LDB_DICT_MAX));
   if (n_distinct) *n_distinct = t->cols[(size_t) col].dict_codes ? t->cols[(size_t) col].dict_size : -1;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_dict_size(const ldb_table* t, int32_t col) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) return -1;
   return t->cols[(size_t) col].dict_codes ? t->cols[(size_t) col].dict_size : -1;
}

// ---------------------------------------------------------------- descriptors over the codes
static void code_dcol(const ldb_rel_side& s, const ldb_column& c, DCol* out) {
   memset(out, 0, sizeof(*out));
   out->values = (uint64_t) c.dict_codes;
   out->validity = (uint64_t) c.validity;
   out->rowids = (uint64_t) s.rowids;
   out->type = LDB_T_INT32;
   out->width = 4;
}
int32_t ldb_make_dcol_dict(const ldb_rel* r, ldb_colref ref, DCol* out) {
   if (ref.side >= 0 && (size_t) ref.side < r->sides.size()) { // the codes first: a lazy column (no bytes yet) must not be written out for a consumer that reads codes
      const ldb_rel_side& s = r->sides[(size_t) ref.side];
      if (ref.col >= 0 && (size_t) ref.col < s.table->cols.size()) {
         const ldb_column& c = s.table->cols[(size_t) ref.col];
         if (c.type.type == LDB_T_UTF8 && c.dict_codes) {
            code_dcol(s, c, out);
            return LDB_OK;
         }
      }
   }
   return ldb_make_dcol(r, ref, out);
}
int32_t ldb_make_dkeys_dict(const ldb_rel* r, const ldb_colref* keys, int32_t n_keys, DKeys* out) {
   if (n_keys < 0 || n_keys > LDB_MAX_KEYS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "%d key columns (max %d)", n_keys, LDB_MAX_KEYS);
   memset(out, 0, sizeof(*out));
   out->n_keys = n_keys;
   for (int32_t k = 0; k < n_keys; k++) LDB_TRY(ldb_make_dcol_dict(r, keys[k], &out->cols[k]));
   return LDB_OK;
}

// `col OP constant(s)` over a dictionary-encoded utf8 column → the set of accepted codes (evaluated on the dictionary
// with the ordinary scan) → a predicate on the code column
int32_t ldb_dict_rewrite_pred(const ldb_rel* r, const ldb_filter_desc* p, DPred* out, bool* done) {
   *done = false;
   const ldb_rel_side& s = r->sides[(size_t) p->col.side];
   const ldb_column& c = s.table->cols[(size_t) p->col.col];
   if (!c.dict_codes || !c.dict || c.type.type != LDB_T_UTF8 || p->rhs_kind != LDB_RHS_STRING || ldb_option("dict_encode", 1) == 0) return LDB_OK;
   ldb_ctx* ctx = r->ctx;
   // cache key: the predicate's own bytes
   std::string key;
   key.append((const char*) &p->op, sizeof(p->op));
   if (p->op == LDB_F_IN) {
      for (int k = 0; k < p->n_in; k++) {
         key.append((const char*) &p->in_str_lens[k], 4);
         key.append(p->in_strs[k], (size_t) p->in_str_lens[k]);
      }
   } else {
      key.append(p->str, (size_t) (p->str_len > 0 ? p->str_len : 0));
   }
   std::string bits;
   auto it = c.dict_pred_cache->find(key);
   if (it != c.dict_pred_cache->end()) {
      bits = it->second;
   } else {
      // the same predicate over the dictionary's rows: accepted dictionary rows = accepted codes
      ldb_filter_desc q = *p;
      q.col = {0, 0};
      Rel dr(ctx), hit(ctx);
      LDB_TRY(ldb_gpu_rel_from_table(ctx, c.dict, &dr.r));
      LDB_TRY(ldb_gpu_scan_filter(ctx, dr.r, &q, 1, &hit.r));
      LDB_TRY(ldb_rel_force(ctx, hit.r));
      const int64_t n = hit.r->n_rows;
      std::vector<uint32_t> ids((size_t) (n ? n : 1));
      if (n) LDB_TRY(ldb_gpu_rel_read_rowids(ctx, hit.r, 0, ids.data(), n));
      bits.assign(LDB_DICT_MAX / 8, '\0');
      for (int64_t i = 0; i < n; i++) bits[ids[(size_t) i] >> 3] = (char) ((uint8_t) bits[ids[(size_t) i] >> 3] | (1u << (ids[(size_t) i] & 7)));
      (*c.dict_pred_cache)[key] = bits;
   }
   memset(out, 0, sizeof(*out));
   code_dcol(s, c, &out->col);
   int64_t count = 0, only = -1;
   for (int32_t k = 0; k < c.dict_size; k++)
      if (((uint8_t) bits[(size_t) k >> 3] >> (k & 7)) & 1) {
         count++;
         only = k;
      }
   if (count == 1) { // one accepted string: a plain equality on the code (the branch-free batched path)
      out->op = LDB_F_EQ;
      out->rhs_kind = LDB_RHS_INT;
      out->lo = (uint64_t) only;
      out->hi = 0;
   } else if (count == 0) { // nothing accepted: no valid row carries code -1
      out->op = LDB_F_EQ;
      out->rhs_kind = LDB_RHS_INT;
      out->lo = ~0ull;
      out->hi = -1;
   } else {
      out->op = LDB_F_CODESET;
      out->rhs_kind = LDB_RHS_CODESET;
      static_assert(sizeof(out->in_blob) >= LDB_DICT_MAX / 8, "the code set lives in DPred::in_blob");
      memcpy(out->in_blob, bits.data(), LDB_DICT_MAX / 8);
   }
   *done = true;
   return LDB_OK;
}
