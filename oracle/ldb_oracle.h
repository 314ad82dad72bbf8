/*
 * ldb_oracle.h — CPU restatement of LingoDB's sub-operator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by, or executed
 * from the product (liblingodb_gpu.so, lingo-db_amd/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker / reported baseline.
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * lingo-db/lingo-db checkout).  Parity pinning: the hash functions are checked against the
 * reference's known-answer vectors (test/lit/DB/hash.mlir:27-34,
 * test/unittests/storage/TestStorage.cpp:289) in tests/test_oracle_golden.py; the operator
 * restatements are cross-checked against the reference's own runtime objects compiled from
 * /root/reference (oracle/_ref, see oracle/ref_build/) where those compile offline.
 * Long-string hashing (XXH64, seed 0 = llvm::xxHash64 of LLVM 20.1) has no absolute
 * known-answer in the reference ("parity unpinned" for that one function; it is checked
 * against the xxhash package's published test vectors instead).
 *
 * Descriptors (ldb_filter_desc, ldb_expr, ldb_agg_spec, ldb_sort_spec) are shared with the
 * C-ABI so the same test input drives both sides.
 */
#ifndef LDB_ORACLE_H
#define LDB_ORACLE_H
#include "../include/lingodb_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
   int32_t type; /* ldb_type */
   int32_t precision, scale;
   int32_t width; /* bytes per value as stored: 16 (or 8 when narrowed) for decimals */
   const void* values;
   const int64_t* offsets; /* utf8: int64[n+1] */
   const uint8_t* validity; /* Arrow validity bitmap or NULL = all valid */
} ora_col;

typedef struct {
   int64_t n_rows;
   int32_t n_cols;
   const ora_col* cols;
} ora_table;

typedef struct {
   int64_t n_rows;
   int32_t n_sides;
   const ora_table* tables[LDB_MAX_SIDES];
   const uint32_t* rowids[LDB_MAX_SIDES]; /* NULL = identity */
} ora_rel;

/* ---- scalar spec (a5) */
uint64_t ora_hash64(int64_t v);
uint64_t ora_hash_combine(uint64_t h_new, uint64_t total);
uint64_t ora_xxh64(const void* data, uint64_t len, uint64_t seed);
void ora_varlen32_image(const uint8_t* p, uint32_t len, uint8_t out16[16]);
uint64_t ora_hash_varlen(const uint8_t* p, uint32_t len);
uint64_t ora_hash_i128(uint64_t lo, int64_t hi, int first, uint64_t total);
uint16_t ora_bloom_mask(uint32_t idx);

/* ---- operators.  `threads` = worker count (morsel size 20 000 rows as the reference). */
int64_t ora_scan_filter(const ora_rel* in, const ldb_filter_desc* preds, int32_t n_preds, uint32_t* out_rows, int32_t threads);
void ora_hash_keys(const ora_rel* in, const ldb_colref* keys, int32_t n_keys, uint64_t* out);
/* returns #groups; rep_rows[g] = logical row of `in` holding the group's key values;
 * vals[g*n_aggs + a] = aggregate as 128-bit integer (lo,hi) or, for is_float args, the f64 bits in lo;
 * valid[g*n_aggs+a] = 0 when the aggregate is NULL (MIN/MAX/SUM over no non-null input). */
int64_t ora_groupby(const ora_rel* in, const ldb_filter_desc* preds, int32_t n_preds, const ldb_colref* keys, int32_t n_keys,
                    const ldb_agg_spec* aggs, int32_t n_aggs, int32_t threads, uint32_t* rep_rows, int64_t* vals_lohi,
                    uint8_t* valid, int64_t cap_groups);
/* returns #output rows (may exceed cap: then only cap rows were written) */
int64_t ora_join(const ora_rel* build, const ldb_colref* bkeys, const ora_rel* probe, const ldb_colref* pkeys, int32_t n_keys,
                 int32_t kind, int32_t threads, uint32_t* out_probe, uint32_t* out_build, uint8_t* out_mark, int64_t cap);
void ora_sort(const ora_rel* in, const ldb_sort_spec* specs, int32_t n_specs, uint32_t* out_perm);
int64_t ora_topk(const ora_rel* in, const ldb_sort_spec* specs, int32_t n_specs, int64_t k, uint32_t* out_perm);
/* evaluate an expression for every row (tests of a16): out = lo/hi pairs */
void ora_eval_expr(const ora_rel* in, const ldb_expr* e, int64_t* out_lohi);
/* partition id per row: (hash >> 16) % nparts */
void ora_partition_ids(const ora_rel* in, const ldb_colref* keys, int32_t n_keys, int32_t nparts, int32_t* out);
/* SQL LIKE (StringRuntime::like, escape '\\') and extract(year from date32) (DateRuntime::extractYear) */
int32_t ora_like(const uint8_t* s, int64_t sl, const uint8_t* p, int64_t pl);
int64_t ora_extract_year(int64_t days);
/* substring(str from `from` for `len`) (StringRuntime::substr): byte range [begin, end) of the result inside str */
void ora_substr(const uint8_t* s, int64_t sl, int64_t from, int64_t len, int64_t* out_begin, int64_t* out_end);
int32_t ora_decimal_muldiv(const int64_t num[2], const int64_t mul[2], int32_t mul_div_pow10, int32_t pow10, const int64_t den[2], int64_t out[2]);
/* ---- §8(f).4 state types
 * SegmentTreeView (src/runtime/SegmentTreeView.cpp:18-79): recursive build over entries {value, valid}, lookup(from, to)
 * inclusive; fn = ldb_window_fn_kind SUM / MIN / MAX / COUNT with the generated combine functions' NULL handling
 * (RelAlgToSubOp.cpp:1843-2027).  128-bit values as (lo, hi) pairs. */
int32_t ora_segment_tree(const int64_t* vals_lohi, const uint8_t* valid, int64_t n, int32_t fn, const int64_t* from, const int64_t* to, int64_t nq, int64_t* out_lohi, uint8_t* out_valid);
/* WindowLowering (RelAlgToSubOp.cpp:2193-2553) over rows already in window order: part_start[i] / part_end[i] = first row and
 * one-past-last row of row i's partition; frame offsets as in ldb_gpu_window (INT64_MIN / INT64_MAX = unbounded), both ends
 * clamped into the partition (OffsetReferenceByLowering, SubOpToControlFlow.cpp:3860-3885); RANK = current - frame begin + 1 */
int32_t ora_window(const int64_t* vals_lohi, const uint8_t* valid, const int64_t* part_start, const int64_t* part_end, int64_t n, int32_t fn, int64_t frame_from, int64_t frame_to,
                   int64_t* out_lohi, uint8_t* out_valid);
/* CountingSetOperationLowering (RelAlgToSubOp.cpp:735-930): rows of a result group from the two per-input counters */
int64_t ora_setop_multiplicity(int32_t op, int64_t c_left, int64_t c_right);
int32_t ora_num_cores(void);

#ifdef __cplusplus
}
#endif
#endif
