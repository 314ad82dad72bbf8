/*
 * lingodb_gpu.h — C-ABI of liblingodb_gpu.so, the MI355X (gfx950) relational-operator
 * runtime that replaces LingoDB's CPU sub-operator hot path.
 *
 * Granularity: one call per `subop.execution_step` building block (the unit handled by
 * handleExecutionStepCPU, reference src/compiler/Conversion/SubOpToControlFlow/
 * SubOpToControlFlow.cpp:4363).  The reference exposes its runtime to JIT'd code as
 * Itanium-mangled C++ methods taking host function pointers (eq/combine/cmp callbacks,
 * include/lingodb/runtime/ headers); host callbacks cannot cross to the device, so every
 * callback is replaced by a declarative descriptor (ldb_filter_desc, ldb_expr,
 * ldb_agg_spec, ldb_sort_spec) that the device kernels evaluate.
 *
 * Conventions
 *   - every function returns LDB_OK (0) or a negative ldb_status; the message is
 *     available per-thread from ldb_gpu_last_error().  No C++ exception crosses the ABI.
 *     (Reference behaviour being replaced: std::runtime_error / assert / exit(1),
 *     src/execution/Execution.cpp:252-271.)
 *   - handles are opaque; the library owns all device memory until the matching
 *     *_release / ldb_gpu_ctx_destroy (reference: ExecutionContext::registerState,
 *     include/lingodb/runtime/ExecutionContext.h:111).
 *   - all work of one ctx is issued on ONE HIP stream (own, or caller supplied);
 *     calls are asynchronous unless they return host-visible counts.
 *   - row ids are uint32 (≤ 4 294 967 294 rows per GPU-resident table fragment).
 *   - plain pointers and sizes only; no torch / Arrow C++ types in any signature.
 */
#ifndef LINGODB_GPU_H
#define LINGODB_GPU_H

#ifndef LDB_NO_STD_HEADERS /* the run-time kernel specialiser (hiprtc) supplies its own typedefs */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Arrow C Data Interface (public, stable ABI; https://arrow.apache.org/docs/format/CDataInterface.html) */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
   const char* format;
   const char* name;
   const char* metadata;
   int64_t flags;
   int64_t n_children;
   struct ArrowSchema** children;
   struct ArrowSchema* dictionary;
   void (*release)(struct ArrowSchema*);
   void* private_data;
};
struct ArrowArray {
   int64_t length;
   int64_t null_count;
   int64_t offset;
   int64_t n_buffers;
   int64_t n_children;
   const void** buffers;
   struct ArrowArray** children;
   struct ArrowArray* dictionary;
   void (*release)(struct ArrowArray*);
   void* private_data;
};
#endif

typedef struct ldb_ctx ldb_ctx;
typedef struct ldb_table ldb_table; /* device-resident columnar table (one buffer per column) */
typedef struct ldb_rel ldb_rel; /* late-materialised relation: n rows over 1..LDB_MAX_SIDES tables */
typedef struct ldb_hashtable ldb_hashtable; /* join hash table (replaces HashIndexedView) */

typedef enum {
   LDB_OK = 0,
   LDB_ERR_INVALID = -1, /* bad argument / descriptor */
   LDB_ERR_UNSUPPORTED = -2, /* legal in the reference but not implemented on the device yet */
   LDB_ERR_OOM = -3,
   LDB_ERR_HIP = -4, /* HIP runtime error; text in ldb_gpu_last_error() */
   LDB_ERR_NO_DEVICE = -5,
   LDB_ERR_RETRY = -6 /* inside a replaying trace only: a recorded count no longer holds — end the trace, run again */
} ldb_status;

/* Physical column types = the Arrow physical types LingoDB stores
 * (toPhysicalType, reference src/runtime/storage/LingoDBTable.cpp:122-195). */
typedef enum {
   LDB_T_INT8 = 0,
   LDB_T_INT16 = 1,
   LDB_T_INT32 = 2,
   LDB_T_INT64 = 3,
   LDB_T_DATE32 = 4, /* days since epoch; hashed/compared as ns where the reference does */
   LDB_T_DECIMAL128 = 5, /* 16 B little-endian two's complement, (precision, scale) */
   LDB_T_CHAR4 = 6, /* fixed_size_binary(4) = char(1) */
   LDB_T_UTF8 = 7, /* int64 offsets on device + bytes */
   LDB_T_FLOAT64 = 8,
   LDB_T_FLOAT32 = 9,
   LDB_T_BOOL8 = 10 /* one byte per value (device-side results only) */
} ldb_type;

typedef struct {
   int32_t type; /* ldb_type */
   int32_t precision; /* decimals */
   int32_t scale;
   int32_t nullable;
} ldb_coltype;

/* Column reference inside a relation: table `side` of the relation, column index in it. */
typedef struct {
   int32_t side;
   int32_t col;
} ldb_colref;

#define LDB_MAX_SIDES 6
#define LDB_NULL_ROW 0xFFFFFFFFu /* build-side row id of an unmatched outer-join row */

/* ------------------------------------------------------------------ context */
/* Replaces ExecutionContext + scheduler worker set (reference ExecutionContext.h:62-127,
 * scheduler/Scheduler.h:29-42).  `stream` = an existing hipStream_t to issue on, or NULL
 * for a private stream. */
int32_t ldb_gpu_ctx_create(int32_t device_id, void* stream, ldb_ctx** out);
int32_t ldb_gpu_ctx_destroy(ldb_ctx* ctx);
int32_t ldb_gpu_ctx_sync(ldb_ctx* ctx);
const char* ldb_gpu_last_error(void);
/* device properties: name (gfx950 expected), CUs, HBM bytes free/total */
int32_t ldb_gpu_device_info(ldb_ctx* ctx, char* name, int32_t name_cap, int32_t* cus, int64_t* hbm_free, int64_t* hbm_total);

/* HIP-event timing on the ctx stream (bench.py must time on the stream the kernels run on). */
int32_t ldb_gpu_timer_create(ldb_ctx* ctx, int32_t* timer_id);
int32_t ldb_gpu_timer_start(ldb_ctx* ctx, int32_t timer_id);
int32_t ldb_gpu_timer_stop(ldb_ctx* ctx, int32_t timer_id);
int32_t ldb_gpu_timer_elapsed_ms(ldb_ctx* ctx, int32_t timer_id, float* ms); /* syncs on the stop event */

/* Per-kernel timing (Tracer-event equivalent, reference include/lingodb/utility/Tracer.h:13-166):
 * when enabled, the dominant kernels of each operator (k_groupby, k_scan_bitmap, k_join_build,
 * k_join_probe_*, …) are bracketed by HIP events on the ctx stream.  prof_get syncs, folds the
 * finished launches into per-name totals and returns launches and total milliseconds. */
int32_t ldb_gpu_prof_enable(ldb_ctx* ctx, int32_t on);
int32_t ldb_gpu_prof_reset(ldb_ctx* ctx);
int32_t ldb_gpu_prof_get(ldb_ctx* ctx, const char* kernel_name, int64_t* launches, double* total_ms);
/* the longest single launch of that kernel since the last reset (the roofline of an operator that launches one
 * kernel on inputs of very different size is priced on its LARGEST launch, not on the average) */
int32_t ldb_gpu_prof_get_max(ldb_ctx* ctx, const char* kernel_name, double* max_ms);
/* names of all kernels seen so far, '\n'-separated, into buf */
int32_t ldb_gpu_prof_names(ldb_ctx* ctx, char* buf, int32_t cap);
/* Trace marker: launches the empty kernel `k_ldb_marker` with a grid of `id` (1 … 65535)
 * workgroups on the ctx stream.  A `rocprofv3 --kernel-trace` of a run that brackets each query with
 * markers can be cut into per-query timelines by the marker's grid size (tools/timeline_summary.py). */
int32_t ldb_gpu_prof_marker(ldb_ctx* ctx, int32_t id);

/* Run-time kernel specialisation (the role LLVM JIT plays in the reference,
 * src/execution/LLVMBackends.cpp:219-406): kernels compiled / cache hits / compile time so far,
 * and a device-less compile check of the specialiser (hiprtc log into `log`). */
int32_t ldb_gpu_jit_stats(int64_t* compiled, int64_t* cache_hits, double* compile_ms);
int32_t ldb_gpu_jit_compile_check(char* log, int32_t cap);
/* Specialisations compile on worker threads (option jit_async, default 1; jit_threads): the operator that asked launches its generic
 * ahead-of-time kernel meanwhile — the reference's answer to compile latency is a baseline backend that emits code in milliseconds with an
 * optimising one behind it (include/lingodb/execution/Execution.h:103-104, src/execution/baseline/).  Code objects are kept on disk under a
 * content hash ($LDB_JIT_CACHE_DIR, default ~/.cache/ldb_jit/<arch>/<hash>.co; option jit_disk_cache = 0 disables), so a second process start
 * compiles nothing.  ldb_gpu_jit_wait blocks until nothing is queued or compiling (timeout_ms < 0: no limit; *pending = still outstanding) — what a
 * benchmark calls between its first execution and its timed region.  ldb_gpu_jit_info: vals[0..9) = compiled here, in-memory hits, disk hits,
 * disk writes, outstanding, failed, calls answered "still compiling", worker threads, code objects taken over from another process that was
 * compiling the same shape (the ranks of a multi-GPU run share the work through claims in the disk cache; option jit_share_compiles). */
int32_t ldb_gpu_jit_wait(int64_t timeout_ms, int64_t* pending);
int32_t ldb_gpu_jit_info(int64_t* vals, int32_t n);
/* stop the compile workers: queued specialisations are dropped, running ones finish.  Call before the process exits (the library also registers an
 * exit handler, but hiprtc's own statics may be torn down first when a compilation is still in flight at exit) */
int32_t ldb_gpu_jit_shutdown(void);
/* device-less self-test of the above (CPU test suite): asynchronous request → wait → code object on disk → second request answered from the disk */
int32_t ldb_gpu_jit_cache_selftest(char* log, int32_t cap);

/* Process-wide tuning options (each also readable from the environment as LDB_<NAME> on first use):
 *   jit (0/1), jit_min_rows      — run-time kernel specialisation and its row threshold (default 4 M)
 *   lazy_filter (0/1), lazy_min_rows — filters fused into the consuming kernel (default >= 1 M rows)
 *   join_ordered, join_chained, join_radix, join_radix_min_rows, join_radix_min_table_bytes, join_radix_part_bytes, probe_batch
 *   join_direct, join_rank, join_coarse (0/1) — the direct / rank-bitmap / LDS coarse-bitmap table layouts (default on)
 *   gb_ordered, gb_sorted, gb_direct, gb_partition (0/1), gb_partition_min_rows (8 M), gb_wgs_per_cu (0 = automatic)
 *   gb_dense_out (0/1)   — sorted keys: groups inside one wave leave the kernel as final rows (round 5, default on)
 *   gb_partition_values (0/1) — any aggregates over partitioned LDS slots, not only COUNT(*) (round 5, default on)
 *   join_pair32 (0/1)    — two 4-byte keys: the key values live in the slot (round 5, default on)
 *   plan_replay (0/1), desc_cache (0/1), desc_cache_mb — prepared plans: replay of the read-back trace, descriptor cache
 *   (environment only: LDB_HOST_TRACE=<ms> reports host calls of the library that took longer, with their site, and names a replayed
 *   read-back that differs from its record; libldb_host.so: LDB_PLAN_STEP_TRACE=1 prints the host time of every plan step)
 *   dict_encode (0/1), dict_min_rows — utf8 dictionary encoding at registration
 *   zone_maps (0/1), zone_min_rows (1 M) — zone maps of selective integer-like columns
 *   comm_transport (0 = RCCL, 1 = shared memory), comm_timeout_ms — the exchange
 *   debug_check (0/1) — range check of every row id a probe produces
 * Tests use it to drive ONE process through both the generic and the specialised / fused code
 * paths (the reference has the same kind of switch: LINGODB_EXECUTION_MODE, Execution.cpp:224-228). */
int32_t ldb_gpu_set_option(const char* name, int64_t value);
int64_t ldb_gpu_get_option(const char* name); /* -1 when never set and no default was read yet */
int64_t ldb_gpu_option_epoch(void); /* number of ldb_gpu_set_option calls so far (part of a prepared plan's validity) */

/* ------------------------------------------------------------------ read-back traces: executing a plan without returning to the host
 * Reference shape: one compiled pipeline runs scan → … → materialise without the query driver in between
 * (ScanRefsTableLowering, src/compiler/Conversion/SubOpToControlFlow/SubOpToControlFlow.cpp:1123-1202), and a query is timed
 * as ONE main() (src/execution/LLVMBackends.cpp:856-865).  At this ABI's granularity every operator whose output size
 * decides the next allocation reads a count back, i.e. waits for the device.  A trace removes those waits for a plan that is
 * executed again over unchanged inputs: bracket each execution with _begin / _end.  The first execution RECORDS every count
 * the operators read back (in call order); a later one REPLAYS them — the operator calls return at once with the recorded
 * counts while the real ones are only queued into a pinned log, so the host issues the whole plan ahead of the device — and
 * ldb_gpu_trace_end waits once and compares log and record.  Counts are pure functions of (plan, data, options): the caller
 * keys a trace by those (ldb_gpu_table_stamp of every input, ldb_gpu_option_epoch) and passes allow_replay = 0 when the key
 * changed.  Should a replayed count still differ (status LDB_TRACE_MISSED, or any call returning LDB_ERR_RETRY), everything
 * the execution produced is void: release it and execute again (the trace records afresh).  An execution that takes another
 * path than the recorded one (a column statistic is cached by now) verifies what it replayed and records from there on.
 * libldb_host.so's ldb_plan_prepare / ldb_plan_execute do all of this. */
typedef struct ldb_trace ldb_trace;
typedef enum { LDB_TRACE_OFF = 0, LDB_TRACE_RECORDED = 1, LDB_TRACE_REPLAYED = 2, LDB_TRACE_MISSED = 3 } ldb_trace_status;
int32_t ldb_gpu_trace_create(ldb_ctx* ctx, ldb_trace** out);
int32_t ldb_gpu_trace_destroy(ldb_ctx* ctx, ldb_trace* t);
/* allow_replay: bit 0 = replay if the trace holds a complete record; LDB_TRACE_COLLECTIVE = the plan exchanges rows with other ranks
 * (ldb_gpu_allgather / _shuffle): all ranks must pass the same bit 0 (agree with ldb_gpu_comm_agree on the minimum of
 * ldb_gpu_trace_replayable first), the exchanges' metadata reads are replayed like any count — the transfers of a replayed plan are queued
 * with the recorded sizes on every rank, no rank waits for another's counts — and a rank whose replayed count turns out wrong runs on over
 * the recorded sizes (its peers are queued against them) and reports LDB_TRACE_MISSED at the end; the ranks then agree on the minimum of
 * their verdicts and all of them repeat the execution recording. */
#define LDB_TRACE_COLLECTIVE 2
int32_t ldb_gpu_trace_begin(ldb_ctx* ctx, ldb_trace* t, int32_t allow_replay);
int32_t ldb_gpu_trace_replayable(const ldb_trace* t); /* 1 = _begin with bit 0 would replay */
int32_t ldb_gpu_trace_end(ldb_ctx* ctx, int32_t* status); /* synchronises the stream; *status = ldb_trace_status */
int32_t ldb_gpu_trace_stats(const ldb_trace* t, int64_t* entries, int64_t* records, int64_t* replays, int64_t* misses);
/* process-wide: replays that failed on a value which depends on the order a kernel's atomics ran in (the open-addressing build's
 * "long probe run" flag, ldb_join.hip) rather than on the data — 0 unless a build's key distribution sits on the run-length threshold */
int64_t ldb_gpu_order_dependent_misses(void);
/* descriptor cache of the context (descriptors of a repeated plan are byte-identical: uploaded once) */
int32_t ldb_gpu_desc_cache_stats(ldb_ctx* ctx, int64_t* hits, int64_t* misses, int64_t* bytes);
/* references operators hold on cached descriptors right now (0 between operator calls: every holder gives its reference back on every
 * path, so an unused entry can always be evicted) and the number of references ever given back that nobody held (0: a caller's bug) */
int32_t ldb_gpu_desc_cache_held(ldb_ctx* ctx, int64_t* held, int64_t* underflows);

/* ------------------------------------------------------------------ tables (a1) */
/* Replaces LingoDBTable::ensureLoaded + TableChunk flattening (LingoDBTable.cpp:27-54,
 * 200-225).  `schema` is a struct schema (format "+s"); each batch a struct array whose
 * children's buffers are exactly ArrayView.buffers[0..2].  Batches are concatenated per
 * column into one device buffer.  narrow_decimals = 1 stores decimal128(p<19) as int64
 * on the device (the width the generated code truncates to anyway, LowerToStd.cpp:128-132);
 * narrow_decimals = 2 (round 6, a compressed resident format) stores every such decimal at the narrowest of
 * 1 / 2 / 4 / 8 bytes the column's value range allows and a char(1) column of ASCII letters at one byte:
 * kernels widen in registers and compute in i64 / i128 as before — results are bit-identical — and
 * ldb_gpu_export widens back to the Arrow widths. */
int32_t ldb_gpu_table_register(ldb_ctx* ctx, const char* name, struct ArrowSchema* schema,
                               struct ArrowArray** batches, int64_t n_batches, int32_t narrow_decimals,
                               ldb_table** out);
/* Arrow IPC FILE → device table, inside the library (LingoDBTable::ensureLoaded, LingoDBTable.cpp:27-54: one
 * `<table>.arrow` file per table, all record batches).  The file is memory-mapped and its footer / schema / record-batch
 * flatbuffers are read without libarrow; every batch is registered from the mapping (ldb_gpu_table_register semantics,
 * incl. narrow_decimals and dictionary encoding).  Flat int8…int64 / float / decimal128 / date32 / utf8 / large_utf8 /
 * fixed_size_binary columns, uncompressed, little-endian, no dictionary batches; anything else → LDB_ERR_UNSUPPORTED naming
 * the column; a truncated or inconsistent file → LDB_ERR_INVALID (every offset is bounds-checked).
 * ldb_gpu_ipc_describe: the parse alone, no device needed — JSON {"columns":[{"name","format","nullable"}],"batches":[rows…],"rows"}. */
int32_t ldb_gpu_table_load_ipc(ldb_ctx* ctx, const char* name, const char* path, int32_t narrow_decimals, ldb_table** out);
int32_t ldb_gpu_ipc_describe(const char* path, char* out, int64_t cap);
/* Zone map of a column: min / max per 16 384 physical rows of a NOT NULL integer-like column, built on the
 * first column-vs-constant comparison over it (one pass, cached with the table) and KEPT only when the zones are selective
 * (sorted / clustered data: together they cover less than half of zones × value range).  Scan, fused-filter and probe
 * kernels then fail the rows of a zone that cannot satisfy the comparison without loading the column (options `zone_maps`,
 * `zone_min_rows`).  Returns the number of zones in use, 0 = none (wrong type, NULLs, small table, unselective), -1 = error. */
int64_t ldb_gpu_table_zones(ldb_ctx* ctx, const ldb_table* t, int32_t col);
/* Allocate an uninitialised device table (generator / shuffle receive side).
 * utf8 columns: data_bytes[i] = byte capacity of column i (ignored for fixed width). */
int32_t ldb_gpu_table_alloc(ldb_ctx* ctx, const char* name, int32_t n_cols, const ldb_coltype* types,
                            const char* const* col_names, int64_t n_rows, const int64_t* data_bytes,
                            int32_t narrow_decimals, ldb_table** out);
int32_t ldb_gpu_table_release(ldb_ctx* ctx, ldb_table* t); /* == evict */
int64_t ldb_gpu_table_rows(const ldb_table* t);
/* identity of the table's content: changes whenever the table is re-created, overwritten (ldb_gpu_table_write_fixed) or
 * resized (ldb_gpu_table_set_rows); never reused inside a process */
uint64_t ldb_gpu_table_stamp(const ldb_table* t);
int32_t ldb_gpu_table_cols(const ldb_table* t);
int32_t ldb_gpu_table_coltype(const ldb_table* t, int32_t col, ldb_coltype* out);
int32_t ldb_gpu_table_col_index(const ldb_table* t, const char* name); /* -1 if absent */
const char* ldb_gpu_table_col_name(const ldb_table* t, int32_t col);
/* rename a column (result tables: group-by names its aggregates agg0, agg1, …) */
int32_t ldb_gpu_table_rename_col(ldb_table* t, int32_t col, const char* name);
/* width in bytes of one value as resident on the device (8 for narrowed decimals) */
int32_t ldb_gpu_table_col_width(const ldb_table* t, int32_t col);
/* utf8 dictionary encoding (the reference stores char(n) / varchar as utf8, LingoDBTable.cpp:184-191): a
 * column with at most 1024 distinct strings gets an ORDER-PRESERVING dictionary beside its strings — codes follow the
 * bytewise string order of StringRuntime.cpp:242-256 — so predicates with constants test 4-byte codes and GROUP BY / ORDER
 * BY keys hash and compare codes; results, joins and the exchange still see the strings.  ldb_gpu_table_register and the
 * generator call it for every utf8 column (option `dict_encode`, tables of at least `dict_min_rows` rows); an explicit
 * call encodes a column of a smaller table.  *n_distinct / the return of _dict_size: dictionary entries, -1 = not encoded. */
int32_t ldb_gpu_table_dict_encode(ldb_ctx* ctx, ldb_table* t, int32_t col, int32_t* n_distinct);
int32_t ldb_gpu_table_dict_size(const ldb_table* t, int32_t col);
/* raw device pointers (for RCCL exchange / zero-copy wrap); offsets/validity may be NULL.  The table's content stamp (ldb_gpu_table_stamp)
 * does NOT change when somebody writes through these pointers: a writer calls ldb_gpu_table_set_rows (same row count is fine) or
 * ldb_gpu_table_write_fixed afterwards so that prepared plans record afresh.  A plan that replays over silently changed bytes is still
 * caught — its recorded counts are compared with the real ones at the end and the execution is repeated (LDB_TRACE_MISSED) — but that
 * costs one wasted execution. */
int32_t ldb_gpu_table_col_ptrs(const ldb_table* t, int32_t col, void** values, void** offsets, void** validity,
                               int64_t* value_bytes);
/* shrink the logical row count (receive buffers allocated at capacity) */
int32_t ldb_gpu_table_set_rows(ldb_table* t, int64_t n_rows);
/* blocking D2H copy of a fixed-width column's values (n_rows * width bytes) */
int32_t ldb_gpu_table_read_fixed(ldb_ctx* ctx, const ldb_table* t, int32_t col, void* host_out, int64_t out_bytes);
/* is `row` of the column non-NULL?  (scalar-subquery results: a key-less aggregate over no rows is one NULL row) */
int32_t ldb_gpu_table_row_valid(ldb_ctx* ctx, const ldb_table* t, int32_t col, int64_t row, int32_t* valid);
/* blocking H2D copy into a fixed-width column */
int32_t ldb_gpu_table_write_fixed(ldb_ctx* ctx, ldb_table* t, int32_t col, const void* host_in, int64_t in_bytes);
/* device-to-device copy on the ctx stream (moving column buffers to / from RCCL exchange buffers) */
int32_t ldb_gpu_memcpy_d2d(ldb_ctx* ctx, void* dst, const void* src, int64_t bytes);

/* Replaces result materialisation into Arrow builders (a15: MaterializeTableLowering,
 * SubOpToControlFlow.cpp:984-1004; ArrowColumnBuilder, ArrowColumn.h:15-37).  Fills a struct
 * ArrowArray + ArrowSchema with host copies in the reference's physical types (narrowed
 * decimals are sign-extended back to 128 bit, LowerToStd.cpp:211-298).  Caller releases. */
int32_t ldb_gpu_export(ldb_ctx* ctx, const ldb_table* t, struct ArrowSchema* out_schema, struct ArrowArray* out_array);

/* ------------------------------------------------------------------ relations */
int32_t ldb_gpu_rel_from_table(ldb_ctx* ctx, const ldb_table* t, ldb_rel** out); /* identity rows */
int32_t ldb_gpu_rel_release(ldb_ctx* ctx, ldb_rel* r);
int64_t ldb_gpu_rel_rows(ldb_ctx* ctx, ldb_rel* r); /* syncs if the count is still on the device */
int32_t ldb_gpu_rel_sides(const ldb_rel* r);
/* blocking D2H of the row-id vector of one side (NULL row ids = identity → fills 0..n-1) */
int32_t ldb_gpu_rel_read_rowids(ldb_ctx* ctx, ldb_rel* r, int32_t side, uint32_t* host_out, int64_t cap);
/* gather the listed columns into a dense device table (late materialisation) */
int32_t ldb_gpu_materialize(ldb_ctx* ctx, ldb_rel* r, const ldb_colref* cols, int32_t n_cols, ldb_table** out);

/* ------------------------------------------------------------------ scan + filter (a2, a3, a4) */
/* FilterOp mirrors lingodb::runtime::FilterOp (include/lingodb/runtime/storage/TableStorage.h:14-24). */
typedef enum {
   LDB_F_EQ = 0,
   LDB_F_NEQ = 1,
   LDB_F_LT = 2,
   LDB_F_LTE = 3,
   LDB_F_GT = 4,
   LDB_F_GTE = 5,
   LDB_F_NOTNULL = 6,
   LDB_F_IN = 7,
   /* beyond FilterOp: string predicates the reference evaluates in generated code by calling
    * StringRuntime::like (src/runtime/StringRuntime.cpp:134-136; escape '\\').  utf8 column,
    * rhs_kind = STRING, pattern in (str, str_len). */
   LDB_F_LIKE = 8,
   LDB_F_NOT_LIKE = 9
} ldb_filter_op;
typedef enum { LDB_RHS_INT = 0, LDB_RHS_STRING = 1, LDB_RHS_COLUMN = 2, LDB_RHS_FLOAT = 3 } ldb_rhs_kind;

/* One conjunct.  Constants are already typed against the column (the host mirror of
 * Restrictions::create, Restrictions.cpp:392-521, does date parsing / decimal rescaling):
 *   ints, dates (days), char(1) (4 raw bytes as int32), decimals (unscaled at column scale)
 *   travel as a 128-bit integer (value_lo, value_hi); strings as (str, str_len).
 * rhs_kind = COLUMN compares two columns (residual predicates such as l_commitdate <
 * l_receiptdate that the reference evaluates in generated code). */
typedef struct {
   ldb_colref col;
   int32_t op; /* ldb_filter_op */
   int32_t rhs_kind; /* ldb_rhs_kind */
   uint64_t value_lo;
   int64_t value_hi;
   double value_f64;
   const char* str;
   int32_t str_len;
   ldb_colref rhs_col;
   /* IN lists: n_in constants; ints as lo/hi pairs (2*n_in words), strings as pointers+lengths */
   int32_t n_in;
   const int64_t* in_values;
   const char* const* in_strs;
   const int32_t* in_str_lens;
} ldb_filter_desc;

/* Replaces ScanBatchesTask::unitRun + Restrictions::applyFilters (LingoDBTable.cpp:382-407,
 * Restrictions.cpp:365-390): conjunction evaluated in descriptor order; the result relation
 * holds the passing rows in ascending row order (the order of the reference's selection
 * vectors inside a morsel, morsels in table order). */
int32_t ldb_gpu_scan_filter(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds, ldb_rel** out);
/* How a LIKE / NOT LIKE pattern will be matched (no device needed): *n_segments = 0 → the general matcher
 * (StringRuntime::like semantics: '_', escapes, non-ASCII); otherwise the pattern is ASCII literals separated
 * by '%' (≤ 4 of ≤ 16 bytes) and is matched by position inside the scan kernel — seg[2j], seg[2j+1] = start and
 * length of literal j in the pattern, *anchors bit 0 / bit 1 = no leading / trailing '%'. */
int32_t ldb_gpu_like_plan(const char* pattern, int32_t len, int32_t* n_segments, int32_t* seg, int32_t* anchors);

/* Disjunctive normal form: rows satisfying (clause 0) OR (clause 1) OR …, each clause a conjunction
 * of clause_sizes[c] consecutive entries of `preds` (<= 4 clauses, <= 24 conjuncts in all) — TPC-H
 * Q19's three alternatives.  The reference evaluates such a predicate as generated residual code
 * (db.or of db.and trees); ascending row order as for ldb_gpu_scan_filter. */
int32_t ldb_gpu_scan_filter_dnf(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, const int32_t* clause_sizes, int32_t n_clauses, ldb_rel** out);
/* count only (no selection written) */
int32_t ldb_gpu_scan_count(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds, int64_t* count);

/* ------------------------------------------------------------------ hash (a5) */
/* db.hash over the key columns, bit-identical to HashLowering (LowerToStd.cpp:1065-1152) +
 * Hash64/HashCombine/VarLenTryCheapHash (LowerToLLVM.cpp:372-391,493-524).
 * Output: a 1-column LDB_T_INT64 device table (`hash`). */
int32_t ldb_gpu_hash_keys(ldb_ctx* ctx, ldb_rel* in, const ldb_colref* keys, int32_t n_keys, ldb_table** out);

/* ------------------------------------------------------------------ scalar functions as computed columns */
/* The generated code calls runtime functions per tuple (rt::DateRuntime::extractYear …,
 * src/runtime/DateRuntime.cpp:99-101); where such a value is a group key or join key it becomes a
 * computed column here: one device column with a value per row of `in` (NULL in → NULL out),
 * attached to the relation with ldb_gpu_rel_zip. */
typedef enum { LDB_FN_EXTRACT_YEAR = 0 /* date32 → int64 civil year */ } ldb_scalar_fn;
int32_t ldb_gpu_map_column(ldb_ctx* ctx, ldb_rel* in, ldb_colref col, int32_t fn, const char* name, ldb_table** out);
/* Arithmetic on aggregate results (a16): `literal * num / den` over two decimal or integer columns
 * as one decimal128(out_precision, out_scale) column,
 *    out = ((((num * mul) sdiv 10^mul_div_pow10) * 10^pow10) sdiv den   in wrapping 128-bit arithmetic,
 * i.e. DecimalMulOpLowering (LowerToStd.cpp:653-677; mul_div_pow10 = sLit + sNum − sProduct, non-zero
 * only when the product's scale was clamped) followed by DecimalOpScaledLowering (:631-651;
 * pow10 = sRes + sDen − sProduct).  `mul` (mul_lo/mul_hi) is the literal as an integer at its own
 * scale (1 for a plain division); the exponents and the result type come from the caller's type
 * derivation (sql_analyzer.cpp:3083-3159).  NULL in → NULL out; den = 0 → NULL (undefined in the
 * reference). */
int32_t ldb_gpu_map_muldiv(ldb_ctx* ctx, ldb_rel* in, ldb_colref num, int64_t mul_lo, int64_t mul_hi, int32_t mul_div_pow10, int32_t pow10, ldb_colref den,
                           int32_t out_precision, int32_t out_scale, const char* name, ldb_table** out);
/* General scalar projection (a16): one computed column from a POSTFIX program over the columns of
 * `in`, evaluated per row on a stack (depth <= 8) of nullable 128-bit integers in wrapping
 * arithmetic — the db.add / db.sub / db.mul / db.div trees of DecimalBinOpLowering,
 * DecimalMulOpLowering and DecimalOpScaledLowering (LowerToStd.cpp:622-699) once the frontend has
 * fixed the scales (casts = MUL_POW10 / SDIV_POW10, sql_analyzer.cpp:3058-3159), comparisons
 * (:374-466), CASE (scf.if on db.derive_truth) and NULL handling.  Arithmetic and comparisons yield
 * NULL when an operand is NULL; AND / OR are three-valued; SDIV by zero yields NULL.
 * out_type: INT32 / INT64 / DATE32 / DECIMAL128(p, s) / BOOL8. */
typedef enum {
   LDB_X_COL = 0, /* push column `col` (integer, decimal, date, char(1), bool) */
   LDB_X_CONST = 1, /* push the 128-bit constant (lo, hi) */
   LDB_X_ADD = 2,
   LDB_X_SUB = 3, /* a b → a - b */
   LDB_X_MUL = 4,
   LDB_X_SDIV = 5, /* a b → a sdiv b (truncating, arith.divsi) */
   LDB_X_MUL_POW10 = 6, /* a → a * 10^arg */
   LDB_X_SDIV_POW10 = 7, /* a → a sdiv 10^arg */
   LDB_X_NEG = 8,
   LDB_X_CMP = 9, /* a b → a OP b, arg = ldb_filter_op (EQ .. GTE) */
   LDB_X_AND = 10,
   LDB_X_OR = 11,
   LDB_X_NOT = 12,
   LDB_X_SELECT = 13, /* c a b → (c is true) ? a : b */
   LDB_X_ISNULL = 14,
   LDB_X_COALESCE = 15, /* a b → a unless NULL, then b */
   LDB_X_ROW = 16 /* push the LOGICAL row number of the relation (0 … n-1): the identity of a tuple of the outer stream when a
                     nested_map's inner pipeline is reduced per outer tuple (subop.nested_map, SubOpToControlFlow.cpp:1204-1250) */
} ldb_xop;
typedef struct {
   int32_t op; /* ldb_xop */
   int32_t arg;
   ldb_colref col;
   int64_t lo;
   int64_t hi;
} ldb_xinstr;
#define LDB_MAX_XPROG 32
int32_t ldb_gpu_map_expr(ldb_ctx* ctx, ldb_rel* in, const ldb_xinstr* prog, int32_t n_instr, ldb_coltype out_type, const char* name, ldb_table** out);
/* substring(col from `from` for `for_len`) of a utf8 column as a new utf8 column, character
 * (UTF-8) positions from 1 with the reference's legalisation of out-of-range arguments
 * (StringRuntime::substr, src/runtime/StringRuntime.cpp:292-319).  NULL in → NULL out. */
int32_t ldb_gpu_map_substr(ldb_ctx* ctx, ldb_rel* in, ldb_colref col, int64_t from, int64_t for_len, const char* name, ldb_table** out);
/* `in` extended by a table of exactly ldb_gpu_rel_rows(in) rows as a new LAST side (identity row
 * ids); the table must outlive the relation. */
int32_t ldb_gpu_rel_zip(ldb_ctx* ctx, ldb_rel* in, const ldb_table* t, ldb_rel** out);

/* ------------------------------------------------------------------ expressions (a16) */
/* Integer/decimal expression in sum-of-products normal form:
 *     value = Σ_t sign_t * ( Π_f (a_f + b_f * col_f) ) / 10^div_pow10_t      (128-bit, wrapping)
 * which is what the reference's decimal lowerings reduce to once the frontend has fixed the
 * scales: DecimalMulOpLowering = integer multiply (+ truncating divide when the result scale
 * was clamped, LowerToStd.cpp:653-677), add/sub = integer add after common-scale casts
 * (:680-699), int→decimal(19,0) casts = multiply by 10^k (folded into a/b). */
#define LDB_MAX_FACTORS 3
#define LDB_MAX_TERMS 2
typedef struct {
   int32_t has_col;
   ldb_colref col;
   int64_t a;
   int64_t b;
} ldb_factor;
typedef struct {
   int32_t n_factors;
   int32_t negate;
   int32_t div_pow10;
   int32_t reserved;
   ldb_factor f[LDB_MAX_FACTORS];
} ldb_term;
typedef struct {
   int32_t n_terms;
   int32_t is_float; /* 1: evaluate the same form in f64 (float/double columns) */
   ldb_term t[LDB_MAX_TERMS];
} ldb_expr;

/* ------------------------------------------------------------------ group-by (a9, a10, a11, a14) */
typedef enum { LDB_AGG_SUM = 0, LDB_AGG_MIN = 1, LDB_AGG_MAX = 2, LDB_AGG_COUNT = 3, LDB_AGG_COUNT_STAR = 4, LDB_AGG_ANY = 5, LDB_AGG_AVG = 6 } ldb_agg_fn;
#define LDB_MAX_AGG_PREDS 3
typedef struct {
   int32_t fn; /* ldb_agg_fn */
   int32_t wide; /* 1: accumulate/emit as 128-bit (decimal p>=19), 0: int64 (SUM type = arg type, sql_analyzer.cpp:2631) */
   ldb_expr arg;
   /* conditional aggregate: sum(case when <preds> then arg else 0 end) */
   int32_t n_preds;
   ldb_filter_desc preds[LDB_MAX_AGG_PREDS];
   /* AVG = (SUM * 10^avg_pow10) sdiv COUNT in 128 bit (DecimalOpScaledLowering, LowerToStd.cpp:631-651) */
   int32_t avg_pow10;
   /* result column type written to the output table: LDB_T_INT64 (COUNT, integer SUM),
    * LDB_T_DECIMAL128 (out_precision, out_scale), LDB_T_DATE32 / LDB_T_INT32 / LDB_T_CHAR4
    * (MIN/MAX/ANY of such columns), LDB_T_FLOAT64 */
   int32_t out_type;
   int32_t out_precision;
   int32_t out_scale;
   /* AVG only, has_count_expr != 0: the divisor is SUM(count_expr) instead of the number of
    * contributing rows — the combine step for partial (sum, count) states (the reference merges
    * thread-local aggregate states the same way, MergeThreadLocal*, SubOpToControlFlow.cpp:1733,
    * 1861-1938); used to merge per-GPU partial aggregates. */
   int32_t has_count_expr;
   ldb_expr count_expr;
} ldb_agg_spec;

/* Replaces PreAggregationHashtableFragment::insert + generated lookup/update
 * (PreAggregationHashtable.cpp:46-60, SubOpToControlFlow.cpp:3065-3157, 3719-3768),
 * PreAggregationHashtable::merge (:76-158), Hashtable (Hashtable.cpp) and, for n_keys == 0,
 * SimpleState (SimpleState.cpp:8-30).  `preds` are fused into the same pass (the scan is not
 * materialised).  Output table: key columns (input types) then one column per aggregate.
 * Group order is unspecified (as in the reference).  est_groups: optimiser estimate, 0 = unknown. */
int32_t ldb_gpu_groupby(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds,
                        const ldb_colref* keys, int32_t n_keys, const ldb_agg_spec* aggs, int32_t n_aggs,
                        int64_t est_groups, ldb_table** out);

/* ------------------------------------------------------------------ hash join (a6, a7, a8) */
typedef enum {
   LDB_JOIN_INNER = 0,
   LDB_JOIN_SEMI = 1,
   LDB_JOIN_ANTI = 2,
   LDB_JOIN_LEFT_OUTER = 3,
   LDB_JOIN_MARK = 4,
   LDB_JOIN_SINGLE = 5,
   /* build-side semi / anti join: the result is the BUILD relation restricted to the rows with at
    * least one (SEMI_BUILD) / no (ANTI_BUILD) matching probe row, in ascending build order — the
    * reference's reverseSides scheme (translateHJWithMarker, RelAlgToSubOp.cpp:1248-1287) */
   LDB_JOIN_SEMI_BUILD = 6,
   LDB_JOIN_ANTI_BUILD = 7,
   /* outer joins that (also) keep the BUILD side's unmatched rows — the reference's HashMultiMap whose entries carry a
    * marker set by matching probe tuples and scanned afterwards (include/lingodb/runtime/HashMultiMap.h:6-35,
    * OuterJoinLowering / FullOuterJoinLowering with reverseSides, RelAlgToSubOp.cpp:1217-1294, 1446-1527): the result is
    * the INNER (RIGHT_OUTER) / LEFT_OUTER (FULL_OUTER) pairs followed by the build rows no probe row matched, their
    * probe sides padded with LDB_NULL_ROW */
   LDB_JOIN_RIGHT_OUTER = 8,
   LDB_JOIN_FULL_OUTER = 9
} ldb_join_kind;

/* Replaces GrowingBuffer::insert materialisation + HashIndexedView::build
 * (GrowingBuffer.cpp:44, LazyJoinHashtable.cpp:12-34): builds an index over the rows of
 * `build` keyed by `keys`.  No payload is copied (late materialisation: the table stores
 * build row numbers).  build_unique != 0: keys are known unique (primary key). */
int32_t ldb_gpu_join_build(ldb_ctx* ctx, ldb_rel* build, const ldb_colref* keys, int32_t n_keys, int32_t build_unique,
                           ldb_hashtable** out);
int32_t ldb_gpu_hashtable_release(ldb_ctx* ctx, ldb_hashtable* ht);
/* The table's device hash index over its (primary-key) columns `cols`: built on first use over ALL rows, owned by the table
 * and released with it — never by the caller.  Replaces the persisted LingoDBHashIndex (include/lingodb/runtime/
 * LingoDBHashIndex.h:18-61) that index nested-loop joins look up (translateINLJ, RelAlgToSubOp.cpp:1129-1205): the index is
 * an ordinary join table (for a dense primary key: the rank-bitmap layout, range / 4 bytes), probing it is ldb_gpu_join_probe. */
int32_t ldb_gpu_table_index(ldb_ctx* ctx, ldb_table* t, const int32_t* cols, int32_t n_cols, ldb_hashtable** out);
int64_t ldb_gpu_hashtable_slots(const ldb_hashtable* ht);
// bytes of the slot array (8 B per open-addressing slot; 4 B per key value of a direct-addressed table)
int64_t ldb_gpu_hashtable_bytes(const ldb_hashtable* ht);
/* Replaces LookupHashIndexedViewLowering + ScanListLowering (SubOpToControlFlow.cpp:2558-2586,
 * 2254-2313).  Output relation: INNER / LEFT_OUTER / SINGLE → sides = probe sides then build sides,
 * one row per match (LEFT_OUTER/SINGLE: unmatched probe rows carry LDB_NULL_ROW on build sides);
 * SEMI / ANTI → probe sides only, ascending; MARK → probe sides, all rows, plus *mark_out =
 * 1-column BOOL8 table.  NULL keys never match. */
int32_t ldb_gpu_join_probe(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys,
                           int32_t kind, ldb_rel** out, ldb_table** mark_out);
/* The same with residual conjuncts of the join predicate: a key match counts only when every
 * `probe_col OP build_col` holds on the candidate pair (integer / decimal / date columns; NULL
 * operands fail).  In the reference a non-equality part of a join predicate is the filter behind
 * the lookup (SpecializeSubOpPass.cpp:152-205) — e.g. Q21's l2.l_suppkey <> l1.l_suppkey inside
 * EXISTS / NOT EXISTS (semi / anti join with residual, RelAlgToSubOp.cpp:1340-1410). */
typedef struct {
   ldb_colref probe_col;
   ldb_colref build_col;
   int32_t op; /* LDB_F_EQ .. LDB_F_GTE */
   int32_t reserved;
} ldb_join_residual;
int32_t ldb_gpu_join_probe_residual(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int32_t kind,
                                    const ldb_join_residual* resid, int32_t n_resid, ldb_rel** out, ldb_table** mark_out);
/* Build-side semi AND anti join over the same hash table in one pass over the probe side (TPC-H Q21's EXISTS … AND NOT EXISTS … pair; the
 * reference runs two marker joins, translateHJWithMarker, RelAlgToSubOp.cpp:1248-1287, residuals as in SpecializeSubOpPass.cpp:152-205):
 * *out = the build rows with a partner among the probe rows (key equality + residual conjuncts) and WITHOUT a partner among the probe
 * rows that also satisfy the conjunction `anti_preds` (columns of the probe relation; at most 2).  Same rows as kind SEMI_BUILD, a table
 * over its result, and kind ANTI_BUILD probed by the filtered probe side. */
int32_t ldb_gpu_join_probe_semi_anti_build(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, const ldb_join_residual* resid,
                                           int32_t n_resid, const ldb_filter_desc* anti_preds, int32_t n_anti_preds, ldb_rel** out);
/* Nested-loop join (translateNLJ, RelAlgToSubOp.cpp:948-1033): no key equality — the join predicate is the conjunction of
 * `resid` (0..2 column-vs-column comparisons between a probe and a build column; none = cross product).  Every build row is
 * visited for every probe row, so this is for SMALL build sides (band joins against dimension tables, scalar-subquery
 * cross products).  kinds: INNER, LEFT_OUTER, SEMI, ANTI, MARK, SINGLE, SEMI_BUILD, ANTI_BUILD; result sides as ldb_gpu_join_probe. */
int32_t ldb_gpu_join_nl(ldb_ctx* ctx, ldb_rel* probe, ldb_rel* build, int32_t kind, const ldb_join_residual* resid, int32_t n_resid, ldb_rel** out,
                        ldb_table** mark_out);
/* count matches only — the probe micro-benchmark kernel (Grows/s) */
int32_t ldb_gpu_join_probe_count(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys,
                                 int64_t* matches);

/* ------------------------------------------------------------------ sort / top-k (a12, a13) */
typedef struct {
   ldb_colref col;
   int32_t descending;
   int32_t reserved;
} ldb_sort_spec;
/* Replaces GrowingBuffer::sort / parallelSort (GrowingBuffer.cpp:54-78, Sorting.cpp:343-393)
 * with comparator semantics of db.sort_compare (LowerToStd.cpp:1046-1064).  Stable w.r.t. input
 * order among equal keys (the reference's std::sort leaves that order unspecified). */
int32_t ldb_gpu_sort(ldb_ctx* ctx, ldb_rel* in, const ldb_sort_spec* specs, int32_t n_specs, ldb_rel** out);
/* Replaces Heap (Heap.cpp:8-72): first k rows of the sorted order. */
int32_t ldb_gpu_topk(ldb_ctx* ctx, ldb_rel* in, const ldb_sort_spec* specs, int32_t n_specs, int64_t k, ldb_rel** out);

/* ------------------------------------------------------------------ multi-GPU shuffle (§8(e), new) */
/* Hash-radix partition the listed columns of `in` into `nparts` destinations:
 * dest = (db.hash(keys) >> 16) % nparts  (reference hash, so every GPU agrees).
 * Output: a dense table whose rows are grouped by destination; counts[nparts] on the host.
 * The exchange itself is an RCCL all-to-all on the column buffers (ldb_gpu_table_col_ptrs). */
int32_t ldb_gpu_partition(ldb_ctx* ctx, ldb_rel* in, const ldb_colref* keys, int32_t n_keys, int32_t nparts,
                          const ldb_colref* cols, int32_t n_cols, ldb_table** out, int64_t* counts);

/* ------------------------------------------------------------------ set operations, window functions */
/* Replaces UnionAllLowering / UnionDistinctLowering / CountingSetOperationLowering
 * (src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp:622-930): the reference keeps a map keyed by ALL columns with
 * one i64 counter per input and emits per key one row (distinct semantics: UNION; INTERSECT needs both counters > 0,
 * EXCEPT the left > 0 and the right = 0) or min(c1, c2) / max(c1 - c2, 0) rows (INTERSECT ALL / EXCEPT ALL).  NULLs
 * compare equal (the lowering's `isa` compare block).  The two column lists must have pairwise equal types (the
 * frontend inserts the casts).  Output: a table with the left input's column names; row order unspecified. */
typedef enum { LDB_SET_UNION_ALL = 0, LDB_SET_UNION = 1, LDB_SET_INTERSECT = 2, LDB_SET_INTERSECT_ALL = 3, LDB_SET_EXCEPT = 4, LDB_SET_EXCEPT_ALL = 5 } ldb_set_op;
int32_t ldb_gpu_set_op(ldb_ctx* ctx, ldb_rel* left, const ldb_colref* left_cols, ldb_rel* right, const ldb_colref* right_cols, int32_t n_cols, int32_t op, ldb_table** out);

/* Replaces WindowLowering (RelAlgToSubOp.cpp:2193-2553) with its SegmentTreeView (include/lingodb/runtime/SegmentTreeView.h:11-43):
 * rows are ordered by (PARTITION BY keys, ORDER BY keys); the frame of a row is ROWS BETWEEN frame_from AND frame_to as
 * offsets from the current row (negative = preceding, 0 = current row, LDB_FRAME_UNBOUNDED_* = partition begin / end),
 * each end clamped into the partition (OffsetReferenceByLowering, SubOpToControlFlow.cpp:3860-3885 — so a frame is never
 * empty).  LDB_WIN_RANK = entries between the frame begin and the current row + 1 (RankWindowFunc :2043-2058; row-number
 * semantics, as in the reference); SUM / MIN / MAX (NULL when the frame holds only NULLs) / COUNT (non-NULL values) /
 * COUNT_STAR over the frame.  Output: *out_rel = the input's rows in window order, *out_cols = one column per function
 * aligned with it (attach it with ldb_gpu_rel_zip).  Integer / decimal / date argument columns. */
typedef enum { LDB_WIN_RANK = 0, LDB_WIN_SUM = 1, LDB_WIN_MIN = 2, LDB_WIN_MAX = 3, LDB_WIN_COUNT = 4, LDB_WIN_COUNT_STAR = 5 } ldb_window_fn_kind;
typedef struct {
   int32_t fn; /* ldb_window_fn_kind */
   ldb_colref col; /* argument (ignored by RANK / COUNT_STAR) */
} ldb_window_fn;
#define LDB_FRAME_UNBOUNDED_PRECEDING INT64_MIN
#define LDB_FRAME_UNBOUNDED_FOLLOWING INT64_MAX
int32_t ldb_gpu_window(ldb_ctx* ctx, ldb_rel* in, const ldb_colref* part_keys, int32_t n_part, const ldb_sort_spec* order, int32_t n_order, int64_t frame_from, int64_t frame_to,
                       const ldb_window_fn* fns, int32_t n_fns, ldb_rel** out_rel, ldb_table** out_cols);

/* The exchange itself: one rank (= one ldb_ctx) per GPU.  Rank 0 makes a 128-byte id, the host process hands
 * it to every rank by whatever channel it has (the LingoDB side: its session layer; bench.py:
 * torch.distributed; the tests: a file), and every rank joins with ldb_gpu_comm_create.  Two transports
 * (option `comm_transport`, read by ldb_gpu_comm_unique_id; the id carries the choice to every rank):
 *   0 = RCCL over xGMI (default; id = ncclGetUniqueId): all transfers are issued on the context's stream as
 *       ONE grouped batch of point-to-point sends / receives per call (every peer pair has its own xGMI
 *       link), straight into the column buffers of the result table;
 *   1 = host-staged through POSIX shared memory: the ranks are processes of one node and may share a GPU
 *       (RCCL refuses that).  Same grouped-transfer interface, so every world > 1 code path of the exchange
 *       runs under test on a one-GPU box; also the fallback when RCCL cannot be initialised.  Waits are
 *       bounded by option `comm_timeout_ms` (default 120 000).
 * The reference has no counterpart (single process: src/runtime/GPU/CUDA/CMakeLists.txt:9 "we do not support
 * multi-gpu"); SURVEY §8(e) assigns the design to this library. */
typedef struct ldb_comm ldb_comm;
int32_t ldb_gpu_comm_available(void); /* 1 = librccl is loadable in this process (transport 0 can work) */
int32_t ldb_gpu_comm_unique_id(void* id128);
int32_t ldb_gpu_comm_create(ldb_ctx* ctx, int32_t rank, int32_t world, const void* id128, ldb_comm** out);
int32_t ldb_gpu_comm_destroy(ldb_comm* comm);
int32_t ldb_gpu_comm_rank(const ldb_comm* comm);
int32_t ldb_gpu_comm_world(const ldb_comm* comm);
const char* ldb_gpu_comm_transport(const ldb_comm* comm); /* "rccl" | "shm" */
/* bytes and time of this rank's transfer groups since the last reset (transfers to / from other ranks only) */
typedef struct {
   int64_t groups; /* grouped batches of sends / receives (one metadata + one data batch per exchange) */
   int64_t bytes_out, bytes_in;
   int64_t max_peer_bytes_out; /* the busiest peer link of this rank */
   double host_ms; /* host wall time inside the groups */
   double device_ms; /* ctx-stream time of the groups (HIP events around each) */
} ldb_comm_stats;
int32_t ldb_gpu_comm_stats(ldb_comm* comm, ldb_comm_stats* out, int32_t reset);
/* the host-staged transport without a device (host pointers): CPU tests of the protocol with world > 1, and
 * exchange of host-side metadata between the ranks' host programs */
int32_t ldb_gpu_comm_create_host(int32_t rank, int32_t world, const void* id128, ldb_comm** out);
/* raw all-to-all of bytes (one grouped batch): send_bytes[p] bytes of `send` (peer runs back to back) go to peer p,
 * recv_bytes[p] bytes from peer p arrive in `recv` (peer runs back to back).  Device pointers for a device
 * communicator, host pointers for ldb_gpu_comm_create_host */
/* *all_min = the minimum of `mine` over all ranks (one 8-byte transfer to and from every peer + one wait): how the ranks of a sharded
 * prepared plan agree to replay together and, afterwards, whether every rank's replay was confirmed.  ctx may be NULL for a host communicator. */
int32_t ldb_gpu_comm_agree(ldb_ctx* ctx, ldb_comm* comm, int32_t mine, int32_t* all_min);
int32_t ldb_gpu_comm_alltoall_bytes(ldb_comm* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes);
/* every rank's rows of `t` concatenated in rank order on every rank (replicated small build sides,
 * partial aggregates; fixed-width and utf8 columns, validity bitmaps travel along) */
int32_t ldb_gpu_allgather(ldb_ctx* ctx, ldb_comm* comm, const ldb_table* t, const char* name, ldb_table** out);
/* `t` holds send_counts[p] rows for rank p, in rank order (ldb_gpu_partition's layout): the result holds
 * the rows this rank receives from rank 0, 1, … in that order */
int32_t ldb_gpu_alltoall(ldb_ctx* ctx, ldb_comm* comm, const ldb_table* t, const int64_t* send_counts, const char* name, ldb_table** out);
/* hash-radix shuffle = ldb_gpu_partition (dest = (db.hash(keys) >> 16) % world) + ldb_gpu_alltoall:
 * afterwards equal keys are on the same rank (one all-to-all per repartitioned input) */
int32_t ldb_gpu_shuffle(ldb_ctx* ctx, ldb_comm* comm, ldb_rel* in, const ldb_colref* keys, int32_t n_keys, const ldb_colref* cols, int32_t n_cols, const char* name,
                        ldb_table** out);

#ifdef __cplusplus
}
#endif
#endif /* LINGODB_GPU_H */
