// ldb_internal.h — host-side structures of liblingodb_gpu.so (not part of the ABI).
#pragma once
#include "../../include/lingodb_gpu.h"
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

// ---------------------------------------------------------------- errors
void ldb_set_error(const char* fmt, ...);
#define LDB_FAIL(code, ...)        \
   do {                            \
      ldb_set_error(__VA_ARGS__);  \
      return (code);               \
   } while (0)
#define LDB_HIP(expr)                                                                       \
   do {                                                                                     \
      hipError_t e_ = (expr);                                                               \
      if (e_ != hipSuccess) {                                                               \
         ldb_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
         return e_ == hipErrorOutOfMemory ? LDB_ERR_OOM : LDB_ERR_HIP;                      \
      }                                                                                     \
   } while (0)
#define LDB_TRY(expr)            \
   do {                          \
      int32_t s_ = (expr);       \
      if (s_ != LDB_OK) return s_; \
   } while (0)

// process-wide option (env LDB_<NAME> on first use, ldb_gpu_set_option afterwards)
int64_t ldb_option(const char* name, int64_t dflt);

// device-visible descriptors (DCol, DPred, DKeys) live in ldb_devtypes.h
#include "ldb_devtypes.h"

// ---------------------------------------------------------------- host objects
struct ldb_column {
   std::string name;
   ldb_coltype type{};
   int32_t width = 0; // device bytes per value (0 for utf8)
   void* values = nullptr; // device
   int64_t* offsets = nullptr; // device, utf8: int64[n+1]
   uint8_t* validity = nullptr; // device bitmap or NULL
   int64_t value_bytes = 0; // bytes in `values`
   int64_t null_count = 0;
   bool owned = true;
   // utf8 dictionary (the reference stores char(n) / varchar as utf8, LingoDBTable.cpp:184-191): for a
   // column with few distinct strings, built at registration — an ORDER-PRESERVING dictionary (codes follow the bytewise
   // order of the strings, StringRuntime.cpp:242-256) kept BESIDE the strings: dict_codes[row] = code (0xFFFFFFFF for
   // NULL), `dict` = the distinct strings in code order.  Predicates with constants become code-set tests, group-by and
   // sort keys read the 4-byte codes; results still show the strings (ldb_dict.hip).
   uint32_t* dict_codes = nullptr; // device, one per physical row (owned with the column)
   struct ldb_table* dict = nullptr; // one utf8 column, dict_size rows (owned)
   int32_t dict_size = 0;
   std::unordered_map<std::string, std::string>* dict_pred_cache = nullptr; // predicate bytes → 128-byte code set
   // value range of a fixed-width integer-like column, computed on first use and cached (the
   // kind of per-column statistic a catalog keeps; see ldb_column_range)
   mutable bool has_range = false;
   mutable int64_t vmin = 0, vmax = -1;
   mutable bool skewed = false; // ordered hash-table slots over this column gave long probe runs once: do not try again
   mutable int8_t sorted_state = -1; // -1 unknown, 0 no, 1 values are non-decreasing and there are no NULLs (ldb_column_sorted)
   // zone map: min / max per LDB_ZONE_ROWS physical rows (device, owned with the column), built on the first range
   // predicate over the column; zone_state 0 = the zones span most of the value range (unclustered data): not used
   mutable int64_t* zone_min = nullptr;
   mutable int64_t* zone_max = nullptr;
   mutable int8_t zone_state = -1; // -1 unknown, 0 useless, 1 useful
};
// zone map of an integer-like NOT NULL column: device addresses of the per-zone minima / maxima, or 0 / 0 when the type
// does not qualify or the zones are not selective.  Built once per column (one pass over it), cached.
int32_t ldb_column_zones(ldb_ctx* ctx, const struct ldb_table* t, int32_t col, uint64_t* zmin, uint64_t* zmax);
// [min, max] over all physical rows of an integer-like column (NULL slots included: a superset is
// fine for its users); cached in the column.  LDB_ERR_UNSUPPORTED for other types.
int32_t ldb_column_range(ldb_ctx* ctx, const struct ldb_table* t, int32_t col, int64_t* lo, int64_t* hi);
// is the (integer-like, NOT NULL) column stored in non-decreasing order?  Cached like the range.
int32_t ldb_column_sorted(ldb_ctx* ctx, const struct ldb_table* t, int32_t col, bool* sorted);

struct ldb_prof_pending {
   const char* name;
   hipEvent_t start, stop;
};
struct ldb_prof_total {
   std::string name;
   int64_t launches = 0;
   double ms = 0;
   double max_ms = 0; // the longest single launch (an operator may launch the same kernel on inputs of very different size)
};
#define LDB_RING_BYTES ((size_t) 1 << 20)

// ---------------------------------------------------------------- read-backs and their trace (prepared plans)
// Every device → host read of a count / flag / control block goes through ldb_readback.  Outside a trace it is an
// asynchronous copy followed by a stream synchronisation (the host needs the value to size the next allocation).  Inside a
// trace (ldb_gpu_trace_begin … _end, what a prepared plan brackets each execution with) the values are RECORDED in call
// order; the next execution over the same, unchanged inputs REPLAYS them: ldb_readback returns the recorded value at once
// and only queues a copy of the real value into a pinned log, so the host runs ahead of the device through the whole plan
// and waits ONCE, in ldb_gpu_trace_end, where the log is compared with the record.  Counts are pure functions of the plan
// and the data, so the comparison can only fail when something the trace key does not cover changed; then the execution is
// discarded and repeated without replay.  The reference's counterpart is the fused pipeline that never returns to the host
// between operators (ScanRefsTableLowering, SubOpToControlFlow.cpp:1123-1202; one main() per query, LLVMBackends.cpp:856-865).
#define LDB_RB_MAX 4096 /* largest read that can be recorded; larger reads always synchronise */
struct ldb_trace_entry {
   uint32_t site; // which call site asked (hash of file + line)
   uint32_t bytes;
   uint32_t off; // offset of the value in ldb_trace::vals and in the pinned log (8-byte aligned)
   uint32_t flags; // LDB_RB_ORDER_DEPENDENT: a mis-speculation here is counted apart (ldb_gpu_order_dependent_misses)
};
struct ldb_trace {
   std::vector<ldb_trace_entry> entries;
   std::vector<uint8_t> vals;
   bool complete = false; // the last execution that wrote it ran to ldb_gpu_trace_end
   int64_t replays = 0, records = 0, misses = 0, diverged = 0;
};
#define LDB_LOG_BYTES ((size_t) 256 << 10)
constexpr uint32_t ldb_site_hash(const char* f, int line) {
   uint32_t h = 2166136261u;
   for (; *f; f++) h = (h ^ (uint32_t) (unsigned char) *f) * 16777619u;
   return (h ^ (uint32_t) line) * 16777619u;
}
// (the call site's file:line is remembered under its hash, so that a mismatching replayed value can be named: LDB_HOST_TRACE)
uint32_t ldb_site_note(uint32_t hash, const char* file, int line);
// registered ONCE per call site (a function-local static inside an immediately-invoked lambda): the hot host path of a replayed plan evaluates
// LDB_SITE at every read-back and must not take a process-wide mutex there
#define LDB_SITE ([]() -> uint32_t { static const uint32_t site_ = ldb_site_note(ldb_site_hash(__FILE__, __LINE__), __FILE__, __LINE__); return site_; }())
// a site derived from another (the scan's per-size sites): registered under the base site's name + the salt, for the mismatch diagnostic
uint32_t ldb_site_derived(uint32_t base, uint32_t salt);
struct ldb_ctx;
// read `bytes` of device memory into `host` (see above); flags: LDB_RB_NEVER_REPLAY for values that may differ between two
// executions over the same data (flags raised by races between insertions) — those always synchronise
#define LDB_RB_NEVER_REPLAY 1
// replayed, but the value may depend on the order in which a kernel's atomics happened (the open-addressing build's "long probe run" bit): a
// differing replay is caught at the trace's end like any other and the execution repeated — such repeats are counted apart, so that spurious
// re-runs (a key distribution sitting on the run-length threshold) are visible instead of looking like data-dependent mis-speculation
#define LDB_RB_ORDER_DEPENDENT 2
int32_t ldb_readback(ldb_ctx* ctx, void* host, const void* dev, size_t bytes, uint32_t site, int flags = 0);
#define LDB_READBACK(ctx, host, dev, bytes) ldb_readback((ctx), (host), (dev), (bytes), LDB_SITE)

struct ldb_ctx {
   bool prof_on = false;
   std::vector<ldb_prof_pending> prof_pending;
   std::vector<ldb_prof_total> prof_totals;
   std::vector<hipEvent_t> prof_free;
   int device = 0;
   hipStream_t stream = nullptr;
   bool own_stream = false;
   int cus = 256;
   std::vector<hipEvent_t> timers; // pairs (start, stop)
   // small pinned staging area for counts read back from the device
   int64_t* h_scratch = nullptr; // pinned, 64 words
   int64_t* d_scratch = nullptr; // device, 64 words
   // Block cache in front of the stream-ordered pool (ldb_dev_alloc / ldb_dev_free): every operator
   // call allocates ~10 temporaries, and a hipMallocAsync + hipFreeAsync pair costs ~15 µs of host
   // time — more than many of the kernels between them.  All work of a context is ordered on
   // ctx->stream, so a freed block may be handed to the next allocation without any wait.
   uint8_t* h_ring = nullptr; // pinned staging ring of ldb_dev_upload
   size_t ring_pos = 0;
   bool cache_on = true;
   size_t cache_bytes = 0, cache_cap = 0; // bytes parked in free lists / their limit
   std::unordered_map<void*, size_t> live; // block → its size class (bytes)
   std::unordered_map<size_t, std::vector<void*>> parked; // size class → free blocks (min-heaps by address: the lowest free block is handed out, so a
                                                          // repeated plan sees the same addresses whatever order its blocks were freed in)
   // descriptor cache of ldb_dev_upload: content hash → device copy (never written by a kernel: descriptors are const)
   struct DescEntry {
      void* dev;
      std::vector<uint8_t> copy;
   };
   std::unordered_multimap<uint64_t, DescEntry> desc_cache;
   std::unordered_map<void*, int> desc_blocks; // device copy → operators holding it (ldb_dev_upload … ldb_dev_free)
   std::unordered_map<void*, int> shared; // block → EXTRA holders beyond the first (ldb_dev_share): ldb_dev_free gives one back, the last one frees
   size_t desc_bytes = 0;
   int64_t desc_hits = 0, desc_misses = 0, desc_underflows = 0; // (underflow: a reference given back that nobody held — a caller's bug, counted and reported)
   // read-back trace (see ldb_readback)
   ldb_trace* trace = nullptr;
   int trace_mode = 0; // 0 none, 1 record, 2 replay
   size_t trace_pos = 0;
   bool trace_poisoned = false; // a replayed value was wrong: everything computed since is void
   bool trace_collective = false; // the traced plan exchanges rows with other ranks: a mis-speculating rank runs on (see ldb_readback)
   uint8_t* h_log = nullptr; // pinned, LDB_LOG_BYTES
   // counter arena (ldb_counters): zeroed 64-bit device words handed out front to back — an operator's counters / flags / totals
   // are words nobody else touches until the arena wraps, so (a) no operator clears its counters itself (one clear per plan
   // instead of one fill launch per operator) and (b) a replaying plan collects all of them with ONE copy kernel at its end
   // instead of one per read-back (ldb_readback defers reads of arena words)
   uint64_t* arena = nullptr; // device, LDB_ARENA_WORDS
   size_t arena_pos = 0;
   bool arena_wrapped = false; // the arena wrapped inside the current plan: allocations clear their own range until the next restart
   struct LogCopy {
      uint64_t src;
      uint32_t off, bytes;
   };
   std::vector<LogCopy> log_pending; // deferred copies of a replaying trace: arena words → pinned log
   // single-pass scans (ldb_exclusive_scan_*): tile status words + the ticket counter, never cleared — every scan call owns a
   // fresh epoch / ticket range (ldb_core.hip)
   uint64_t* scan_status = nullptr;
   size_t scan_status_tiles = 0;
   unsigned long long* scan_ticket = nullptr; // device
   uint64_t scan_ticket_base = 0; // tickets handed out so far (host mirror)
   uint32_t scan_epoch = 0;
};

uint64_t ldb_next_serial();
struct ldb_table {
   ldb_ctx* ctx = nullptr;
   std::string name;
   int64_t n_rows = 0;
   // identity of the table's CONTENT for prepared plans (ldb_gpu_table_stamp): a process-wide serial number taken at
   // creation and again whenever the rows are overwritten or the row count changes
   uint64_t serial = ldb_next_serial();
   std::vector<ldb_column> cols;
   int32_t dict_refs = 1; // a dictionary table is shared by the columns gathered from the one it was built for
   // device hash indexes over this table's key columns (ldb_gpu_table_index): built on first use, kept with the table —
   // the counterpart of the reference's persisted LingoDBHashIndex (include/lingodb/runtime/LingoDBHashIndex.h:18-61)
   struct Index {
      std::vector<int32_t> cols;
      struct ldb_rel* rel = nullptr; // identity relation over the table (what the hash table's build rows refer to)
      struct ldb_hashtable* ht = nullptr;
   };
   std::vector<Index> indexes;
};

struct ldb_rel_side {
   const ldb_table* table = nullptr;
   uint32_t* rowids = nullptr; // device, NULL = identity
   bool owned = false;
   bool may_null = false; // rowids may hold LDB_NULL_ROW (outer-join padding): only then gathered columns need a validity pass
};
struct ldb_rel {
   ldb_ctx* ctx = nullptr;
   int64_t n_rows = 0;
   std::vector<ldb_rel_side> sides;
   // A LAZY filtered relation: the rows of `sides` (all dense, no row ids) that satisfy the
   // conjunction `pending` — not evaluated yet.  ldb_gpu_scan_filter over a large base table
   // returns this; the consumers that fuse the filter into their own kernel (join probe,
   // group-by, count) evaluate the conjuncts per row and never materialise a row-id vector, which
   // is what the reference's fused pipelines do (scan → filter → probe in one generated loop).
   // Everything else calls ldb_rel_force first.  n_rows is the UNFILTERED row count while lazy.
   std::vector<DPred> pending;
};
// materialise a lazy relation in place (no-op otherwise)
int32_t ldb_rel_force(ldb_ctx* ctx, ldb_rel* r);

// kernel timing scope: LdbProf p(ctx, "k_name"); <launch>; (destructor records the stop event)
struct LdbProf {
   ldb_ctx* ctx;
   size_t idx = 0;
   bool active = false;
   LdbProf(ldb_ctx* c, const char* name);
   ~LdbProf();
};

// diagnostics (env LDB_HOST_TRACE=<ms>): host calls that took longer than <ms> milliseconds are reported on stderr with their site — how a
// blocking call inside a replayed plan is found without a profiler (LDB_HOST_TRACE=0.2 python bench.py …)
double ldb_host_trace_threshold();
struct LdbSlow {
   const char* what;
   size_t arg;
   double thr;
   std::chrono::steady_clock::time_point t0;
   LdbSlow(const char* w, size_t a = 0) : what(w), arg(a), thr(ldb_host_trace_threshold()) {
      if (thr >= 0) t0 = std::chrono::steady_clock::now();
   }
   ~LdbSlow() {
      if (thr < 0) return;
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms >= thr) fprintf(stderr, "[ldb host] %s(%zu): %.3f ms\n", what, arg, ms);
   }
};
// device allocation helpers (stream-ordered pool)
int32_t ldb_dev_alloc(ldb_ctx* ctx, void** out, size_t bytes);
void ldb_dev_free(ldb_ctx* ctx, void* p);
// one more holder of a read-only block (the row-id vector of a relation side carried over unchanged into the next relation): every holder calls
// ldb_dev_free, the last call frees
void ldb_dev_share(ldb_ctx* ctx, void* p);
// upload a host descriptor struct into device memory (stream-ordered)
// (read-only descriptors are cached by content, see ldb_core.hip; cacheable = false for memory a kernel will write)
int32_t ldb_dev_upload(ldb_ctx* ctx, const void* host, size_t bytes, void** dev_out, bool cacheable = true);
// a few bytes host → device through the pinned staging ring: a plain asynchronous DMA (hipMemcpyAsync from pageable memory is staged by the
// runtime and WAITS for the stream — inside a replayed plan that is the whole queue)
int32_t ldb_h2d_small(ldb_ctx* ctx, void* dev, const void* host, size_t bytes);

// device temporaries of one call: whatever is still listed when the scope ends is freed (error returns included)
struct LdbBufs {
   ldb_ctx* ctx;
   std::vector<void*> ptrs;
   explicit LdbBufs(ldb_ctx* c) : ctx(c) {}
   LdbBufs(const LdbBufs&) = delete;
   LdbBufs& operator=(const LdbBufs&) = delete;
   ~LdbBufs() {
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
   void keep(void* p) { // the caller takes the block over (it outlives the call)
      for (size_t i = 0; i < ptrs.size(); i++)
         if (ptrs[i] == p) {
            ptrs.erase(ptrs.begin() + (long) i);
            return;
         }
   }
};

// a descriptor uploaded for one call (ldb_dev_upload): its reference on the descriptor cache — or its block, when it was not cacheable — is
// given back when the scope ends, error returns included, so an entry no operator uses any more can always be evicted (desc_cache_mb stays
// a bound) and no holder can give back a reference twice
template <typename T>
struct LdbDesc {
   ldb_ctx* ctx;
   T* p = nullptr;
   explicit LdbDesc(ldb_ctx* c) : ctx(c) {}
   LdbDesc(const LdbDesc&) = delete;
   LdbDesc& operator=(const LdbDesc&) = delete;
   ~LdbDesc() { release(); }
   int32_t upload(const void* host, size_t bytes, bool cacheable = true) {
      release();
      return ldb_dev_upload(ctx, host, bytes, (void**) &p, cacheable);
   }
   void release() {
      if (p) ldb_dev_free(ctx, p);
      p = nullptr;
   }
};

// A utf8 column gathered from a dictionary-encoded one is LAZY: it holds the gathered codes and shares the dictionary, but has
// no offsets / bytes (values == offsets == NULL, value_bytes = -1) until a consumer needs them — group-by, join, sort and
// predicate evaluation read the codes; ldb_make_dcol (byte-wise consumers), export, the exchange, raw pointer access and
// concatenation call ldb_column_strings first, which writes the strings out of the dictionary (one pass over the codes).
// Q16: the 11 M distinct (brand, type, size, supplier) rows never had their two string columns written (2.5 ms of 10).
int32_t ldb_column_strings(ldb_ctx* ctx, const ldb_column& c, int64_t n_rows);
static inline bool ldb_column_is_lazy(const ldb_column& c) { return c.type.type == LDB_T_UTF8 && !c.offsets && c.dict_codes && c.dict; }
int32_t ldb_make_dcol(const ldb_rel* r, ldb_colref ref, DCol* out);
int32_t ldb_make_dpred(const ldb_rel* r, const ldb_filter_desc* p, DPred* out);
void ldb_mark_same_col(DPred* preds, int32_t n);
void ldb_like_plan(DPred* d);
void ldb_order_preds(DPred* preds, int32_t n);
int32_t ldb_make_dkeys(const ldb_rel* r, const ldb_colref* keys, int32_t n_keys, DKeys* out);
// the same with dictionary-encoded utf8 columns replaced by their int32 code columns — for consumers that only hash /
// compare / order the key INSIDE one relation (group-by, sort); never for joins, hash_keys or the exchange's partitioning
int32_t ldb_make_dcol_dict(const ldb_rel* r, ldb_colref ref, DCol* out);
int32_t ldb_make_dkeys_dict(const ldb_rel* r, const ldb_colref* keys, int32_t n_keys, DKeys* out);
// builds the dictionary of a utf8 column if it has at most max_distinct distinct strings (a no-op otherwise)
int32_t ldb_table_dict_encode(ldb_ctx* ctx, ldb_table* t, int32_t col, int32_t max_distinct);
int32_t ldb_table_dict_encode_all(ldb_ctx* ctx, ldb_table* t); // every eligible utf8 column (option dict_encode)
void ldb_column_dict_release(ldb_ctx* ctx, ldb_column& c);
// a string predicate with a constant over a dictionary-encoded column → a test on the codes; *done = false if not applicable
int32_t ldb_dict_rewrite_pred(const ldb_rel* r, const ldb_filter_desc* p, DPred* out, bool* done);
#define LDB_F_CODESET 100 /* internal DPred op: pass iff bit (code) of in_blob is set */
#define LDB_RHS_CODESET 100
#define LDB_DICT_MAX 1024 /* codes per dictionary = bits of DPred::in_blob */
int32_t ldb_width_of(const ldb_coltype& t, int narrow);

// launch geometry: blocks for a grid-stride kernel over n items
static inline int ldb_grid_for(const ldb_ctx* ctx, int64_t n, int block, int per_cu) {
   int64_t want = (n + block - 1) / block;
   int64_t cap = (int64_t) ctx->cus * per_cu;
   if (want < 1) want = 1;
   return (int) (want < cap ? want : cap);
}

// exclusive scan of n uint32 values on the device (in place allowed); total written to *d_total (uint64)
int32_t ldb_exclusive_scan_u32(ldb_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, uint64_t* d_total);
int32_t ldb_exclusive_scan_i64(ldb_ctx* ctx, const int64_t* d_in, int64_t* d_out, int64_t n, int64_t* d_total);
// out[j][i] = compose(ids[j], sel[which[j]][i]) for up to 12 row-id vectors in ONE launch (LDB_NULL_ROW passes through; ids[j] == NULL:
// the selection itself) — the hand-over of a join / selection result: one vector per side of the result relation
struct LdbComposeJob {
   const uint32_t* ids;
   uint32_t* out;
   int which;
};
int32_t ldb_compose_rowids(ldb_ctx* ctx, const uint32_t* sel0, const uint32_t* sel1, const LdbComposeJob* jobs, int n_jobs, uint64_t n);
// write-combining radix partition (ldb_wc.hip): n keys (+ payload: pay_in, or the item number when pay_in == NULL and pay_out
// != NULL) into nparts partitions q(key) = (key - bias) >> shift (0 for keys outside [bias, bias + range]), partitions back to
// back in q order in keys_out / pay_out; two tile-sorting passes above 64 partitions.  *part_offs_out (device, owned by the
// caller afterwards, may be NULL): begin of partition q at [q * *chunks_out].  prof_* = string literals.
int32_t ldb_wc_partition(ldb_ctx* ctx, const uint32_t* keys_in, const uint32_t* pay_in, uint64_t n, uint32_t bias, uint32_t range, uint32_t shift, uint32_t nparts, uint32_t* keys_out,
                         uint32_t* pay_out, uint32_t** part_offs_out, uint32_t* chunks_out, const char* prof_hist, const char* prof_scatter);
#define LDB_ARENA_WORDS 16384
// n_words zeroed 64-bit device words (64-byte aligned) that stay this caller's until the arena wraps: written by its kernels,
// read back with ldb_readback; never cleared, never reused for a second purpose by the caller after the read-back
int32_t ldb_counters(ldb_ctx* ctx, int n_words, uint64_t** out);
// selection bitmap (n_words x 64 rows) → ascending row numbers in out[0, total), total ≤ cap; entries [total, cap) are set
// to 0; with `match`, second[j] = match[out[j]]; *d_total (device, may be NULL) receives the count.  One launch.
int32_t ldb_bitmap_compact(ldb_ctx* ctx, const uint64_t* bitmap, int64_t n_words, uint32_t* out, uint64_t cap, const uint32_t* match, uint32_t* second, uint64_t* d_total);
// read of one 64-bit device word (ldb_readback)
int32_t ldb_read_u64_at(ldb_ctx* ctx, const void* d_word, uint64_t* out, uint32_t site, int flags = 0);
#define ldb_read_u64(ctx, d_word, out) ldb_read_u64_at((ctx), (d_word), (out), LDB_SITE)
// new relation helpers
ldb_rel* ldb_rel_new(ldb_ctx* ctx);
