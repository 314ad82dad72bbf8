// ldb_core.hip — context, device tables (Arrow C Data Interface in/out), relations, gather.
// Replaces (reference): ExecutionContext (include/lingodb/runtime/ExecutionContext.h:62-127),
// LingoDBTable::ensureLoaded + TableChunk flattening (src/runtime/storage/LingoDBTable.cpp:27-54,
// 200-225) and result materialisation (ArrowColumnBuilder, include/lingodb/runtime/ArrowColumn.h:15-37).
#include "ldb_internal.h"
#include "ldb_device.h"
#include "ldb_chain.h"
#include <algorithm>
#include <cstdlib>
#include <memory>

// ---------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
void ldb_set_error(const char* fmt, ...) {
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof(g_err), fmt, ap);
   va_end(ap);
}
extern "C" const char* ldb_gpu_last_error(void) { return g_err; }

// ---------------------------------------------------------------- options
// Process-wide tuning knobs: first read comes from the environment (LDB_<NAME>), later
// ldb_gpu_set_option calls override it — so a test can run the same process through both the
// generic and the run-time specialised / lazily fused code paths.
#include <atomic>
#include <functional>
#include <mutex>
static std::mutex g_opt_mu;
static std::atomic<int64_t> g_option_epoch{0}; // bumped by every ldb_gpu_set_option: a prepared plan recorded under other options is not replayed
static std::unordered_map<std::string, int64_t> g_opts;
int64_t ldb_option(const char* name, int64_t dflt) {
   std::lock_guard<std::mutex> lock(g_opt_mu);
   auto it = g_opts.find(name);
   if (it != g_opts.end()) return it->second;
   std::string env = "LDB_";
   for (const char* c = name; *c; c++) env += (char) toupper((unsigned char) *c);
   int64_t v = dflt;
   if (const char* e = getenv(env.c_str())) v = atoll(e);
   g_opts[name] = v;
   return v;
}
extern "C" int32_t ldb_gpu_set_option(const char* name, int64_t value) {
   if (!name) LDB_FAIL(LDB_ERR_INVALID, "set_option: NULL name");
   static const char* known[] = {"jit", "jit_min_rows", "lazy_filter", "lazy_min_rows", "join_ordered", "join_chained", "gb_ordered", "gb_sorted", "zone_maps", "zone_min_rows", "gb_direct", "gb_wgs_per_cu", "gb_partition", "gb_partition_min_rows", "join_radix", "join_radix_min_rows", "join_radix_min_table_bytes", "join_radix_part_bytes", "probe_batch", "debug_check", "join_direct", "join_rank", "join_coarse", "dict_encode", "dict_min_rows", "comm_transport", "comm_timeout_ms", "desc_cache", "desc_cache_mb", "plan_replay", "scan_single_pass", "lazy_strings", "lazy_strings_min_rows", "gb_partition_wc", "join_radix_wc", "join_radix_lds", "gb_dense_out", "gb_partition_values", "join_pair32", "gb_dense_keys", "jit_async", "jit_threads", "jit_disk_cache", "join_all_match", "topk_short_select", "gb_fits64", "join_coarse_fine", "join_coarse_filtered", "compact_wide_tiles", "compact_max_words", "join_fuse_max_conjuncts", "join_coarse_finest", "jit_min_rows_like", "scan_split", "jit_share_compiles"};
   bool ok = false;
   for (const char* k : known) ok |= strcmp(k, name) == 0;
   if (!ok) LDB_FAIL(LDB_ERR_INVALID, "set_option: unknown option '%s'", name);
   std::lock_guard<std::mutex> lock(g_opt_mu);
   g_opts[name] = value;
   g_option_epoch.fetch_add(1);
   return LDB_OK;
}
// the value in effect if the option was set or read before, else what LDB_<NAME> says, else -1 ("the
// use site's default"); never caches anything itself
extern "C" int64_t ldb_gpu_get_option(const char* name) {
   if (!name) return -1;
   std::lock_guard<std::mutex> lock(g_opt_mu);
   auto it = g_opts.find(name);
   if (it != g_opts.end()) return it->second;
   std::string env = "LDB_";
   for (const char* c = name; *c; c++) env += (char) toupper((unsigned char) *c);
   if (const char* e = getenv(env.c_str())) return atoll(e);
   return -1;
}
/* (ldb_gpu_set_option(name, v) with the use site's default restores the default behaviour) */

double ldb_host_trace_threshold() {
   static const double thr = [] {
      const char* e = getenv("LDB_HOST_TRACE");
      return e && *e ? atof(e) : -1.0;
   }();
   return thr;
}
// ---------------------------------------------------------------- memory
// size classes of the block cache: 8 per doubling (≤ 12.5 % internal waste), 256 B minimum
static size_t ldb_size_class(size_t bytes) {
   if (bytes <= 256) return 256;
   const int lg = 63 - __builtin_clzll((unsigned long long) (bytes - 1));
   const size_t step = (size_t) 1 << (lg >= 3 ? lg - 3 : 0);
   return (bytes + step - 1) / step * step;
}
static void ldb_cache_release(ldb_ctx* ctx) {
   for (auto& kv : ctx->parked)
      for (void* p : kv.second) (void) hipFreeAsync(p, ctx->stream);
   ctx->parked.clear();
   ctx->cache_bytes = 0;
}
int32_t ldb_dev_alloc(ldb_ctx* ctx, void** out, size_t bytes) {
   // 16 bytes of slack behind every buffer: string kernels read through an 8-byte window (d_bytes8)
   const size_t cls = ctx->cache_on ? ldb_size_class(bytes + 16) : bytes + 16;
   if (ctx->cache_on) {
      auto it = ctx->parked.find(cls);
      if (it != ctx->parked.end() && !it->second.empty()) {
         // the LOWEST parked address of the class (min-heap): which block an allocation gets then depends only on the set of
         // free blocks, not on the order they were freed in — a plan that runs again finds its buffers at the same addresses,
         // so its descriptors are byte-identical (ldb_dev_upload's cache) from the second execution on
         std::pop_heap(it->second.begin(), it->second.end(), std::greater<void*>());
         *out = it->second.back();
         it->second.pop_back();
         ctx->cache_bytes -= cls;
         ctx->live[*out] = cls;
         return LDB_OK;
      }
   }
   hipError_t e;
   {
      LdbSlow slow_("hipMallocAsync", cls);
      e = hipMallocAsync(out, cls, ctx->stream);
   }
   if (e != hipSuccess && ctx->cache_bytes) { // the parked blocks are the only memory we can give back
      (void) hipGetLastError();
      ldb_cache_release(ctx);
      (void) hipStreamSynchronize(ctx->stream);
      e = hipMallocAsync(out, cls, ctx->stream);
   }
   LDB_HIP(e);
   if (ctx->cache_on) ctx->live[*out] = cls;
   return LDB_OK;
}
void ldb_dev_free(ldb_ctx* ctx, void* p) {
   if (!p) return;
   auto dref = ctx->desc_blocks.find(p);
   if (dref != ctx->desc_blocks.end()) { // a cached descriptor (ldb_dev_upload): owned by the cache, this holder is done with it
      if (dref->second > 0) dref->second--;
      else {
         ctx->desc_underflows++; // nobody held it: the count stays at 0 (never negative), and another holder's reference is not touched
         if (ldb_host_trace_threshold() >= 0) fprintf(stderr, "[ldb] descriptor %p given back twice\n", p);
      }
      return;
   }
   if (!ctx->shared.empty()) {
      auto sh = ctx->shared.find(p);
      if (sh != ctx->shared.end()) { // another relation still reads this vector
         if (--sh->second == 0) ctx->shared.erase(sh);
         return;
      }
   }
   auto it = ctx->live.find(p);
   if (it == ctx->live.end()) {
      LdbSlow slow_("hipFreeAsync (untracked block)");
      (void) hipFreeAsync(p, ctx->stream);
      return;
   }
   const size_t cls = it->second;
   ctx->live.erase(it);
   if (ctx->cache_bytes + cls <= ctx->cache_cap) {
      auto& heap = ctx->parked[cls];
      heap.push_back(p);
      std::push_heap(heap.begin(), heap.end(), std::greater<void*>());
      ctx->cache_bytes += cls;
   } else {
      LdbSlow slow_("hipFreeAsync (cache full)", cls);
      (void) hipFreeAsync(p, ctx->stream);
   }
}
void ldb_dev_share(ldb_ctx* ctx, void* p) {
   if (p) ctx->shared[p]++;
}
// Descriptor cache.  A plan that runs again builds byte-identical descriptors (same operators over the same buffers: the
// block cache above hands out the same addresses), so the device copy made by the previous execution can be used as it is:
// no staging, no H2D copy, no allocation.  Entries are keyed by a 64-bit content hash and verified by memcmp; kernels take
// descriptors as `const D*`, nothing on the device ever writes one.  ldb_dev_free recognises a cached block and leaves it
// alone (it only gives its reference back).  Bounded (desc_cache_mb, default 64 MB): when full, every entry that no operator
// holds any more is dropped (stream-ordered frees, so launches already queued are unaffected).  Option desc_cache = 0 switches it off.
static uint64_t desc_hash(const void* p, size_t bytes) {
   const uint8_t* b = (const uint8_t*) p;
   uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
   size_t i = 0;
   for (; i + 8 <= bytes; i += 8) {
      uint64_t w;
      memcpy(&w, b + i, 8);
      h = (h ^ w) * 0xFF51AFD7ED558CCDull;
      h ^= h >> 29;
   }
   for (; i < bytes; i++) h = (h ^ b[i]) * 0x100000001B3ull;
   return h ^ (h >> 32);
}
// eviction: only entries no operator holds (reference count 0 — every ldb_dev_upload is paired with an ldb_dev_free of the pointer it
// returned); an entry that is still referenced keeps its device copy and its place in the cache, so a pointer handed out earlier never
// dangles and is never freed twice.  `force` (context teardown) drops everything.
static void desc_cache_drop(ldb_ctx* ctx, bool force = false) {
   for (auto it = ctx->desc_cache.begin(); it != ctx->desc_cache.end();) {
      auto ref = ctx->desc_blocks.find(it->second.dev);
      if (!force && ref != ctx->desc_blocks.end() && ref->second > 0) {
         ++it;
         continue;
      }
      (void) hipFreeAsync(it->second.dev, ctx->stream);
      ctx->desc_bytes -= std::min(ctx->desc_bytes, it->second.copy.size());
      if (ref != ctx->desc_blocks.end()) ctx->desc_blocks.erase(ref);
      it = ctx->desc_cache.erase(it);
   }
   if (force) {
      ctx->desc_blocks.clear();
      ctx->desc_bytes = 0;
   }
}
int32_t ldb_h2d_small(ldb_ctx* ctx, void* dev, const void* host, size_t bytes) {
   // descriptors (a few KB) are staged through a pinned ring so that the copy is a plain async DMA
   // (a pageable source makes the runtime stage it synchronously, ~10 µs per call); `host` may be a
   // local either way.  A slot is reused only after the stream has drained (wrap → synchronize).
   const size_t need = (bytes + 63) & ~(size_t) 63;
   if (ctx->h_ring && need <= LDB_RING_BYTES / 8) {
      if (ctx->ring_pos + need > LDB_RING_BYTES) {
         LdbSlow slow_("staging ring wrap: hipStreamSynchronize");
         LDB_HIP(hipStreamSynchronize(ctx->stream));
         ctx->ring_pos = 0;
      }
      void* slot = ctx->h_ring + ctx->ring_pos;
      ctx->ring_pos += need;
      memcpy(slot, host, bytes);
      LDB_HIP(hipMemcpyAsync(dev, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
      return LDB_OK;
   }
   LDB_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
   return LDB_OK;
}
int32_t ldb_dev_upload(ldb_ctx* ctx, const void* host, size_t bytes, void** dev_out, bool cacheable) {
   const bool cache_wanted = ldb_option("desc_cache", 1) != 0;
   if (cacheable && cache_wanted && bytes >= 64 && bytes <= (64u << 10)) {
      const uint64_t h = desc_hash(host, bytes);
      auto range = ctx->desc_cache.equal_range(h);
      for (auto it = range.first; it != range.second; ++it)
         if (it->second.copy.size() == bytes && memcmp(it->second.copy.data(), host, bytes) == 0) {
            *dev_out = it->second.dev;
            ctx->desc_blocks[it->second.dev]++;
            ctx->desc_hits++;
            return LDB_OK;
         }
      const size_t cap = (size_t) ldb_option("desc_cache_mb", 64) << 20;
      if (ctx->desc_bytes + bytes > cap) desc_cache_drop(ctx);
      void* dev = nullptr;
      LDB_TRY(ldb_dev_alloc(ctx, &dev, bytes)); // from the block cache; the cache entry takes it out of circulation
      ctx->live.erase(dev);
      int32_t st = ldb_h2d_small(ctx, dev, host, bytes);
      if (st != LDB_OK) {
         (void) hipFreeAsync(dev, ctx->stream);
         return st;
      }
      ldb_ctx::DescEntry e;
      e.dev = dev;
      e.copy.assign((const uint8_t*) host, (const uint8_t*) host + bytes);
      ctx->desc_cache.emplace(h, std::move(e));
      ctx->desc_blocks[dev] = 1;
      ctx->desc_bytes += bytes;
      ctx->desc_misses++;
      *dev_out = dev;
      return LDB_OK;
   }
   LDB_TRY(ldb_dev_alloc(ctx, dev_out, bytes));
   return ldb_h2d_small(ctx, *dev_out, host, bytes);
}

// ---------------------------------------------------------------- read-backs (ldb_internal.h: ldb_readback)
// copies `bytes` (a multiple of 4 after padding) from device memory into the pinned log; one wave
__global__ void k_log_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t bytes) {
   for (uint32_t i = threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
}
// the deferred copies of a replaying trace: one kernel gathers every pending (source, bytes) into the pinned log
struct DLogList {
   uint32_t n, pad;
   ldb_ctx::LogCopy e[255];
};
__global__ void k_log_gather(const DLogList* __restrict__ l, uint8_t* __restrict__ log) {
   for (uint32_t i = 0; i < l->n; i++) {
      const uint8_t* src = (const uint8_t*) l->e[i].src;
      for (uint32_t b = threadIdx.x; b < l->e[i].bytes; b += blockDim.x) log[l->e[i].off + b] = src[b];
   }
}
static int32_t log_flush(ldb_ctx* ctx) {
   size_t at = 0;
   while (at < ctx->log_pending.size()) {
      auto l = std::make_unique<DLogList>();
      memset(l.get(), 0, sizeof(DLogList));
      l->n = (uint32_t) std::min<size_t>(255, ctx->log_pending.size() - at);
      for (uint32_t i = 0; i < l->n; i++) l->e[i] = ctx->log_pending[at + i];
      at += l->n;
      LdbDesc<DLogList> d(ctx);
      LDB_TRY(d.upload(l.get(), 8 + sizeof(ldb_ctx::LogCopy) * l->n)); // (the same list every execution: served by the descriptor cache)
      hipLaunchKernelGGL(k_log_gather, dim3(1), dim3(64), 0, ctx->stream, (const DLogList*) d.p, ctx->h_log);
   }
   ctx->log_pending.clear();
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}
int32_t ldb_counters(ldb_ctx* ctx, int n_words, uint64_t** out) {
   const size_t need = ((size_t) n_words + 7) & ~(size_t) 7;
   if (!ctx->arena) {
      LDB_HIP(hipMalloc((void**) &ctx->arena, 8 * LDB_ARENA_WORDS));
      LDB_HIP(hipMemsetAsync(ctx->arena, 0, 8 * LDB_ARENA_WORDS, ctx->stream));
      ctx->arena_pos = 0;
   }
   if (need > LDB_ARENA_WORDS) LDB_FAIL(LDB_ERR_INVALID, "ldb_counters: %d words", n_words);
   if (ctx->arena_pos + need > LDB_ARENA_WORDS) { // wrap: what a replaying trace still wants from the old words is collected first
      LDB_TRY(log_flush(ctx));
      // words handed out earlier may not have been read back yet (a caller that allocates, launches and meets a nested allocation
      // before its read-back): nothing is cleared wholesale — from here until the next restart every allocation clears its own range
      ctx->arena_wrapped = true;
      ctx->arena_pos = 0;
   }
   *out = ctx->arena + ctx->arena_pos;
   if (ctx->arena_wrapped) LDB_HIP(hipMemsetAsync(*out, 0, 8 * need, ctx->stream));
   ctx->arena_pos += need;
   return LDB_OK;
}
// a plan starts at word 0 (the same words every execution → byte-identical descriptors): one clear of what the last one used
static int32_t arena_restart(ldb_ctx* ctx) {
   if (ctx->arena && (ctx->arena_pos || ctx->arena_wrapped)) LDB_HIP(hipMemsetAsync(ctx->arena, 0, 8 * (ctx->arena_wrapped ? (size_t) LDB_ARENA_WORDS : ctx->arena_pos), ctx->stream));
   ctx->arena_pos = 0;
   ctx->arena_wrapped = false;
   return LDB_OK;
}
static int32_t readback_sync(ldb_ctx* ctx, void* host, const void* dev, size_t bytes) {
   LdbSlow slow_("read-back with a stream wait", bytes);
   if (bytes <= 512 && ctx->h_scratch) { // a pinned landing area: the copy is a plain DMA / blit, no staging by the runtime
      LDB_HIP(hipMemcpyAsync(ctx->h_scratch, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
      LDB_HIP(hipStreamSynchronize(ctx->stream));
      memcpy(host, ctx->h_scratch, bytes);
      return LDB_OK;
   }
   LDB_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   return LDB_OK;
}
static std::mutex g_site_mu;
static std::unordered_map<uint32_t, std::string> g_sites;
uint32_t ldb_site_note(uint32_t hash, const char* file, int line) {
   std::lock_guard<std::mutex> lock(g_site_mu);
   auto it = g_sites.find(hash);
   if (it == g_sites.end()) {
      const char* base = strrchr(file, '/');
      g_sites.emplace(hash, std::string(base ? base + 1 : file) + ":" + std::to_string(line));
   }
   return hash;
}
uint32_t ldb_site_derived(uint32_t base, uint32_t salt) {
   const uint32_t hash = base ^ (salt * 2654435761u);
   if (ldb_host_trace_threshold() < 0) return hash; // (names are only ever printed under LDB_HOST_TRACE)
   std::lock_guard<std::mutex> lock(g_site_mu);
   if (!g_sites.count(hash)) {
      auto it = g_sites.find(base);
      g_sites.emplace(hash, (it != g_sites.end() ? it->second : std::string("?")) + " n=" + std::to_string(salt));
   }
   return hash;
}
static std::atomic<int64_t> g_order_misses{0};
extern "C" int64_t ldb_gpu_order_dependent_misses(void) { return g_order_misses.load(); }
// replay: everything read so far must equal the record
static bool trace_prefix_ok(ldb_ctx* ctx, size_t upto_entries) {
   LdbSlow slow_("trace check: wait for the replayed plan", upto_entries);
   if (log_flush(ctx) != LDB_OK) return false;
   if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
   const ldb_trace* t = ctx->trace;
   for (size_t i = 0; i < upto_entries; i++) {
      const ldb_trace_entry& e = t->entries[i];
      if (memcmp(ctx->h_log + e.off, t->vals.data() + e.off, e.bytes) != 0) {
         if (e.flags & LDB_RB_ORDER_DEPENDENT) g_order_misses.fetch_add(1);
         if (ldb_host_trace_threshold() >= 0) { // which read-back was it, and what did it say
            std::lock_guard<std::mutex> lock(g_site_mu);
            auto it = g_sites.find(e.site);
            unsigned long long rec[2] = {0, 0}, got[2] = {0, 0};
            memcpy(rec, t->vals.data() + e.off, std::min<size_t>(16, e.bytes));
            memcpy(got, ctx->h_log + e.off, std::min<size_t>(16, e.bytes));
            fprintf(stderr, "[ldb host] replayed read-back %zu of %zu differs (%s, %u bytes): recorded %llx %llx, now %llx %llx\n", i, t->entries.size(),
                    it != g_sites.end() ? it->second.c_str() : "?", e.bytes, rec[0], rec[1], got[0], got[1]);
         }
         return false;
      }
   }
   return true;
}
int32_t ldb_readback(ldb_ctx* ctx, void* host, const void* dev, size_t bytes, uint32_t site, int flags) {
   if (bytes == 0) return LDB_OK;
   if (ctx->trace_mode == 0 || bytes > LDB_RB_MAX) return readback_sync(ctx, host, dev, bytes);
   // a mis-speculated execution is void.  Alone, the rank stops at once (LDB_ERR_RETRY); inside a COLLECTIVE trace (a plan that exchanges rows
   // with other ranks) it runs on over the recorded counts — its peers are queued against exactly those transfer sizes and would wait for ever
   // for a rank that left — and the verdict at the trace's end makes every rank repeat the execution
   if (ctx->trace_poisoned && !ctx->trace_collective) LDB_FAIL(LDB_ERR_RETRY, "read-back after a mis-speculated value (the plan execution is being repeated)");
   ldb_trace* t = ctx->trace;
   if (flags & LDB_RB_NEVER_REPLAY) { // not a function of the data alone: read for real, in both modes, and keep out of the record
      if (ctx->trace_mode == 2 && !ctx->trace_poisoned && !trace_prefix_ok(ctx, ctx->trace_pos)) {
         ctx->trace_poisoned = true;
         t->misses++;
         if (!ctx->trace_collective) LDB_FAIL(LDB_ERR_RETRY, "a replayed read-back differs from the recorded value");
      }
      return readback_sync(ctx, host, dev, bytes);
   }
   if (ctx->trace_mode == 2) {
      if (ctx->trace_pos < t->entries.size() && t->entries[ctx->trace_pos].site == site && t->entries[ctx->trace_pos].bytes == bytes) {
         const ldb_trace_entry& e = t->entries[ctx->trace_pos++];
         const uint64_t* w = (const uint64_t*) dev;
         if (ctx->arena && w >= ctx->arena && w < ctx->arena + LDB_ARENA_WORDS) { // an arena word keeps its value until the plan ends: collected with the others
            ctx->log_pending.push_back({(uint64_t) dev, e.off, (uint32_t) bytes});
         } else {
            hipLaunchKernelGGL(k_log_copy, dim3(1), dim3(64), 0, ctx->stream, (const uint8_t*) dev, ctx->h_log + e.off, (uint32_t) bytes);
            LDB_HIP(hipGetLastError());
         }
         memcpy(host, t->vals.data() + e.off, bytes);
         return LDB_OK;
      }
      // the execution took another path than the recorded one (a statistic is cached now, an option changed …): what was
      // replayed so far is checked, and from here on the execution records
      if (ctx->trace_poisoned || !trace_prefix_ok(ctx, ctx->trace_pos)) {
         if (!ctx->trace_poisoned) t->misses++;
         ctx->trace_poisoned = true;
         if (!ctx->trace_collective) LDB_FAIL(LDB_ERR_RETRY, "a replayed read-back differs from the recorded value");
         // collective: a rank on void data easily leaves the recorded sequence (an empty input skips a read, a table overflows …).  Its peers are
         // still inside their exchanges, queued against this rank's transfers: abandoning the plan here would stall them until the communicator's
         // time-out (shm) or for good (RCCL).  It reads the real value — consistent with what is on the device now — leaves the record alone and
         // goes on to the end, where the agreed verdict makes every rank repeat the execution
         return readback_sync(ctx, host, dev, bytes);
      }
      t->diverged++;
      t->entries.resize(ctx->trace_pos);
      t->vals.resize(ctx->trace_pos ? t->entries.back().off + ((t->entries.back().bytes + 7) & ~7u) : 0);
      ctx->trace_mode = 1;
   }
   LDB_TRY(readback_sync(ctx, host, dev, bytes));
   const size_t off = t->vals.size();
   if (off + bytes + 8 <= LDB_LOG_BYTES) {
      t->entries.push_back({site, (uint32_t) bytes, (uint32_t) off, (uint32_t) (flags & LDB_RB_ORDER_DEPENDENT)});
      t->vals.resize(off + ((bytes + 7) & ~(size_t) 7), 0);
      memcpy(t->vals.data() + off, host, bytes);
   } else {
      t->complete = false;
      ctx->trace_mode = 0; // more read-backs than the log holds: this plan is never replayed
      t->entries.clear();
      t->vals.clear();
   }
   return LDB_OK;
}
int32_t ldb_read_u64_at(ldb_ctx* ctx, const void* d_word, uint64_t* out, uint32_t site, int flags) { return ldb_readback(ctx, out, d_word, 8, site, flags); }

extern "C" int32_t ldb_gpu_trace_create(ldb_ctx* ctx, ldb_trace** out) {
   if (!ctx || !out) LDB_FAIL(LDB_ERR_INVALID, "trace_create: NULL argument");
   *out = new ldb_trace();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_trace_destroy(ldb_ctx* ctx, ldb_trace* t) {
   if (ctx && ctx->trace == t) {
      ctx->trace = nullptr;
      ctx->trace_mode = 0;
   }
   delete t;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_trace_begin(ldb_ctx* ctx, ldb_trace* t, int32_t allow_replay) {
   if (!ctx || !t) LDB_FAIL(LDB_ERR_INVALID, "trace_begin: NULL argument");
   if (ctx->trace_mode != 0) LDB_FAIL(LDB_ERR_INVALID, "trace_begin: a trace is already active on this context");
   if (!ctx->h_log) LDB_HIP(hipHostMalloc((void**) &ctx->h_log, LDB_LOG_BYTES, hipHostMallocDefault));
   const bool replay_wanted = ldb_option("plan_replay", 1) != 0;
   ctx->trace = t;
   ctx->trace_pos = 0;
   ctx->trace_poisoned = false;
   ctx->trace_collective = (allow_replay & LDB_TRACE_COLLECTIVE) != 0;
   ctx->log_pending.clear();
   LDB_TRY(arena_restart(ctx));
   if ((allow_replay & 1) && replay_wanted && t->complete && !t->entries.empty()) {
      ctx->trace_mode = 2;
   } else {
      ctx->trace_mode = 1;
      t->entries.clear();
      t->vals.clear();
   }
   t->complete = false;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_trace_end(ldb_ctx* ctx, int32_t* status) {
   if (!ctx || !status) LDB_FAIL(LDB_ERR_INVALID, "trace_end: NULL argument");
   ldb_trace* t = ctx->trace;
   const int mode = ctx->trace_mode;
   ctx->trace = nullptr;
   ctx->trace_mode = 0;
   ctx->trace_collective = false;
   if (!t || mode == 0) {
      *status = LDB_TRACE_OFF;
      if (t) t->complete = false;
      return LDB_OK;
   }
   if (ctx->trace_poisoned) {
      ctx->trace_poisoned = false;
      t->complete = false;
      t->entries.clear();
      t->vals.clear();
      (void) hipStreamSynchronize(ctx->stream);
      *status = LDB_TRACE_MISSED;
      return LDB_OK;
   }
   if (mode == 2) {
      ctx->trace = t;
      const bool ok = trace_prefix_ok(ctx, ctx->trace_pos);
      ctx->trace = nullptr;
      if (!ok) {
         t->misses++;
         t->complete = false;
         t->entries.clear();
         t->vals.clear();
         *status = LDB_TRACE_MISSED;
         return LDB_OK;
      }
      if (ctx->trace_pos < t->entries.size()) { // ended earlier than the record: keep what was used
         t->entries.resize(ctx->trace_pos);
         t->vals.resize(ctx->trace_pos ? t->entries.back().off + ((t->entries.back().bytes + 7) & ~7u) : 0);
      }
      t->replays++;
      t->complete = true;
      *status = LDB_TRACE_REPLAYED;
      return LDB_OK;
   }
   t->records++;
   t->complete = true;
   *status = LDB_TRACE_RECORDED;
   return LDB_OK;
}
// would ldb_gpu_trace_begin(ctx, t, 1) replay?  (what the ranks of a sharded plan agree on BEFORE any of them begins: they replay together or not at all)
extern "C" int32_t ldb_gpu_trace_replayable(const ldb_trace* t) { return t && t->complete && !t->entries.empty() && ldb_option("plan_replay", 1) != 0 ? 1 : 0; }
extern "C" int32_t ldb_gpu_trace_stats(const ldb_trace* t, int64_t* entries, int64_t* records, int64_t* replays, int64_t* misses) {
   if (!t) LDB_FAIL(LDB_ERR_INVALID, "trace_stats: NULL trace");
   if (entries) *entries = (int64_t) t->entries.size();
   if (records) *records = t->records;
   if (replays) *replays = t->replays;
   if (misses) *misses = t->misses;
   return LDB_OK;
}
static std::atomic<uint64_t> g_serial{1};
uint64_t ldb_next_serial() { return g_serial.fetch_add(1); }
extern "C" uint64_t ldb_gpu_table_stamp(const ldb_table* t) { return t ? t->serial : 0; }
extern "C" int64_t ldb_gpu_option_epoch(void) { return g_option_epoch.load(); }
extern "C" int32_t ldb_gpu_desc_cache_held(ldb_ctx* ctx, int64_t* held, int64_t* underflows) {
   if (!ctx) LDB_FAIL(LDB_ERR_INVALID, "desc_cache_held: NULL ctx");
   int64_t n = 0;
   for (auto& b : ctx->desc_blocks) n += b.second;
   if (held) *held = n;
   if (underflows) *underflows = ctx->desc_underflows;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_desc_cache_stats(ldb_ctx* ctx, int64_t* hits, int64_t* misses, int64_t* bytes) {
   if (!ctx) LDB_FAIL(LDB_ERR_INVALID, "desc_cache_stats: NULL ctx");
   if (hits) *hits = ctx->desc_hits;
   if (misses) *misses = ctx->desc_misses;
   if (bytes) *bytes = (int64_t) ctx->desc_bytes;
   return LDB_OK;
}

// ---------------------------------------------------------------- context
extern "C" int32_t ldb_gpu_ctx_create(int32_t device_id, void* stream, ldb_ctx** out) {
   if (!out) LDB_FAIL(LDB_ERR_INVALID, "ctx_create: out is NULL");
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess || n == 0) LDB_FAIL(LDB_ERR_NO_DEVICE, "no HIP device visible (liblingodb_gpu has no CPU fallback)");
   if (device_id < 0 || device_id >= n) LDB_FAIL(LDB_ERR_INVALID, "device %d out of range (have %d)", device_id, n);
   LDB_HIP(hipSetDevice(device_id));
   auto ctx = std::make_unique<ldb_ctx>();
   ctx->device = device_id;
   if (stream) {
      ctx->stream = (hipStream_t) stream;
   } else {
      LDB_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = true;
   }
   hipDeviceProp_t prop;
   LDB_HIP(hipGetDeviceProperties(&prop, device_id));
   ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
   // keep freed blocks cached in the stream-ordered pool (operators allocate per call)
   hipMemPool_t pool;
   if (hipDeviceGetDefaultMemPool(&pool, device_id) == hipSuccess) {
      uint64_t thr = UINT64_MAX;
      (void) hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
   }
   const char* cache_env = getenv("LDB_ALLOC_CACHE");
   ctx->cache_on = !(cache_env && cache_env[0] == '0');
   size_t hbm_free = 0, hbm_total = 0;
   if (hipMemGetInfo(&hbm_free, &hbm_total) == hipSuccess) ctx->cache_cap = hbm_total / 4;
   LDB_HIP(hipHostMalloc((void**) &ctx->h_scratch, 64 * sizeof(int64_t), hipHostMallocDefault));
   LDB_HIP(hipHostMalloc((void**) &ctx->h_ring, LDB_RING_BYTES, hipHostMallocDefault));
   LDB_HIP(hipMalloc((void**) &ctx->d_scratch, 64 * sizeof(int64_t)));
   *out = ctx.release();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_ctx_destroy(ldb_ctx* ctx) {
   if (!ctx) return LDB_OK;
   (void) hipSetDevice(ctx->device);
   desc_cache_drop(ctx, true);
   ldb_cache_release(ctx);
   (void) hipStreamSynchronize(ctx->stream);
   for (auto e : ctx->timers) (void) hipEventDestroy(e);
   for (auto& p : ctx->prof_pending) {
      (void) hipEventDestroy(p.start);
      (void) hipEventDestroy(p.stop);
   }
   for (auto e : ctx->prof_free) (void) hipEventDestroy(e);
   if (ctx->h_scratch) (void) hipHostFree(ctx->h_scratch);
   if (ctx->h_ring) (void) hipHostFree(ctx->h_ring);
   if (ctx->h_log) (void) hipHostFree(ctx->h_log);
   if (ctx->d_scratch) (void) hipFree(ctx->d_scratch);
   if (ctx->scan_status) (void) hipFree(ctx->scan_status);
   if (ctx->scan_ticket) (void) hipFree(ctx->scan_ticket);
   if (ctx->arena) (void) hipFree(ctx->arena);
   if (ctx->own_stream) (void) hipStreamDestroy(ctx->stream);
   delete ctx;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_ctx_sync(ldb_ctx* ctx) {
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_device_info(ldb_ctx* ctx, char* name, int32_t name_cap, int32_t* cus, int64_t* hbm_free, int64_t* hbm_total) {
   hipDeviceProp_t prop;
   LDB_HIP(hipGetDeviceProperties(&prop, ctx->device));
   if (name && name_cap > 0) snprintf(name, (size_t) name_cap, "%s (%s)", prop.name, prop.gcnArchName);
   if (cus) *cus = prop.multiProcessorCount;
   size_t f = 0, t = 0;
   LDB_HIP(hipMemGetInfo(&f, &t));
   if (hbm_free) *hbm_free = (int64_t) f;
   if (hbm_total) *hbm_total = (int64_t) t;
   return LDB_OK;
}

__global__ void k_ldb_marker() {}
extern "C" int32_t ldb_gpu_prof_marker(ldb_ctx* ctx, int32_t id) {
   if (!ctx || id < 1 || id > 65535) LDB_FAIL(LDB_ERR_INVALID, "prof_marker: id must be in 1..65535");
   hipLaunchKernelGGL(k_ldb_marker, dim3((unsigned) id), dim3(64), 0, ctx->stream);
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_timer_create(ldb_ctx* ctx, int32_t* timer_id) {
   hipEvent_t a, b;
   LDB_HIP(hipEventCreate(&a));
   LDB_HIP(hipEventCreate(&b));
   *timer_id = (int32_t) (ctx->timers.size() / 2);
   ctx->timers.push_back(a);
   ctx->timers.push_back(b);
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_timer_start(ldb_ctx* ctx, int32_t id) {
   if (id < 0 || (size_t) id * 2 + 1 >= ctx->timers.size() + 0) LDB_FAIL(LDB_ERR_INVALID, "bad timer id %d", id);
   LDB_HIP(hipEventRecord(ctx->timers[(size_t) id * 2], ctx->stream));
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_timer_stop(ldb_ctx* ctx, int32_t id) {
   if (id < 0 || (size_t) id * 2 + 1 >= ctx->timers.size() + 0) LDB_FAIL(LDB_ERR_INVALID, "bad timer id %d", id);
   LDB_HIP(hipEventRecord(ctx->timers[(size_t) id * 2 + 1], ctx->stream));
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_timer_elapsed_ms(ldb_ctx* ctx, int32_t id, float* ms) {
   if (id < 0 || (size_t) id * 2 + 1 >= ctx->timers.size() + 0) LDB_FAIL(LDB_ERR_INVALID, "bad timer id %d", id);
   LDB_HIP(hipEventSynchronize(ctx->timers[(size_t) id * 2 + 1]));
   LDB_HIP(hipEventElapsedTime(ms, ctx->timers[(size_t) id * 2], ctx->timers[(size_t) id * 2 + 1]));
   return LDB_OK;
}

// ---------------------------------------------------------------- per-kernel profiling
static hipEvent_t prof_event(ldb_ctx* ctx) {
   if (!ctx->prof_free.empty()) {
      hipEvent_t e = ctx->prof_free.back();
      ctx->prof_free.pop_back();
      return e;
   }
   hipEvent_t e = nullptr;
   (void) hipEventCreate(&e);
   return e;
}
LdbProf::LdbProf(ldb_ctx* c, const char* name) : ctx(c) {
   if (!c->prof_on) return;
   ldb_prof_pending p{name, prof_event(c), prof_event(c)};
   if (!p.start || !p.stop) return;
   (void) hipEventRecord(p.start, c->stream);
   idx = c->prof_pending.size();
   c->prof_pending.push_back(p);
   active = true;
}
LdbProf::~LdbProf() {
   if (active) (void) hipEventRecord(ctx->prof_pending[idx].stop, ctx->stream);
}
static void prof_fold(ldb_ctx* ctx) {
   if (ctx->prof_pending.empty()) return;
   (void) hipStreamSynchronize(ctx->stream);
   for (auto& p : ctx->prof_pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
         ldb_prof_total* t = nullptr;
         for (auto& x : ctx->prof_totals)
            if (x.name == p.name) t = &x;
         if (!t) {
            ctx->prof_totals.push_back({p.name, 0, 0, 0});
            t = &ctx->prof_totals.back();
         }
         t->launches++;
         t->ms += ms;
         if (ms > t->max_ms) t->max_ms = ms;
      }
      ctx->prof_free.push_back(p.start);
      ctx->prof_free.push_back(p.stop);
   }
   ctx->prof_pending.clear();
}
extern "C" int32_t ldb_gpu_prof_enable(ldb_ctx* ctx, int32_t on) {
   if (!ctx) LDB_FAIL(LDB_ERR_INVALID, "prof_enable: NULL ctx");
   prof_fold(ctx);
   ctx->prof_on = on != 0;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_prof_reset(ldb_ctx* ctx) {
   if (!ctx) LDB_FAIL(LDB_ERR_INVALID, "prof_reset: NULL ctx");
   prof_fold(ctx);
   ctx->prof_totals.clear();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_prof_get(ldb_ctx* ctx, const char* kernel_name, int64_t* launches, double* total_ms) {
   if (!ctx || !kernel_name) LDB_FAIL(LDB_ERR_INVALID, "prof_get: NULL argument");
   prof_fold(ctx);
   if (launches) *launches = 0;
   if (total_ms) *total_ms = 0;
   for (auto& x : ctx->prof_totals)
      if (x.name == kernel_name) {
         if (launches) *launches = x.launches;
         if (total_ms) *total_ms = x.ms;
      }
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_prof_get_max(ldb_ctx* ctx, const char* kernel_name, double* max_ms) {
   if (!ctx || !kernel_name || !max_ms) LDB_FAIL(LDB_ERR_INVALID, "prof_get_max: NULL argument");
   prof_fold(ctx);
   *max_ms = 0;
   for (auto& x : ctx->prof_totals)
      if (x.name == kernel_name) *max_ms = x.max_ms;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_prof_names(ldb_ctx* ctx, char* buf, int32_t cap) {
   if (!ctx || !buf || cap < 1) LDB_FAIL(LDB_ERR_INVALID, "prof_names: bad argument");
   prof_fold(ctx);
   std::string s;
   for (auto& x : ctx->prof_totals) s += x.name + "\n";
   snprintf(buf, (size_t) cap, "%s", s.c_str());
   return LDB_OK;
}

// ---------------------------------------------------------------- tables
int32_t ldb_width_of(const ldb_coltype& t, int narrow) {
   switch (t.type) {
      case LDB_T_INT8:
      case LDB_T_BOOL8: return 1;
      case LDB_T_INT16: return 2;
      case LDB_T_INT32:
      case LDB_T_DATE32:
      case LDB_T_CHAR4:
      case LDB_T_FLOAT32: return 4;
      case LDB_T_INT64:
      case LDB_T_FLOAT64: return 8;
      case LDB_T_DECIMAL128: return (narrow && t.precision < 19) ? 8 : 16;
      default: return 0;
   }
}

// narrow == 2 (round 6, "compressed resident format"): a column whose values fit is stored at the narrowest of 1 / 2 / 4 / 8 bytes — decimals of
// precision < 19 (the generated code truncates those to 64 bits anyway, LowerToStd.cpp:128-132) and char(1) (one byte when every value is ASCII).
// Kernels widen in registers (d_load_i64 dispatches on the column's width) and compute in i64 / i128 exactly as before: results are bit-identical.
int32_t ldb_narrowest_width(int64_t lo, int64_t hi) {
   if (lo >= -128 && hi <= 127) return 1;
   if (lo >= -32768 && hi <= 32767) return 2;
   if (lo >= INT32_MIN && hi <= INT32_MAX) return 4;
   return 8;
}
bool ldb_type_narrows(const ldb_coltype& t) { return (t.type == LDB_T_DECIMAL128 && t.precision < 19) || t.type == LDB_T_CHAR4; }

static int parse_format(const char* f, ldb_coltype* t, bool* large) {
   *large = false;
   t->precision = t->scale = 0;
   if (!strcmp(f, "c")) t->type = LDB_T_INT8;
   else if (!strcmp(f, "s")) t->type = LDB_T_INT16;
   else if (!strcmp(f, "i")) t->type = LDB_T_INT32;
   else if (!strcmp(f, "l")) t->type = LDB_T_INT64;
   else if (!strcmp(f, "tdD")) t->type = LDB_T_DATE32;
   else if (!strcmp(f, "g")) t->type = LDB_T_FLOAT64;
   else if (!strcmp(f, "f")) t->type = LDB_T_FLOAT32;
   else if (!strcmp(f, "u")) t->type = LDB_T_UTF8;
   else if (!strcmp(f, "U")) {
      t->type = LDB_T_UTF8;
      *large = true;
   } else if (!strcmp(f, "w:4")) t->type = LDB_T_CHAR4;
   else if (!strncmp(f, "d:", 2)) {
      int p = 0, s = 0, bits = 128;
      int n = sscanf(f + 2, "%d,%d,%d", &p, &s, &bits);
      if (n < 2 || bits != 128) return -1;
      t->type = LDB_T_DECIMAL128;
      t->precision = p;
      t->scale = s;
   } else
      return -1;
   return 0;
}

extern "C" int32_t ldb_gpu_table_register(ldb_ctx* ctx, const char* name, struct ArrowSchema* schema, struct ArrowArray** batches,
                                          int64_t n_batches, int32_t narrow, ldb_table** out) {
   if (!ctx || !schema || !out) LDB_FAIL(LDB_ERR_INVALID, "table_register: NULL argument");
   if (strcmp(schema->format, "+s")) LDB_FAIL(LDB_ERR_INVALID, "table_register: schema must be a struct (+s), got %s", schema->format);
   auto t = std::make_unique<ldb_table>();
   t->ctx = ctx;
   t->name = name ? name : "";
   int64_t rows = 0;
   for (int64_t b = 0; b < n_batches; b++) {
      if (batches[b]->n_children != schema->n_children) LDB_FAIL(LDB_ERR_INVALID, "batch %ld has %ld children, schema %ld", (long) b, (long) batches[b]->n_children, (long) schema->n_children);
      // a sliced struct array (parent offset / length) selects rows [offset, offset + length) of every child
      for (int64_t c = 0; c < schema->n_children; c++) {
         const struct ArrowArray* ch = batches[b]->children[c];
         if (batches[b]->offset < 0 || batches[b]->length < 0 || ch->length < batches[b]->offset + batches[b]->length)
            LDB_FAIL(LDB_ERR_INVALID, "batch %ld: child %ld has %ld rows, the struct slice needs %ld", (long) b, (long) c, (long) ch->length, (long) (batches[b]->offset + batches[b]->length));
      }
      rows += batches[b]->length;
   }
   if (rows >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "table_register: %ld rows exceed uint32 row ids", (long) rows);
   t->n_rows = rows;
   t->cols.resize((size_t) schema->n_children);
   for (int64_t c = 0; c < schema->n_children; c++) {
      ldb_column& col = t->cols[(size_t) c];
      struct ArrowSchema* cs = schema->children[c];
      col.name = cs->name ? cs->name : "";
      bool large = false;
      if (parse_format(cs->format, &col.type, &large)) LDB_FAIL(LDB_ERR_UNSUPPORTED, "column %s: Arrow format '%s' not supported", col.name.c_str(), cs->format);
      col.type.nullable = (cs->flags & ARROW_FLAG_NULLABLE) ? 1 : 0;
      col.width = ldb_width_of(col.type, narrow);
      // validity: only materialised when some batch really has nulls
      bool any_null = false;
      for (int64_t b = 0; b < n_batches; b++) {
         struct ArrowArray* a = batches[b]->children[c];
            const int64_t alen = batches[b]->length, aoff = batches[b]->offset + a->offset; // the parent's slice applies to every child
         if (a->null_count != 0 && a->buffers[0]) {
            const uint8_t* v = (const uint8_t*) a->buffers[0];
            for (int64_t i = 0; i < alen && !any_null; i++) {
               int64_t k = i + aoff;
               if (!((v[k >> 3] >> (k & 7)) & 1)) any_null = true;
            }
         }
      }
      if (any_null) {
         std::vector<uint8_t> bm((size_t) ((rows + 7) / 8), 0);
         int64_t pos = 0;
         for (int64_t b = 0; b < n_batches; b++) {
            struct ArrowArray* a = batches[b]->children[c];
            const int64_t alen = batches[b]->length, aoff = batches[b]->offset + a->offset; // the parent's slice applies to every child
            const uint8_t* v = (const uint8_t*) a->buffers[0];
            for (int64_t i = 0; i < alen; i++, pos++) {
               int64_t k = i + aoff;
               bool ok = !v || ((v[k >> 3] >> (k & 7)) & 1);
               if (ok) bm[(size_t) (pos >> 3)] |= (uint8_t) (1u << (pos & 7));
               else col.null_count++;
            }
         }
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.validity, bm.size()));
         LDB_HIP(hipMemcpyAsync(col.validity, bm.data(), bm.size(), hipMemcpyHostToDevice, ctx->stream));
         LDB_HIP(hipStreamSynchronize(ctx->stream));
      }
      if (col.type.type == LDB_T_UTF8) {
         std::vector<int64_t> offs((size_t) rows + 1);
         int64_t pos = 0, bytes = 0;
         offs[0] = 0;
         for (int64_t b = 0; b < n_batches; b++) {
            struct ArrowArray* a = batches[b]->children[c];
            const int64_t alen = batches[b]->length, aoff = batches[b]->offset + a->offset; // the parent's slice applies to every child
            for (int64_t i = 0; i < alen; i++) {
               int64_t lo, hi;
               if (large) {
                  const int64_t* o = (const int64_t*) a->buffers[1] + aoff;
                  lo = o[i];
                  hi = o[i + 1];
               } else {
                  const int32_t* o = (const int32_t*) a->buffers[1] + aoff;
                  lo = o[i];
                  hi = o[i + 1];
               }
               bytes += hi - lo;
               offs[(size_t) ++pos] = bytes;
            }
         }
         col.value_bytes = bytes;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) bytes));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.offsets, sizeof(int64_t) * ((size_t) rows + 1)));
         LDB_HIP(hipMemcpyAsync(col.offsets, offs.data(), sizeof(int64_t) * ((size_t) rows + 1), hipMemcpyHostToDevice, ctx->stream));
         int64_t dpos = 0;
         for (int64_t b = 0; b < n_batches; b++) {
            struct ArrowArray* a = batches[b]->children[c];
            const int64_t alen = batches[b]->length, aoff = batches[b]->offset + a->offset; // the parent's slice applies to every child
            if (alen == 0) continue;
            int64_t lo, hi;
            if (large) {
               const int64_t* o = (const int64_t*) a->buffers[1] + aoff;
               lo = o[0];
               hi = o[alen];
            } else {
               const int32_t* o = (const int32_t*) a->buffers[1] + aoff;
               lo = o[0];
               hi = o[alen];
            }
            if (hi > lo) LDB_HIP(hipMemcpyAsync((uint8_t*) col.values + dpos, (const uint8_t*) a->buffers[2] + lo, (size_t) (hi - lo), hipMemcpyHostToDevice, ctx->stream));
            dpos += hi - lo;
         }
         LDB_HIP(hipStreamSynchronize(ctx->stream));
      } else {
         int src_w = ldb_width_of(col.type, 0);
         if (narrow >= 2 && ldb_type_narrows(col.type)) { // the narrowest width the column's value range allows (NULL slots count too: they only ever widen it)
            int64_t lo = INT64_MAX, hi = INT64_MIN;
            for (int64_t b = 0; b < n_batches; b++) {
               struct ArrowArray* a = batches[b]->children[c];
               const int64_t alen = batches[b]->length, aoff = batches[b]->offset + a->offset;
               const uint8_t* src = (const uint8_t*) a->buffers[1] + aoff * src_w;
               for (int64_t i = 0; i < alen; i++) {
                  int64_t v = 0;
                  if (src_w == 16) {
                     memcpy(&v, src + i * 16, 8);
                  } else {
                     int32_t w32;
                     memcpy(&w32, src + i * 4, 4);
                     v = w32;
                  }
                  lo = std::min(lo, v);
                  hi = std::max(hi, v);
               }
            }
            if (lo > hi) lo = hi = 0;
            col.width = ldb_narrowest_width(lo, hi);
            if (col.type.type == LDB_T_CHAR4 && (col.width > 1 || lo < 0)) col.width = 4; // (a byte >= 0x80 or a second character: the four raw bytes stay)
         }
         col.value_bytes = rows * col.width;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) col.value_bytes));
         int64_t pos = 0;
         std::vector<int64_t> tmp;
         for (int64_t b = 0; b < n_batches; b++) {
            struct ArrowArray* a = batches[b]->children[c];
            const int64_t alen = batches[b]->length, aoff = batches[b]->offset + a->offset; // the parent's slice applies to every child
            if (alen == 0) continue;
            const uint8_t* src = (const uint8_t*) a->buffers[1] + aoff * src_w;
            if (src_w == col.width) {
               LDB_HIP(hipMemcpyAsync((uint8_t*) col.values + pos * col.width, src, (size_t) (alen * src_w), hipMemcpyHostToDevice, ctx->stream));
            } else { // narrowed column: keep the low `width` bytes (little-endian two's complement: the value fits)
               const size_t w = (size_t) col.width;
               tmp.resize((size_t) ((alen * (int64_t) w + 7) / 8));
               uint8_t* dstb = (uint8_t*) tmp.data();
               for (int64_t i = 0; i < alen; i++) memcpy(dstb + (size_t) i * w, src + i * src_w, w);
               LDB_HIP(hipMemcpyAsync((uint8_t*) col.values + pos * (int64_t) w, tmp.data(), (size_t) alen * w, hipMemcpyHostToDevice, ctx->stream));
               LDB_HIP(hipStreamSynchronize(ctx->stream));
            }
            pos += alen;
         }
         LDB_HIP(hipStreamSynchronize(ctx->stream));
      }
   }
   LDB_TRY(ldb_table_dict_encode_all(ctx, t.get())); // low-cardinality utf8 columns get an order-preserving dictionary
   *out = t.release();
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_table_alloc(ldb_ctx* ctx, const char* name, int32_t n_cols, const ldb_coltype* types, const char* const* col_names,
                                       int64_t n_rows, const int64_t* data_bytes, int32_t narrow, ldb_table** out) {
   if (!ctx || !out || n_cols < 0) LDB_FAIL(LDB_ERR_INVALID, "table_alloc: bad argument");
   if (n_rows >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "table_alloc: %ld rows exceed uint32 row ids", (long) n_rows);
   auto t = std::make_unique<ldb_table>();
   t->ctx = ctx;
   t->name = name ? name : "";
   t->n_rows = n_rows;
   t->cols.resize((size_t) n_cols);
   for (int32_t c = 0; c < n_cols; c++) {
      ldb_column& col = t->cols[(size_t) c];
      col.name = col_names && col_names[c] ? col_names[c] : "";
      col.type = types[c];
      col.width = ldb_width_of(col.type, narrow);
      if (col.type.type == LDB_T_UTF8) {
         col.value_bytes = data_bytes ? data_bytes[c] : 0;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) col.value_bytes));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.offsets, sizeof(int64_t) * ((size_t) n_rows + 1)));
      } else {
         col.value_bytes = n_rows * col.width;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) col.value_bytes));
      }
   }
   *out = t.release();
   return LDB_OK;
}
// The table's hash index over `cols` (primary key): built on first use with ldb_gpu_join_build over ALL rows, owned by the
// table, released with it.  Replaces the reference's persisted LingoDBHashIndex (LingoDBHashIndex.h:18-61: hash → row id,
// looked up by index nested-loop joins when the inner side is a bare base table whose primary key equals the join
// columns, translateINLJ RelAlgToSubOp.cpp:1129-1205, OptimizeImplementations.cpp:226-245): here the index IS a join
// table — for a dense primary key the rank-bitmap layout, range / 4 bytes — so probing it is the ordinary probe.
extern "C" int32_t ldb_gpu_table_index(ldb_ctx* ctx, ldb_table* t, const int32_t* cols, int32_t n_cols, ldb_hashtable** out) {
   if (!ctx || !t || !cols || !out || n_cols < 1 || n_cols > LDB_MAX_KEYS) LDB_FAIL(LDB_ERR_INVALID, "table_index: bad argument");
   for (auto& ix : t->indexes)
      if (ix.cols.size() == (size_t) n_cols && std::equal(ix.cols.begin(), ix.cols.end(), cols)) {
         *out = ix.ht;
         return LDB_OK;
      }
   ldb_table::Index ix;
   ix.cols.assign(cols, cols + n_cols);
   std::vector<ldb_colref> keys;
   for (int32_t k = 0; k < n_cols; k++) {
      if (cols[k] < 0 || (size_t) cols[k] >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "table_index: column %d out of range", cols[k]);
      keys.push_back({0, cols[k]});
   }
   LDB_TRY(ldb_gpu_rel_from_table(ctx, t, &ix.rel));
   const int32_t st = ldb_gpu_join_build(ctx, ix.rel, keys.data(), n_cols, 1, &ix.ht);
   if (st != LDB_OK) {
      ldb_gpu_rel_release(ctx, ix.rel);
      return st;
   }
   t->indexes.push_back(ix);
   *out = ix.ht;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_release(ldb_ctx* ctx, ldb_table* t) {
   if (!t) return LDB_OK;
   for (auto& ix : t->indexes) {
      ldb_gpu_hashtable_release(ctx, ix.ht);
      ldb_gpu_rel_release(ctx, ix.rel);
   }
   t->indexes.clear();
   for (auto& c : t->cols) {
      if (!c.owned) continue;
      ldb_dev_free(ctx, c.values);
      ldb_dev_free(ctx, c.offsets);
      ldb_dev_free(ctx, c.validity);
      ldb_dev_free(ctx, c.zone_min);
      ldb_dev_free(ctx, c.zone_max);
      ldb_column_dict_release(ctx, c);
   }
   delete t;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_rename_col(ldb_table* t, int32_t col, const char* name) {
   if (!t || !name || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "rename_col: bad argument");
   t->cols[(size_t) col].name = name;
   return LDB_OK;
}
extern "C" int64_t ldb_gpu_table_rows(const ldb_table* t) { return t ? t->n_rows : -1; }
extern "C" int32_t ldb_gpu_table_cols(const ldb_table* t) { return t ? (int32_t) t->cols.size() : -1; }
extern "C" int32_t ldb_gpu_table_coltype(const ldb_table* t, int32_t col, ldb_coltype* out) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "coltype: bad column %d", col);
   *out = t->cols[(size_t) col].type;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_col_index(const ldb_table* t, const char* name) {
   for (size_t i = 0; i < t->cols.size(); i++)
      if (t->cols[i].name == name) return (int32_t) i;
   return -1;
}
extern "C" const char* ldb_gpu_table_col_name(const ldb_table* t, int32_t col) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) return nullptr;
   return t->cols[(size_t) col].name.c_str();
}
extern "C" int32_t ldb_gpu_table_col_width(const ldb_table* t, int32_t col) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) return -1;
   return t->cols[(size_t) col].width;
}
extern "C" int32_t ldb_gpu_table_col_ptrs(const ldb_table* t, int32_t col, void** values, void** offsets, void** validity, int64_t* value_bytes) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "col_ptrs: bad column %d", col);
   const ldb_column& c = t->cols[(size_t) col];
   if (ldb_column_is_lazy(c)) LDB_TRY(ldb_column_strings(t->ctx, c, t->n_rows));
   c.has_range = false, c.sorted_state = -1, c.skewed = false; // the caller may write through the raw pointers
   if (values) *values = c.values;
   if (offsets) *offsets = c.offsets;
   if (validity) *validity = c.validity;
   if (value_bytes) *value_bytes = c.value_bytes;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_set_rows(ldb_table* t, int64_t n_rows) {
   if (!t || n_rows < 0) LDB_FAIL(LDB_ERR_INVALID, "set_rows: bad argument");
   for (auto& c : t->cols)
      if (c.type.type != LDB_T_UTF8 && n_rows * c.width > c.value_bytes) LDB_FAIL(LDB_ERR_INVALID, "set_rows: %ld rows exceed capacity of column %s", (long) n_rows, c.name.c_str());
   t->n_rows = n_rows;
   t->serial = ldb_next_serial();
   for (auto& c : t->cols) c.has_range = false, c.sorted_state = -1, c.skewed = false;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_read_fixed(ldb_ctx* ctx, const ldb_table* t, int32_t col, void* host_out, int64_t out_bytes) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "read_fixed: bad column %d", col);
   const ldb_column& c = t->cols[(size_t) col];
   if (c.type.type == LDB_T_UTF8) LDB_FAIL(LDB_ERR_INVALID, "read_fixed: column %s is utf8 (use export)", c.name.c_str());
   int64_t need = t->n_rows * c.width;
   if (out_bytes < need) LDB_FAIL(LDB_ERR_INVALID, "read_fixed: buffer %ld < %ld bytes", (long) out_bytes, (long) need);
   // (a scalar-subquery result read by the plan layer is a read-back like any count: recorded / replayed inside a trace)
   if (need) return LDB_READBACK(ctx, host_out, c.values, (size_t) need);
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_row_valid(ldb_ctx* ctx, const ldb_table* t, int32_t col, int64_t row, int32_t* valid) {
   if (!ctx || !t || !valid || col < 0 || (size_t) col >= t->cols.size() || row < 0 || row >= t->n_rows) LDB_FAIL(LDB_ERR_INVALID, "row_valid: bad argument");
   const ldb_column& c = t->cols[(size_t) col];
   *valid = 1;
   if (!c.validity) return LDB_OK;
   uint8_t byte = 0xFF;
   LDB_TRY(LDB_READBACK(ctx, &byte, c.validity + (row >> 3), 1));
   *valid = (byte >> (row & 7)) & 1;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_write_fixed(ldb_ctx* ctx, ldb_table* t, int32_t col, const void* host_in, int64_t in_bytes) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "write_fixed: bad column %d", col);
   ldb_column& c = t->cols[(size_t) col];
   if (c.type.type == LDB_T_UTF8) LDB_FAIL(LDB_ERR_INVALID, "write_fixed: column %s is utf8", c.name.c_str());
   if (in_bytes > c.value_bytes) LDB_FAIL(LDB_ERR_INVALID, "write_fixed: %ld bytes exceed column capacity %ld", (long) in_bytes, (long) c.value_bytes);
   if (in_bytes) LDB_HIP(hipMemcpyAsync(c.values, host_in, (size_t) in_bytes, hipMemcpyHostToDevice, ctx->stream));
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   c.has_range = false, c.sorted_state = -1, c.skewed = false;
   t->serial = ldb_next_serial();
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_memcpy_d2d(ldb_ctx* ctx, void* dst, const void* src, int64_t bytes) {
   if (!ctx || bytes < 0 || (bytes && (!dst || !src))) LDB_FAIL(LDB_ERR_INVALID, "memcpy_d2d: bad argument");
   if (bytes) LDB_HIP(hipMemcpyAsync(dst, src, (size_t) bytes, hipMemcpyDeviceToDevice, ctx->stream));
   return LDB_OK;
}

// ---------------------------------------------------------------- export (Arrow C Data Interface out)
struct ExportPriv {
   std::vector<void*> bufs;
   std::vector<struct ArrowArray*> child_arrays;
   std::vector<struct ArrowSchema*> child_schemas;
   std::vector<char*> strings;
   const void** buffers = nullptr;
};
static void release_array(struct ArrowArray* a) {
   if (!a || !a->release) return;
   ExportPriv* p = (ExportPriv*) a->private_data;
   for (int64_t i = 0; i < a->n_children; i++) {
      if (a->children[i]->release) a->children[i]->release(a->children[i]);
      free(a->children[i]);
   }
   free(a->children);
   for (void* b : p->bufs) free(b);
   free(p->buffers);
   delete p;
   a->release = nullptr;
}
static void release_schema(struct ArrowSchema* s) {
   if (!s || !s->release) return;
   ExportPriv* p = (ExportPriv*) s->private_data;
   for (int64_t i = 0; i < s->n_children; i++) {
      if (s->children[i]->release) s->children[i]->release(s->children[i]);
      free(s->children[i]);
   }
   free(s->children);
   for (char* c : p->strings) free(c);
   delete p;
   s->release = nullptr;
}
static char* dup_str(ExportPriv* p, const std::string& s) {
   char* c = strdup(s.c_str());
   p->strings.push_back(c);
   return c;
}

// all buffers of a small result → the pinned ring in ONE launch (a hipMemcpyAsync per buffer was 10 – 17 copy launches for a
// TPC-H result with string columns): workgroup b copies buffer b, 8 bytes per lane and step
struct DExportPack {
   int32_t n, pad;
   const uint8_t* src[32];
   uint8_t* dst[32];
   uint64_t bytes[32];
};
__global__ __launch_bounds__(256) void k_export_pack(DExportPack d) {
   const int b = blockIdx.x;
   if (b >= d.n) return;
   const uint64_t words = d.bytes[b] / 8;
   const uint64_t* s8 = (const uint64_t*) d.src[b];
   uint64_t* d8 = (uint64_t*) d.dst[b];
   for (uint64_t i = threadIdx.x; i < words; i += 256) d8[i] = s8[i];
   for (uint64_t i = words * 8 + threadIdx.x; i < d.bytes[b]; i += 256) d.dst[b][i] = d.src[b][i];
}
extern "C" int32_t ldb_gpu_export(ldb_ctx* ctx, const ldb_table* t, struct ArrowSchema* out_schema, struct ArrowArray* out_array) {
   if (!ctx || !t || !out_schema || !out_array) LDB_FAIL(LDB_ERR_INVALID, "export: NULL argument");
   const int64_t n = t->n_rows;
   const int64_t nc = (int64_t) t->cols.size();
   for (auto& col : t->cols)
      if (ldb_column_is_lazy(col)) LDB_TRY(ldb_column_strings(ctx, col, n)); // (dictionary-coded strings are written out only here)
   // Small results (the usual case: a query's final rows) come over in ONE batch: every device
   // buffer is copied asynchronously into the pinned ring, one synchronize, then plain memcpys —
   // instead of a blocking pageable hipMemcpy (~20 µs) per buffer.
   struct Staged {
      const uint8_t *validity = nullptr, *offsets = nullptr, *values = nullptr;
   };
   std::vector<Staged> staged((size_t) nc);
   {
      auto pad = [](size_t b) { return (b + 63) & ~(size_t) 63; };
      size_t total = 0;
      bool ok = ctx->h_ring != nullptr && n <= 65536;
      for (int64_t c = 0; c < nc && ok; c++) {
         const ldb_column& col = t->cols[(size_t) c];
         if (col.validity) total += pad((size_t) ((n + 7) / 8));
         if (col.type.type == LDB_T_UTF8) {
            if (col.value_bytes < 0) ok = false;
            total += pad(sizeof(int64_t) * ((size_t) n + 1)) + pad((size_t) col.value_bytes);
         } else {
            total += pad((size_t) n * (size_t) col.width);
         }
      }
      if (ok && total <= LDB_RING_BYTES / 4 && n > 0) {
         if (ctx->ring_pos + total > LDB_RING_BYTES) {
            LDB_HIP(hipStreamSynchronize(ctx->stream));
            ctx->ring_pos = 0;
         }
         DExportPack pack;
         memset(&pack, 0, sizeof(pack));
         auto flush = [&]() {
            if (pack.n) hipLaunchKernelGGL(k_export_pack, dim3((unsigned) pack.n), dim3(256), 0, ctx->stream, pack);
            pack.n = 0;
         };
         auto fetch = [&](const void* src, size_t bytes, const uint8_t** slot_out) -> int32_t {
            if (!bytes) return LDB_OK;
            uint8_t* slot = ctx->h_ring + ctx->ring_pos;
            ctx->ring_pos += pad(bytes);
            if (((uintptr_t) src & 7) == 0) { // (device buffers are 256-byte aligned; a view's base need not be)
               if (pack.n == 32) flush();
               pack.src[pack.n] = (const uint8_t*) src;
               pack.dst[pack.n] = slot;
               pack.bytes[pack.n] = bytes;
               pack.n++;
            } else {
               LDB_HIP(hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            }
            *slot_out = slot;
            return LDB_OK;
         };
         for (int64_t c = 0; c < nc; c++) {
            const ldb_column& col = t->cols[(size_t) c];
            Staged& st = staged[(size_t) c];
            if (col.validity) LDB_TRY(fetch(col.validity, (size_t) ((n + 7) / 8), &st.validity));
            if (col.type.type == LDB_T_UTF8) {
               LDB_TRY(fetch(col.offsets, sizeof(int64_t) * ((size_t) n + 1), &st.offsets));
               LDB_TRY(fetch(col.values, (size_t) col.value_bytes, &st.values));
            } else {
               LDB_TRY(fetch(col.values, (size_t) n * (size_t) col.width, &st.values));
            }
         }
         flush();
         LDB_HIP(hipGetLastError());
      }
   }
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   auto d2h = [&](void* dst, const void* src, size_t bytes, const uint8_t* from_ring) -> int32_t {
      if (!bytes) return LDB_OK;
      if (from_ring) {
         memcpy(dst, from_ring, bytes);
         return LDB_OK;
      }
      LDB_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
      return LDB_OK;
   };
   memset(out_schema, 0, sizeof(*out_schema));
   memset(out_array, 0, sizeof(*out_array));
   auto* sp = new ExportPriv();
   out_schema->format = dup_str(sp, "+s");
   out_schema->name = dup_str(sp, t->name);
   out_schema->n_children = nc;
   out_schema->children = (struct ArrowSchema**) calloc((size_t) (nc ? nc : 1), sizeof(void*));
   out_schema->release = release_schema;
   out_schema->private_data = sp;
   auto* ap = new ExportPriv();
   ap->buffers = (const void**) calloc(1, sizeof(void*));
   out_array->length = n;
   out_array->n_buffers = 1;
   out_array->buffers = ap->buffers;
   out_array->n_children = nc;
   out_array->children = (struct ArrowArray**) calloc((size_t) (nc ? nc : 1), sizeof(void*));
   out_array->release = release_array;
   out_array->private_data = ap;
   for (int64_t c = 0; c < nc; c++) {
      const ldb_column& col = t->cols[(size_t) c];
      auto* cs = (struct ArrowSchema*) calloc(1, sizeof(struct ArrowSchema));
      auto* ca = (struct ArrowArray*) calloc(1, sizeof(struct ArrowArray));
      out_schema->children[c] = cs;
      out_array->children[c] = ca;
      auto* csp = new ExportPriv();
      auto* cap = new ExportPriv();
      cs->release = release_schema;
      cs->private_data = csp;
      cs->children = nullptr;
      ca->release = release_array;
      ca->private_data = cap;
      cs->name = dup_str(csp, col.name);
      cs->flags = ARROW_FLAG_NULLABLE;
      ca->length = n;
      ca->null_count = col.validity ? col.null_count : 0;
      char fmt[64];
      bool utf8 = col.type.type == LDB_T_UTF8;
      bool large = false;
      switch (col.type.type) {
         case LDB_T_INT8: strcpy(fmt, "c"); break;
         case LDB_T_BOOL8: strcpy(fmt, "C"); break; // uint8 0/1
         case LDB_T_INT16: strcpy(fmt, "s"); break;
         case LDB_T_INT32: strcpy(fmt, "i"); break;
         case LDB_T_INT64: strcpy(fmt, "l"); break;
         case LDB_T_DATE32: strcpy(fmt, "tdD"); break;
         case LDB_T_FLOAT64: strcpy(fmt, "g"); break;
         case LDB_T_FLOAT32: strcpy(fmt, "f"); break;
         case LDB_T_CHAR4: strcpy(fmt, "w:4"); break;
         case LDB_T_DECIMAL128: snprintf(fmt, sizeof(fmt), "d:%d,%d", col.type.precision, col.type.scale); break;
         case LDB_T_UTF8:
            large = col.value_bytes > 0x7fffffffLL;
            strcpy(fmt, large ? "U" : "u");
            break;
         default: LDB_FAIL(LDB_ERR_UNSUPPORTED, "export: column type %d", col.type.type);
      }
      cs->format = dup_str(csp, fmt);
      ca->n_buffers = utf8 ? 3 : 2;
      cap->buffers = (const void**) calloc(3, sizeof(void*));
      ca->buffers = cap->buffers;
      if (col.validity) {
         size_t vb = (size_t) ((n + 7) / 8);
         void* hv = malloc(vb ? vb : 1);
         cap->bufs.push_back(hv);
         LDB_TRY(d2h(hv, col.validity, vb, staged[(size_t) c].validity));
         cap->buffers[0] = hv;
      }
      if (utf8) {
         std::vector<int64_t> offs((size_t) n + 1, 0);
         LDB_TRY(d2h(offs.data(), col.offsets, sizeof(int64_t) * ((size_t) n + 1), staged[(size_t) c].offsets));
         int64_t bytes = offs[(size_t) n];
         void* data = malloc((size_t) (bytes ? bytes : 1));
         cap->bufs.push_back(data);
         LDB_TRY(d2h(data, col.values, (size_t) bytes, bytes <= col.value_bytes ? staged[(size_t) c].values : nullptr));
         if (large) {
            void* ho = malloc(sizeof(int64_t) * ((size_t) n + 1));
            memcpy(ho, offs.data(), sizeof(int64_t) * ((size_t) n + 1));
            cap->bufs.push_back(ho);
            cap->buffers[1] = ho;
         } else {
            int32_t* ho = (int32_t*) malloc(sizeof(int32_t) * ((size_t) n + 1));
            for (int64_t i = 0; i <= n; i++) ho[i] = (int32_t) offs[(size_t) i];
            cap->bufs.push_back(ho);
            cap->buffers[1] = ho;
         }
         cap->buffers[2] = data;
      } else {
         int out_w = ldb_width_of(col.type, 0);
         size_t bytes = (size_t) (n * out_w);
         uint8_t* hv = (uint8_t*) malloc(bytes ? bytes : 1);
         cap->bufs.push_back(hv);
         if (out_w == col.width) {
            LDB_TRY(d2h(hv, col.values, bytes, staged[(size_t) c].values));
         } else { // narrowed column → sign-extend back to its Arrow width (decimal128: reference LowerToStd.cpp:211-298; char(1): its byte + three zero bytes)
            const size_t w = (size_t) col.width;
            std::vector<uint8_t> tmp((size_t) n * w + 8);
            LDB_TRY(d2h(tmp.data(), col.values, (size_t) n * w, staged[(size_t) c].values));
            for (int64_t i = 0; i < n; i++) {
               int64_t lo = 0;
               switch (w) {
                  case 1: lo = (int8_t) tmp[(size_t) i]; break;
                  case 2: { int16_t v; memcpy(&v, tmp.data() + (size_t) i * 2, 2); lo = v; break; }
                  case 4: { int32_t v; memcpy(&v, tmp.data() + (size_t) i * 4, 4); lo = v; break; }
                  default: memcpy(&lo, tmp.data() + (size_t) i * 8, 8); break;
               }
               const int64_t hi = lo >> 63;
               memcpy(hv + i * out_w, &lo, (size_t) std::min(out_w, 8));
               if (out_w == 16) memcpy(hv + i * 16 + 8, &hi, 8);
            }
         }
         cap->buffers[1] = hv;
      }
   }
   return LDB_OK;
}

// ---------------------------------------------------------------- relations
ldb_rel* ldb_rel_new(ldb_ctx* ctx) {
   auto* r = new ldb_rel();
   r->ctx = ctx;
   return r;
}
extern "C" int32_t ldb_gpu_rel_from_table(ldb_ctx* ctx, const ldb_table* t, ldb_rel** out) {
   if (!ctx || !t || !out) LDB_FAIL(LDB_ERR_INVALID, "rel_from_table: NULL argument");
   ldb_rel* r = ldb_rel_new(ctx);
   r->n_rows = t->n_rows;
   r->sides.push_back({t, nullptr, false});
   *out = r;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_rel_release(ldb_ctx* ctx, ldb_rel* r) {
   if (!r) return LDB_OK;
   for (auto& s : r->sides)
      if (s.owned) ldb_dev_free(ctx, s.rowids);
   delete r;
   return LDB_OK;
}
extern "C" int64_t ldb_gpu_rel_rows(ldb_ctx* ctx, ldb_rel* r) {
   if (!r) return -1;
   if (!r->pending.empty() && ldb_rel_force(ctx ? ctx : r->ctx, r) != LDB_OK) return -1; // a lazy filter is evaluated now
   return r->n_rows;
}
extern "C" int32_t ldb_gpu_rel_sides(const ldb_rel* r) { return r ? (int32_t) r->sides.size() : -1; }

__global__ void k_iota_u32(uint32_t* out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = (uint32_t) i;
}
extern "C" int32_t ldb_gpu_rel_read_rowids(ldb_ctx* ctx, ldb_rel* r, int32_t side, uint32_t* host_out, int64_t cap) {
   if (!r || side < 0 || (size_t) side >= r->sides.size()) LDB_FAIL(LDB_ERR_INVALID, "read_rowids: bad side %d", side);
   LDB_TRY(ldb_rel_force(ctx, r));
   if (cap < r->n_rows) LDB_FAIL(LDB_ERR_INVALID, "read_rowids: buffer too small");
   if (r->sides[(size_t) side].rowids) {
      if (r->n_rows) LDB_HIP(hipMemcpyAsync(host_out, r->sides[(size_t) side].rowids, (size_t) r->n_rows * 4, hipMemcpyDeviceToHost, ctx->stream));
      LDB_HIP(hipStreamSynchronize(ctx->stream));
   } else {
      for (int64_t i = 0; i < r->n_rows; i++) host_out[i] = (uint32_t) i;
   }
   return LDB_OK;
}

int32_t ldb_make_dcol(const ldb_rel* r, ldb_colref ref, DCol* out) {
   if (ref.side < 0 || (size_t) ref.side >= r->sides.size()) LDB_FAIL(LDB_ERR_INVALID, "column ref: side %d out of range", ref.side);
   const ldb_rel_side& s = r->sides[(size_t) ref.side];
   if (ref.col < 0 || (size_t) ref.col >= s.table->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "column ref: col %d out of range on side %d", ref.col, ref.side);
   const ldb_column& c = s.table->cols[(size_t) ref.col];
   if (ldb_column_is_lazy(c)) LDB_TRY(ldb_column_strings(r->ctx, c, s.table->n_rows)); // a byte-wise consumer: the strings are needed now
   out->values = (uint64_t) c.values;
   out->offsets = (uint64_t) c.offsets;
   out->validity = (uint64_t) c.validity;
   out->rowids = (uint64_t) s.rowids;
   out->type = c.type.type;
   out->width = c.width;
   out->precision = c.type.precision;
   out->scale = c.type.scale;
   return LDB_OK;
}

int32_t ldb_make_dpred(const ldb_rel* r, const ldb_filter_desc* p, DPred* out) {
   memset(out, 0, sizeof(*out));
   LDB_TRY(ldb_make_dcol(r, p->col, &out->col));
   out->op = p->op;
   out->rhs_kind = p->rhs_kind;
   out->lo = p->value_lo;
   out->hi = p->value_hi;
   out->f = p->value_f64;
   if (p->op < LDB_F_EQ || p->op > LDB_F_NOT_LIKE) LDB_FAIL(LDB_ERR_INVALID, "filter: bad op %d", p->op);
   if ((p->op == LDB_F_LIKE || p->op == LDB_F_NOT_LIKE) && (out->col.type != LDB_T_UTF8 || p->rhs_kind != LDB_RHS_STRING))
      LDB_FAIL(LDB_ERR_INVALID, "filter: LIKE needs a utf8 column and a string pattern");
   if (p->op == LDB_F_NOTNULL) return LDB_OK;
   bool is_str = out->col.type == LDB_T_UTF8;
   if (p->rhs_kind == LDB_RHS_COLUMN) {
      if (p->op == LDB_F_IN) LDB_FAIL(LDB_ERR_INVALID, "filter: IN needs constants");
      LDB_TRY(ldb_make_dcol(r, p->rhs_col, &out->rhs));
      if ((out->rhs.type == LDB_T_UTF8) != is_str) LDB_FAIL(LDB_ERR_INVALID, "filter: string compared with non-string column");
      return LDB_OK;
   }
   if (is_str != (p->rhs_kind == LDB_RHS_STRING)) LDB_FAIL(LDB_ERR_INVALID, "filter: constant kind %d does not match column type %d", p->rhs_kind, out->col.type);
   if (is_str) { // a dictionary-encoded column: the predicate becomes a test on the 4-byte codes
      bool done = false;
      LDB_TRY(ldb_dict_rewrite_pred(r, p, out, &done));
      if (done) return LDB_OK;
   }
   if (p->op == LDB_F_IN) {
      if (p->n_in < 0 || p->n_in > LDB_MAX_IN) LDB_FAIL(LDB_ERR_UNSUPPORTED, "filter: IN list of %d values (max %d)", p->n_in, LDB_MAX_IN);
      out->n_in = p->n_in;
      if (is_str) {
         int32_t pos = 0;
         for (int k = 0; k < p->n_in; k++) {
            if (pos + p->in_str_lens[k] > (int32_t) sizeof(out->in_blob)) LDB_FAIL(LDB_ERR_UNSUPPORTED, "filter: IN string constants exceed %zu bytes", sizeof(out->in_blob));
            out->in_off[k] = pos;
            memcpy(out->in_blob + pos, p->in_strs[k], (size_t) p->in_str_lens[k]);
            pos += p->in_str_lens[k];
         }
         out->in_off[p->n_in] = pos;
      } else {
         for (int k = 0; k < p->n_in; k++) {
            out->in_lo[k] = (uint64_t) p->in_values[2 * k];
            out->in_hi[k] = p->in_values[2 * k + 1];
         }
      }
      return LDB_OK;
   }
   // zone map: column-vs-constant comparison over a dense, NOT NULL integer-like column of a large base table
   if (!is_str && p->rhs_kind == LDB_RHS_INT && p->op >= LDB_F_EQ && p->op <= LDB_F_GTE && !out->col.rowids && !out->col.validity && out->hi == ((int64_t) out->lo >> 63) && r->ctx &&
       ldb_option("zone_maps", 1) != 0) {
      const ldb_table* zt = r->sides[(size_t) p->col.side].table;
      if (zt->n_rows >= ldb_option("zone_min_rows", 1 << 20)) LDB_TRY(ldb_column_zones(r->ctx, zt, p->col.col, &out->zmin, &out->zmax));
   }
   if (is_str) {
      if (p->str_len < 0 || p->str_len > LDB_STR_INLINE) LDB_FAIL(LDB_ERR_UNSUPPORTED, "filter: string constant of %d bytes (max %d)", p->str_len, LDB_STR_INLINE);
      out->str_len = p->str_len;
      memcpy(out->str, p->str, (size_t) p->str_len);
      if (p->op == LDB_F_LIKE || p->op == LDB_F_NOT_LIKE) ldb_like_plan(out);
   }
   return LDB_OK;
}

// "Simple" LIKE patterns — ASCII literals separated by '%', no '_' and no escape: %A%B%, A%, %A,
// A%B … — are matched by position instead of by row (ldb_device.h d_like_simple_wave).  For such a
// pattern byte-wise substring search IS the reference's character-wise iterativeLike
// (StringRuntime.cpp:28-93): an ASCII pattern byte never equals a UTF-8 lead or continuation byte.
// Plan: n_in = number of literal segments (0 = not simple), in_off[2j] / in_off[2j+1] = start and
// length of segment j inside str, lo bit 0 / bit 1 = the pattern is anchored at the start / end.
void ldb_like_plan(DPred* d);
// the planner's verdict on a pattern, for host-side tests and for an emitter that wants to know which
// matcher a LIKE will get: *n_segments = 0 → general matcher; else seg[2j] / seg[2j+1] = start / length of
// literal j inside the pattern and *anchors bit 0 / bit 1 = anchored at the start / end.  No device needed.
extern "C" int32_t ldb_gpu_like_plan(const char* pattern, int32_t len, int32_t* n_segments, int32_t* seg, int32_t* anchors) {
   if (!pattern || !n_segments || len < 0 || len > LDB_STR_INLINE) LDB_FAIL(LDB_ERR_INVALID, "like_plan: bad argument (patterns are at most %d bytes)", LDB_STR_INLINE);
   DPred d;
   memset(&d, 0, sizeof(d));
   d.str_len = len;
   memcpy(d.str, pattern, (size_t) len);
   ldb_like_plan(&d);
   *n_segments = d.n_in;
   if (seg)
      for (int j = 0; j < 2 * d.n_in; j++) seg[j] = d.in_off[j];
   if (anchors) *anchors = d.n_in ? (int32_t) (d.lo & 3) : 0;
   return LDB_OK;
}
void ldb_like_plan(DPred* d) {
   d->n_in = 0;
   const int n = d->str_len;
   int nseg = 0, seg_start = -1, off[2 * LDB_LIKE_MAX_SEG];
   for (int k = 0; k <= n; k++) {
      const unsigned char c = k < n ? (unsigned char) d->str[k] : (unsigned char) '%';
      if (k < n && (c >= 0x80 || c == '_' || c == '\\')) return;
      if (c == '%') {
         if (seg_start >= 0) {
            if (nseg == LDB_LIKE_MAX_SEG || k - seg_start > 16) return;
            off[2 * nseg] = seg_start;
            off[2 * nseg + 1] = k - seg_start;
            nseg++;
            seg_start = -1;
         }
      } else if (seg_start < 0) {
         seg_start = k;
      }
   }
   if (nseg == 0) return; // "", "%", "%%": the general matcher decides at once
   for (int j = 0; j < 2 * nseg; j++) d->in_off[j] = off[j];
   d->n_in = nseg;
   d->lo = (d->str[0] != '%' ? 1u : 0u) | (d->str[n - 1] != '%' ? 2u : 0u);
}

// marks conjuncts whose lhs column is the previous conjunct's (range filters): the batched
// evaluator loads the column once for both (ldb_device.h d_eval_conj_batch)
// A conjunction may be evaluated in any order (no side effects; NULL operands just fail).  Cheap
// conjuncts go first so that the expensive ones (string compares: offsets + bytes per row) only
// run for the surviving lanes: Q12's lineitem filter lists l_shipmode IN ('MAIL','SHIP') first and
// spent 6.3 ms reading 600 M strings before the date conjuncts had rejected 97 % of the rows.
static int pred_cost(const DPred& p) {
   const bool str = p.col.type == LDB_T_UTF8;
   const bool wide = (p.col.type == LDB_T_DECIMAL128 && p.col.precision >= 19) || p.col.type == LDB_T_FLOAT64 || p.col.type == LDB_T_FLOAT32;
   if (p.op == LDB_F_NOTNULL) return 0;
   if (str) return p.op == LDB_F_LIKE || p.op == LDB_F_NOT_LIKE ? 7 : (p.op == LDB_F_IN || p.rhs_kind == LDB_RHS_COLUMN ? 6 : 5);
   if (wide) return 3;
   if (p.rhs_kind == LDB_RHS_COLUMN || p.op == LDB_F_IN) return 2;
   return 1;
}
// ---------------------------------------------------------------- column statistics
__global__ void k_column_range(DCol col, uint64_t n, long long* __restrict__ out) {
   long long lo = 0x7FFFFFFFFFFFFFFFll, hi = -0x7FFFFFFFFFFFFFFFll - 1;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      long long k = d_load_i64(col, (uint32_t) i);
      lo = k < lo ? k : lo;
      hi = k > hi ? k : hi;
   }
   for (int off = 32; off > 0; off >>= 1) {
      long long l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
   }
   if ((threadIdx.x & 63) == 0) {
      atomicMin(&out[0], lo);
      atomicMax(&out[1], hi);
   }
}
int32_t ldb_column_range(ldb_ctx* ctx, const ldb_table* t, int32_t col, int64_t* lo, int64_t* hi) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "column_range: bad column %d", col);
   const ldb_column& c = t->cols[(size_t) col];
   const int ty = c.type.type;
   const bool ok = ty == LDB_T_INT8 || ty == LDB_T_INT16 || ty == LDB_T_INT32 || ty == LDB_T_INT64 || ty == LDB_T_DATE32 ||
                   (ty == LDB_T_DECIMAL128 && c.type.precision < 19);
   if (!ok) return LDB_ERR_UNSUPPORTED;
   if (!c.has_range) {
      long long* d_out = (long long*) (ctx->d_scratch + 40);
      const long long init[2] = {INT64_MAX, INT64_MIN};
      long long got[2] = {0, -1};
      if (t->n_rows > 0) {
         DCol dc;
         memset(&dc, 0, sizeof(dc));
         dc.values = (uint64_t) c.values;
         dc.type = ty;
         dc.width = c.width;
         dc.precision = c.type.precision;
         dc.scale = c.type.scale;
         LDB_TRY(ldb_h2d_small(ctx, d_out, init, 16)); // (pinned staging: a pageable source makes the copy wait for the stream)
         hipLaunchKernelGGL(k_column_range, dim3(ldb_grid_for(ctx, t->n_rows, 256, 8)), dim3(256), 0, ctx->stream, dc, (uint64_t) t->n_rows, d_out);
         LDB_TRY(LDB_READBACK(ctx, got, d_out, 16));
      }
      c.vmin = got[0];
      c.vmax = got[1];
      c.has_range = true;
   }
   *lo = c.vmin;
   *hi = c.vmax;
   return LDB_OK;
}

// one workgroup per zone of LDB_ZONE_ROWS rows
__global__ __launch_bounds__(256) void k_column_zones(DCol col, uint64_t n, int64_t* __restrict__ zmin, int64_t* __restrict__ zmax) {
   __shared__ int64_t s_lo[4], s_hi[4];
   const uint64_t b = (uint64_t) blockIdx.x << LDB_ZONE_SHIFT, e = b + LDB_ZONE_ROWS < n ? b + LDB_ZONE_ROWS : n;
   int64_t lo = INT64_MAX, hi = INT64_MIN;
   for (uint64_t i = b + threadIdx.x; i < e; i += 256) {
      const int64_t v = d_load_i64(col, (uint32_t) i);
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
   }
   for (int o = 32; o > 0; o >>= 1) {
      const int64_t l2 = __shfl_xor((long long) lo, o), h2 = __shfl_xor((long long) hi, o);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
   }
   if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6] = lo;
      s_hi[threadIdx.x >> 6] = hi;
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      for (int k = 1; k < 4; k++) {
         lo = s_lo[k] < lo ? s_lo[k] : lo;
         hi = s_hi[k] > hi ? s_hi[k] : hi;
      }
      zmin[blockIdx.x] = lo;
      zmax[blockIdx.x] = hi;
   }
}
int32_t ldb_column_zones(ldb_ctx* ctx, const ldb_table* t, int32_t col, uint64_t* zmin, uint64_t* zmax) {
   *zmin = *zmax = 0;
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "column_zones: bad column %d", col);
   const ldb_column& c = t->cols[(size_t) col];
   const int ty = c.type.type;
   const bool ok = ty == LDB_T_INT8 || ty == LDB_T_INT16 || ty == LDB_T_INT32 || ty == LDB_T_INT64 || ty == LDB_T_DATE32 || (ty == LDB_T_DECIMAL128 && c.type.precision < 19 && c.width == 8);
   if (!ok || !c.owned || c.validity || t->n_rows < 2 * (int64_t) LDB_ZONE_ROWS) return LDB_OK; // (views share their owner's values: no statistics of their own)
   if (c.zone_state < 0) {
      const int64_t nz = (t->n_rows + LDB_ZONE_ROWS - 1) >> LDB_ZONE_SHIFT;
      int64_t *zl, *zh;
      LdbBufs tmp(ctx); // (freed on the error returns below)
      LDB_TRY(tmp.alloc(&zl, 8 * (size_t) nz));
      LDB_TRY(tmp.alloc(&zh, 8 * (size_t) nz));
      DCol dc;
      memset(&dc, 0, sizeof(dc));
      dc.values = (uint64_t) c.values;
      dc.type = ty;
      dc.width = c.width;
      dc.precision = c.type.precision;
      dc.scale = c.type.scale;
      hipLaunchKernelGGL(k_column_zones, dim3((unsigned) nz), dim3(256), 0, ctx->stream, dc, (uint64_t) t->n_rows, zl, zh);
      std::vector<int64_t> hl((size_t) nz), hh((size_t) nz);
      LDB_HIP(hipMemcpyAsync(hl.data(), zl, 8 * (size_t) nz, hipMemcpyDeviceToHost, ctx->stream));
      LDB_HIP(hipMemcpyAsync(hh.data(), zh, 8 * (size_t) nz, hipMemcpyDeviceToHost, ctx->stream));
      LDB_HIP(hipStreamSynchronize(ctx->stream));
      // selective? the zones together must cover well under the whole value range (sorted / clustered columns do; a
      // uniformly scattered column has every zone span the full range and a zone test could never exclude anything)
      int64_t gl = INT64_MAX, gh = INT64_MIN;
      long double span = 0;
      for (int64_t z = 0; z < nz; z++) {
         gl = std::min(gl, hl[(size_t) z]);
         gh = std::max(gh, hh[(size_t) z]);
         span += (long double) hh[(size_t) z] - (long double) hl[(size_t) z];
      }
      const long double full = ((long double) gh - (long double) gl) * (long double) nz;
      const bool useful = full > 0 && span < 0.5L * full;
      if (useful) {
         c.zone_min = zl;
         c.zone_max = zh;
         tmp.keep(zl);
         tmp.keep(zh);
      }
      c.zone_state = useful ? 1 : 0;
   }
   if (c.zone_state == 1) {
      *zmin = (uint64_t) c.zone_min;
      *zmax = (uint64_t) c.zone_max;
   }
   return LDB_OK;
}

extern "C" int64_t ldb_gpu_table_zones(ldb_ctx* ctx, const ldb_table* t, int32_t col) {
   uint64_t a = 0, b = 0;
   if (!ctx || ldb_column_zones(ctx, t, col, &a, &b) != LDB_OK) return -1;
   return a ? (t->n_rows + LDB_ZONE_ROWS - 1) >> LDB_ZONE_SHIFT : 0;
}
__global__ void k_column_sorted(DCol col, uint64_t n, unsigned int* __restrict__ unsorted) {
   bool bad = false;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x + 1; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      bad = bad || d_load_i64(col, (uint32_t) i) < d_load_i64(col, (uint32_t) (i - 1));
   if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(unsorted, 1u);
}
int32_t ldb_column_sorted(ldb_ctx* ctx, const ldb_table* t, int32_t col, bool* sorted) {
   if (!t || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "column_sorted: bad column %d", col);
   const ldb_column& c = t->cols[(size_t) col];
   const int ty = c.type.type;
   const bool ok = ty == LDB_T_INT8 || ty == LDB_T_INT16 || ty == LDB_T_INT32 || ty == LDB_T_INT64 || ty == LDB_T_DATE32 ||
                   (ty == LDB_T_DECIMAL128 && c.type.precision < 19);
   if (c.sorted_state < 0) {
      if (!ok || c.validity) {
         c.sorted_state = 0;
      } else if (t->n_rows < 2) {
         c.sorted_state = 1;
      } else {
         DCol dc;
         memset(&dc, 0, sizeof(dc));
         dc.values = (uint64_t) c.values;
         dc.type = ty;
         dc.width = c.width;
         dc.precision = c.type.precision;
         dc.scale = c.type.scale;
         unsigned int* d_flag = (unsigned int*) (ctx->d_scratch + 44);
         LDB_HIP(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
         hipLaunchKernelGGL(k_column_sorted, dim3(ldb_grid_for(ctx, t->n_rows, 256, 8)), dim3(256), 0, ctx->stream, dc, (uint64_t) t->n_rows, d_flag);
         uint64_t f = 0;
         LDB_TRY(ldb_read_u64(ctx, d_flag, &f));
         c.sorted_state = (f & 1) ? 0 : 1;
      }
   }
   *sorted = c.sorted_state == 1;
   return LDB_OK;
}

void ldb_order_preds(DPred* preds, int32_t n) {
   std::stable_sort(preds, preds + n, [](const DPred& a, const DPred& b) { return pred_cost(a) < pred_cost(b); });
   ldb_mark_same_col(preds, n);
}

void ldb_mark_same_col(DPred* preds, int32_t n) {
   for (int32_t p = 1; p < n; p++) {
      const DCol &a = preds[p - 1].col, &b = preds[p].col;
      preds[p].same_col = a.values == b.values && a.offsets == b.offsets && a.validity == b.validity && a.rowids == b.rowids && a.type == b.type && a.width == b.width &&
                                a.precision == b.precision
                             ? 1
                             : 0;
   }
}

// ---------------------------------------------------------------- row-id composition
struct DComposeMulti {
   uint64_t n;
   const uint32_t* sel[2];
   const uint32_t* ids[12];
   uint32_t* out[12];
   int32_t n_out;
   uint8_t which[12];
};
__global__ void k_compose_multi(DComposeMulti d) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < d.n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t s0 = d.sel[0][i];
      const uint32_t s1 = d.sel[1] ? d.sel[1][i] : 0u;
#pragma unroll
      for (int j = 0; j < 12; j++) {
         if (j >= d.n_out) break;
         const uint32_t s = d.which[j] ? s1 : s0;
         d.out[j][i] = s == LDB_NULL_ROW ? LDB_NULL_ROW : (d.ids[j] ? d.ids[j][s] : s);
      }
   }
}
int32_t ldb_compose_rowids(ldb_ctx* ctx, const uint32_t* sel0, const uint32_t* sel1, const LdbComposeJob* jobs, int n_jobs, uint64_t n) {
   if (!n || !n_jobs) return LDB_OK;
   LdbProf prof_(ctx, "k_compose_rowids");
   for (int at = 0; at < n_jobs; at += 12) {
      DComposeMulti d;
      memset(&d, 0, sizeof(d));
      d.n = n;
      d.sel[0] = sel0;
      d.sel[1] = sel1;
      d.n_out = std::min(12, n_jobs - at);
      for (int j = 0; j < d.n_out; j++) {
         d.ids[j] = jobs[at + j].ids;
         d.out[j] = jobs[at + j].out;
         d.which[j] = (uint8_t) jobs[at + j].which;
      }
      hipLaunchKernelGGL(k_compose_multi, dim3(ldb_grid_for(ctx, (int64_t) n, 256, 8)), dim3(256), 0, ctx->stream, d);
   }
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}

// ---------------------------------------------------------------- gather / materialize
template <typename T>
__global__ void k_gather_fixed(const T* __restrict__ src, const uint32_t* __restrict__ rowids, T* __restrict__ dst, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint32_t r = rowids ? rowids[i] : (uint32_t) i;
      T v{};
      if (r != LDB_NULL_ROW) v = src[r];
      dst[i] = v;
   }
}
// validity bitmap of a gathered column + its NULL count in one pass: one row per lane, the wave's
// ballot is the 64-bit bitmap word (the bitmap buffer is 8-byte granular: ldb_dev_alloc pads)
__global__ void k_gather_validity(const uint8_t* __restrict__ validity, const uint32_t* __restrict__ rowids, uint64_t* __restrict__ bitmap_words, uint64_t n,
                                  unsigned long long* __restrict__ null_count) {
   const uint64_t n_pad = (n + 63) & ~(uint64_t) 63;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n_pad; i += (uint64_t) gridDim.x * blockDim.x) {
      bool ok = false;
      if (i < n) {
         const uint32_t r = rowids ? rowids[i] : (uint32_t) i;
         ok = r != LDB_NULL_ROW && (!validity || ((validity[r >> 3] >> (r & 7)) & 1));
      }
      const uint64_t m = __ballot(ok);
      if ((threadIdx.x & 63) == 0) {
         bitmap_words[i >> 6] = m;
         const uint64_t rows_here = n - i >= 64 ? ~(uint64_t) 0 : (((uint64_t) 1 << (n - i)) - 1);
         const int nulls = __popcll(~m & rows_here);
         if (nulls) atomicAdd(null_count, (unsigned long long) nulls);
      }
   }
}
__global__ void k_str_lens(const int64_t* __restrict__ offsets, const uint32_t* __restrict__ rowids, int64_t* __restrict__ lens, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint32_t r = rowids ? rowids[i] : (uint32_t) i;
      lens[i] = r == LDB_NULL_ROW ? 0 : offsets[r + 1] - offsets[r];
   }
}
__global__ void k_str_copy(const uint8_t* __restrict__ src, const int64_t* __restrict__ src_off, const uint32_t* __restrict__ rowids,
                           int64_t* __restrict__ dst_off, uint8_t* __restrict__ dst, uint64_t n, int64_t total) {
   if (blockIdx.x == 0 && threadIdx.x == 0) dst_off[n] = total; // the closing offset of the Arrow layout
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint32_t r = rowids ? rowids[i] : (uint32_t) i;
      if (r == LDB_NULL_ROW) continue;
      int64_t b = src_off[r], len = src_off[r + 1] - b, d = dst_off[i];
      if (d + len > total) len = total > d ? total - d : 0; // (total = the host's byte count; see ldb_readback)
      for (int64_t k = 0; k < len; k++) dst[d + k] = src[b + k];
   }
}

struct u128x {
   uint64_t a, b;
};

// Gather columns of `r` into new owned columns.  All launches of all columns are queued first; the
// NULL counts of the nullable ones and the byte totals of the string ones come back in ONE read
// (one stream synchronisation per call, none when no column is nullable or a string).  A column
// can be NULL in the result only if the source column has a validity bitmap or its side carries
// outer-join padding (ldb_rel_side::may_null).
int32_t ldb_gather_columns(ldb_ctx* ctx, const ldb_rel* r, const ldb_colref* refs, int32_t n_cols, ldb_column* outs) {
   const uint64_t n = (uint64_t) r->n_rows;
   const int grid = ldb_grid_for(ctx, (int64_t) n, 256, 8);
   struct Pending {
      int32_t col;
      int slot_nulls = -1, slot_bytes = -1;
      const uint32_t* rowids = nullptr;
      int64_t* lens = nullptr;
   };
   std::vector<Pending> pend((size_t) n_cols);
   int n_slots = 0;
   const bool lazy_on = ldb_option("lazy_strings", 1) != 0;
   const uint64_t lazy_min = (uint64_t) ldb_option("lazy_strings_min_rows", 4096);
   std::vector<ldb_column*> lazy_small; // lazy outputs too short to be worth keeping lazy: written out before returning
   for (int32_t c = 0; c < n_cols; c++) {
      const ldb_rel_side& side = r->sides[(size_t) refs[c].side];
      const ldb_column& src = side.table->cols[(size_t) refs[c].col];
      pend[(size_t) c].col = c;
      if (src.validity || (side.rowids && side.may_null)) pend[(size_t) c].slot_nulls = n_slots++;
      const bool will_be_lazy = src.type.type == LDB_T_UTF8 && src.dict_codes && src.dict && (ldb_column_is_lazy(src) || (n >= 64 && n >= lazy_min && lazy_on));
      if (src.type.type == LDB_T_UTF8 && !will_be_lazy) pend[(size_t) c].slot_bytes = n_slots++;
   }
   unsigned long long* d_words = nullptr;
   if (n_slots) LDB_TRY(ldb_counters(ctx, n_slots, (uint64_t**) &d_words)); // zeroed arena words
   LdbProf prof_(ctx, "k_gather"); // (all gather launches of the call; the string copies after the read-back are not inside)
   for (int32_t c = 0; c < n_cols; c++) {
      Pending& p = pend[(size_t) c];
      // (only the side's row ids are needed here; ldb_make_dcol would write a lazy source column's strings out — the very
      // thing a code-only gather avoids — and change its laziness between the slot assignment above and the branch below)
      p.rowids = r->sides[(size_t) refs[c].side].rowids;
      const ldb_column& src = r->sides[(size_t) refs[c].side].table->cols[(size_t) refs[c].col];
      ldb_column* out = &outs[c];
      out->name = src.name;
      out->type = src.type;
      out->width = src.width;
      out->owned = true;
      if (p.slot_nulls >= 0) {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &out->validity, (size_t) ((n + 7) / 8 + 8)));
         if (n) hipLaunchKernelGGL(k_gather_validity, dim3(grid), dim3(256), 0, ctx->stream, src.validity, p.rowids, (uint64_t*) out->validity, n, d_words + p.slot_nulls);
      }
      const bool src_lazy = ldb_column_is_lazy(src);
      if (src.type.type == LDB_T_UTF8 && src.dict_codes && src.dict && (n >= 64 || src_lazy)) {
         // the gathered column inherits the source's dictionary: codes gathered alongside, the dictionary table shared
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &out->dict_codes, 4 * (size_t) (n ? n : 1)));
         if (n) hipLaunchKernelGGL(k_gather_fixed<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*) src.dict_codes, p.rowids, out->dict_codes, n);
         out->dict = src.dict;
         out->dict->dict_refs++;
         out->dict_size = src.dict_size;
         out->dict_pred_cache = new std::unordered_map<std::string, std::string>();
      }
      if (src.type.type == LDB_T_UTF8 && out->dict_codes && (src_lazy || (n >= lazy_min && lazy_on))) {
         // LAZY: codes + dictionary only; whoever needs the bytes writes them out of the dictionary (ldb_column_strings)
         out->value_bytes = -1;
         p.slot_bytes = -1;
         lazy_small.push_back(src_lazy && n < 64 ? out : nullptr);
      } else if (src.type.type == LDB_T_UTF8) {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &p.lens, sizeof(int64_t) * (size_t) (n + 1)));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &out->offsets, sizeof(int64_t) * (size_t) (n + 1)));
         hipLaunchKernelGGL(k_str_lens, dim3(grid), dim3(256), 0, ctx->stream, src.offsets, p.rowids, p.lens, n);
         LDB_TRY(ldb_exclusive_scan_i64(ctx, p.lens, out->offsets, (int64_t) n, (int64_t*) (d_words + p.slot_bytes)));
      } else {
         out->value_bytes = (int64_t) n * src.width;
         // the gathered values are a subset of the source's: its cached [min, max] stays a valid (superset) statistic of the new column —
         // an intermediate's key range is then known without a pass over it in every execution (Q18: 150 M group keys, k_column_range
         // 0.35 ms inside the timed run; Q14, Q7).  Not where padding rows (outer joins) or NULL slots carry other bytes.
         if (src.has_range && p.slot_nulls < 0) {
            out->has_range = true;
            out->vmin = src.vmin;
            out->vmax = src.vmax;
         }
         LDB_TRY(ldb_dev_alloc(ctx, &out->values, (size_t) out->value_bytes));
         switch (src.width) {
            case 1: hipLaunchKernelGGL(k_gather_fixed<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) src.values, p.rowids, (uint8_t*) out->values, n); break;
            case 2: hipLaunchKernelGGL(k_gather_fixed<uint16_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint16_t*) src.values, p.rowids, (uint16_t*) out->values, n); break;
            case 4: hipLaunchKernelGGL(k_gather_fixed<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*) src.values, p.rowids, (uint32_t*) out->values, n); break;
            case 8: hipLaunchKernelGGL(k_gather_fixed<uint64_t>, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t*) src.values, p.rowids, (uint64_t*) out->values, n); break;
            default: hipLaunchKernelGGL(k_gather_fixed<u128x>, dim3(grid), dim3(256), 0, ctx->stream, (const u128x*) src.values, p.rowids, (u128x*) out->values, n); break;
         }
      }
   }
   LDB_HIP(hipGetLastError());
   auto finish_lazy = [&]() -> int32_t {
      for (ldb_column* lc : lazy_small)
         if (lc) LDB_TRY(ldb_column_strings(ctx, *lc, (int64_t) n));
      return LDB_OK;
   };
   if (!n_slots) return finish_lazy();
   std::vector<unsigned long long> words((size_t) n_slots);
   LDB_TRY(LDB_READBACK(ctx, words.data(), d_words, 8 * (size_t) n_slots));
   for (int32_t c = 0; c < n_cols; c++) {
      Pending& p = pend[(size_t) c];
      ldb_column* out = &outs[c];
      if (p.slot_nulls >= 0) {
         out->null_count = (int64_t) words[(size_t) p.slot_nulls];
         if (out->null_count == 0) {
            ldb_dev_free(ctx, out->validity);
            out->validity = nullptr;
         }
      }
      if (p.slot_bytes >= 0) {
         const ldb_column& src = r->sides[(size_t) refs[c].side].table->cols[(size_t) refs[c].col];
         const uint64_t total = words[(size_t) p.slot_bytes];
         ldb_dev_free(ctx, p.lens);
         out->value_bytes = (int64_t) total;
         LDB_TRY(ldb_dev_alloc(ctx, &out->values, (size_t) total));
         hipLaunchKernelGGL(k_str_copy, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) src.values, src.offsets, p.rowids, out->offsets, (uint8_t*) out->values, n,
                            (int64_t) total);
      }
   }
   LDB_HIP(hipGetLastError());
   return finish_lazy();
}
int32_t ldb_gather_column(ldb_ctx* ctx, const ldb_rel* r, ldb_colref ref, ldb_column* out) { return ldb_gather_columns(ctx, r, &ref, 1, out); }

// writes a lazy utf8 column's strings out of its dictionary: string i = dictionary[code i] (a NULL code gives the empty string;
// the column's validity bitmap is separate).  The column object is logically const (the same value, another representation).
int32_t ldb_column_strings(ldb_ctx* ctx, const ldb_column& cc, int64_t n_rows) {
   if (!ldb_column_is_lazy(cc)) return LDB_OK;
   ldb_column& c = const_cast<ldb_column&>(cc);
   const ldb_column& dict = c.dict->cols[0];
   const uint64_t n = (uint64_t) n_rows;
   const int grid = ldb_grid_for(ctx, n_rows, 256, 8);
   int64_t *lens = nullptr, *offsets = nullptr;
   LdbBufs tmp(ctx);
   LDB_TRY(tmp.alloc(&lens, 8 * (size_t) (n + 1)));
   LDB_TRY(tmp.alloc(&offsets, 8 * (size_t) (n + 1)));
   uint64_t* d_total;
   LDB_TRY(ldb_counters(ctx, 1, &d_total));
   if (n) hipLaunchKernelGGL(k_str_lens, dim3(grid), dim3(256), 0, ctx->stream, (const int64_t*) dict.offsets, (const uint32_t*) c.dict_codes, lens, n);
   LDB_TRY(ldb_exclusive_scan_i64(ctx, lens, offsets, n_rows, (int64_t*) d_total));
   uint64_t total = 0;
   LDB_TRY(ldb_read_u64(ctx, d_total, &total));
   void* values = nullptr;
   LDB_TRY(tmp.alloc((uint8_t**) &values, (size_t) total));
   hipLaunchKernelGGL(k_str_copy, dim3(grid > 0 ? grid : 1), dim3(256), 0, ctx->stream, (const uint8_t*) dict.values, (const int64_t*) dict.offsets, (const uint32_t*) c.dict_codes, offsets, (uint8_t*) values, n,
                      (int64_t) total);
   LDB_HIP(hipGetLastError());
   tmp.keep(offsets);
   tmp.keep(values);
   c.offsets = offsets;
   c.values = values;
   c.value_bytes = (int64_t) total;
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_materialize(ldb_ctx* ctx, ldb_rel* r, const ldb_colref* cols, int32_t n_cols, ldb_table** out) {
   if (!ctx || !r || !out || n_cols < 0) LDB_FAIL(LDB_ERR_INVALID, "materialize: bad argument");
   LDB_TRY(ldb_rel_force(ctx, r));
   for (int32_t c = 0; c < n_cols; c++)
      if (cols[c].side < 0 || (size_t) cols[c].side >= r->sides.size() || cols[c].col < 0 || (size_t) cols[c].col >= r->sides[(size_t) cols[c].side].table->cols.size())
         LDB_FAIL(LDB_ERR_INVALID, "materialize: column %d:%d out of range", cols[c].side, cols[c].col);
   auto t = std::make_unique<ldb_table>();
   t->ctx = ctx;
   t->name = "materialized";
   t->n_rows = r->n_rows;
   t->cols.resize((size_t) n_cols);
   int32_t s = ldb_gather_columns(ctx, r, cols, n_cols, t->cols.data());
   if (s != LDB_OK) {
      ldb_gpu_table_release(ctx, t.release());
      return s;
   }
   *out = t.release();
   return LDB_OK;
}

// ---------------------------------------------------------------- exclusive scans
// Small inputs: one workgroup (k_scan_block).  Everything else: ONE launch of a chained scan with decoupled look-back
// (k_scan_chain) — a tile of 2048 elements per workgroup, tile numbers handed out by a ticket counter (a tile only ever
// waits for tiles whose workgroups are already running), per-tile status words {epoch, state, value} that are never
// cleared: every call owns a fresh epoch, and a word of another epoch reads as "not there yet".  The three-level version
// this replaces (per-block sums → recursive scan → add back) was 22 % of all launches of a 22-query pass.
// Status word: value in the low VBITS bits, state (1 = the tile's own sum, 2 = inclusive prefix) above it, epoch on top.
template <typename T, typename TO, typename TT, int VBITS>
__global__ __launch_bounds__(256) void k_scan_chain(const T* __restrict__ in, TO* __restrict__ out, uint64_t n, uint64_t n_tiles, TT* __restrict__ total,
                                                    unsigned long long* __restrict__ status, unsigned long long* __restrict__ ticket, unsigned long long ticket_base,
                                                    unsigned long long epoch) {
   __shared__ unsigned long long s_tile;
   __shared__ TO s_wave[4];
   __shared__ TO s_prefix;
   if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1ull) - ticket_base;
   __syncthreads();
   const uint64_t tile = s_tile;
   const uint64_t base = tile * CHAIN_TILE + (uint64_t) threadIdx.x * CHAIN_ITEMS;
   TO v[CHAIN_ITEMS];
   TO sum = 0;
#pragma unroll
   for (int k = 0; k < CHAIN_ITEMS; k++) {
      v[k] = base + k < n ? (TO) in[base + k] : (TO) 0;
      sum += v[k];
   }
   TO agg;
   TO excl = d_block_scan256<TO>(sum, s_wave, &agg);
   if (threadIdx.x < 64) {
      const TO prefix = d_chain_prefix<TO, VBITS>(status, tile, epoch, agg, threadIdx.x);
      if (threadIdx.x == 0) s_prefix = prefix;
   }
   __syncthreads();
   excl += s_prefix;
   if (total && tile == n_tiles - 1 && threadIdx.x == 0) *total = (TT) (s_prefix + agg);
#pragma unroll
   for (int k = 0; k < CHAIN_ITEMS; k++) {
      if (base + k < n) out[base + k] = excl;
      excl += v[k];
   }
}
// the chain's device state: status words for `tiles` tiles in both formats (32-bit values / 42-bit values) + the ticket
// counter.  Returns the (masked) epoch of this call and the ticket base; grows the status arrays on demand.
int32_t ldb_chain_begin(ldb_ctx* ctx, uint64_t n_tiles, bool wide, ChainCall* c) {
   if (!ctx->scan_ticket) {
      LDB_HIP(hipMalloc((void**) &ctx->scan_ticket, 64));
      LDB_HIP(hipMemsetAsync(ctx->scan_ticket, 0, 64, ctx->stream));
      ctx->scan_ticket_base = 0;
   }
   if (n_tiles > ctx->scan_status_tiles) {
      size_t want = 4096;
      while (want < n_tiles) want *= 2;
      uint64_t* fresh = nullptr;
      LDB_HIP(hipMallocAsync((void**) &fresh, 16 * want, ctx->stream)); // [0, want): 32-bit format, [want, 2 want): 42-bit format
      LDB_HIP(hipMemsetAsync(fresh, 0, 16 * want, ctx->stream));
      if (ctx->scan_status) (void) hipFreeAsync(ctx->scan_status, ctx->stream);
      ctx->scan_status = fresh;
      ctx->scan_status_tiles = want;
   }
   // epochs: 30 bits beside a 32-bit value, 20 bits beside a 42-bit value; epoch 0 is "never written".  When the narrower
   // counter wraps, the words are cleared once (a stale word of the same epoch would read as ready)
   ctx->scan_epoch++;
   const uint32_t mask = wide ? (1u << 20) - 1 : (1u << 30) - 1;
   if ((ctx->scan_epoch & ((1u << 20) - 1)) == 0) {
      LDB_HIP(hipMemsetAsync(ctx->scan_status, 0, 16 * ctx->scan_status_tiles, ctx->stream));
      ctx->scan_epoch++;
   }
   c->status = (unsigned long long*) ctx->scan_status + (wide ? ctx->scan_status_tiles : 0);
   c->ticket = ctx->scan_ticket;
   c->ticket_base = ctx->scan_ticket_base;
   c->epoch = ctx->scan_epoch & mask;
   ctx->scan_ticket_base += n_tiles;
   return LDB_OK;
}
// a launch that did not happen took no tickets: the host mirror of the counter is re-based (the next tile numbers must
// start at 0 again, or every later chain would wait for tiles that never run)
int32_t ldb_chain_failed(ldb_ctx* ctx) {
   (void) hipMemsetAsync(ctx->scan_ticket, 0, 64, ctx->stream);
   ctx->scan_ticket_base = 0;
   LDB_FAIL(LDB_ERR_HIP, "chained scan: the launch failed");
}

// the three-level scan (option scan_single_pass = 0, and the single-workgroup case): per-block sums → recursive scan → add back
template <typename T, typename TO, typename TT>
__global__ void k_scan_block(const T* __restrict__ in, TO* __restrict__ out, TO* __restrict__ block_sums, uint64_t n, TT* __restrict__ total) {
   __shared__ TO sh[256];
   const int ITEMS = 8;
   uint64_t base = (uint64_t) blockIdx.x * 256 * ITEMS + (uint64_t) threadIdx.x * ITEMS;
   TO v[ITEMS];
   TO sum = 0;
#pragma unroll
   for (int k = 0; k < ITEMS; k++) {
      v[k] = base + k < n ? (TO) in[base + k] : (TO) 0;
      sum += v[k];
   }
   sh[threadIdx.x] = sum;
   __syncthreads();
   for (int off = 1; off < 256; off <<= 1) {
      TO t = threadIdx.x >= (unsigned) off ? sh[threadIdx.x - off] : (TO) 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
   }
   TO excl = sh[threadIdx.x] - sum;
   if (threadIdx.x == 255) {
      if (block_sums) block_sums[blockIdx.x] = sh[255];
      if (total && gridDim.x == 1) *total = (TT) sh[255]; // the last level of the recursion holds the grand total
   }
#pragma unroll
   for (int k = 0; k < ITEMS; k++) {
      if (base + k < n) out[base + k] = excl;
      excl += v[k];
   }
}
template <typename TO>
__global__ void k_scan_add(TO* __restrict__ out, const TO* __restrict__ block_offsets, uint64_t n) {
   const int ITEMS = 8;
   uint64_t base = (uint64_t) blockIdx.x * 256 * ITEMS + (uint64_t) threadIdx.x * ITEMS;
   TO add = block_offsets[blockIdx.x];
#pragma unroll
   for (int k = 0; k < ITEMS; k++)
      if (base + k < n) out[base + k] += add;
}
template <typename T, typename TO, typename TT, int VBITS>
static int32_t scan_impl(ldb_ctx* ctx, const T* d_in, TO* d_out, int64_t n, TT* d_total) {
   if (n <= 0) {
      if (d_total) LDB_HIP(hipMemsetAsync(d_total, 0, sizeof(TT), ctx->stream));
      return LDB_OK;
   }
   const int64_t per_block = 256 * 8;
   int64_t nb = (n + per_block - 1) / per_block;
   if (nb > 1 && ldb_option("scan_single_pass", 1) != 0) {
      ChainCall c;
      LDB_TRY(ldb_chain_begin(ctx, (uint64_t) nb, VBITS > 32, &c));
      hipLaunchKernelGGL((k_scan_chain<T, TO, TT, VBITS>), dim3((unsigned) nb), dim3(256), 0, ctx->stream, d_in, d_out, (uint64_t) n, (uint64_t) nb, d_total, c.status, c.ticket, c.ticket_base,
                         c.epoch);
      if (hipGetLastError() != hipSuccess) return ldb_chain_failed(ctx);
      return LDB_OK;
   }
   TO* sums = nullptr;
   if (nb > 1) LDB_TRY(ldb_dev_alloc(ctx, (void**) &sums, sizeof(TO) * (size_t) (nb + 1)));
   hipLaunchKernelGGL((k_scan_block<T, TO, TT>), dim3((unsigned) nb), dim3(256), 0, ctx->stream, d_in, d_out, sums, (uint64_t) n, d_total);
   if (nb > 1) {
      TO* sums_scanned;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &sums_scanned, sizeof(TO) * (size_t) (nb + 1)));
      LDB_TRY((scan_impl<TO, TO, TT, VBITS>(ctx, sums, sums_scanned, nb, d_total)));
      hipLaunchKernelGGL((k_scan_add<TO>), dim3((unsigned) nb), dim3(256), 0, ctx->stream, d_out, sums_scanned, (uint64_t) n);
      ldb_dev_free(ctx, sums_scanned);
      ldb_dev_free(ctx, sums);
   }
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}
int32_t ldb_exclusive_scan_u32(ldb_ctx* ctx, const uint32_t* d_in, uint32_t* d_out, int64_t n, uint64_t* d_total) {
   // the running sums are 32-bit (callers bound their totals by the uint32 row-id space); the total is stored widened
   return scan_impl<uint32_t, uint32_t, uint64_t, 32>(ctx, d_in, d_out, n, d_total);
}
int32_t ldb_exclusive_scan_i64(ldb_ctx* ctx, const int64_t* d_in, int64_t* d_out, int64_t n, int64_t* d_total) {
   // non-negative inputs (string lengths) whose total stays below 2^42 bytes — 15 x the HBM of one MI355X
   return scan_impl<int64_t, int64_t, int64_t, 42>(ctx, d_in, d_out, n, d_total);
}

// ---------------------------------------------------------------- bitmap → ascending row numbers, one launch
// A selection bitmap (one bit per row: ballot words of a probe / filter kernel) becomes the ascending list of set positions:
// popcount, chained scan (above) and expansion in ONE kernel — the word_pop + scan + expand sequence it replaces was five to
// seven launches and two temporaries per join.  A tile is 512 words (32 768 rows).  Dense tiles expand a word per wave
// iteration (the lanes whose bit is set write one coalesced run), sparse tiles a word per lane.  With `match` the kernel also
// writes second[j] = match[row] (the build row of a unique-key join's j-th result pair).  Entries [total, cap) are filled
// with row 0, so that a consumer that was sized from a REPLAYED count (ldb_readback) never meets an uninitialised row id.
// (a tile of 512 words = 32 768 rows, two words per thread: the word-per-wave expansion of a dense tile is then 128 dependent
// steps per wave instead of 512 — a compaction over few tiles runs at that latency)
// (round 6: two words per thread only where the bitmap is short.  A tile costs a fixed latency — ticket, loads, block scan, look-back — whatever
// it holds: the 18 311 tiles of a 600 M-row bitmap took 16 – 27 ns each, 0.3 – 0.5 ms where its 75 MB + the row ids written are 0.05 ms of
// traffic.  Bitmaps of >= 2 048 larger tiles take four or eight words per thread.)
template <int BC_ITEMS>
__global__ __launch_bounds__(256) void k_bitmap_compact(const uint64_t* __restrict__ bitmap, uint64_t n_words, uint64_t n_tiles, uint32_t* __restrict__ out, uint64_t cap,
                                                        const uint32_t* __restrict__ match, uint32_t* __restrict__ second, unsigned long long* __restrict__ total,
                                                        unsigned long long* __restrict__ status, unsigned long long* __restrict__ ticket, unsigned long long ticket_base,
                                                        unsigned long long epoch) {
   __shared__ unsigned long long s_tile;
   __shared__ uint32_t s_wave[4];
   __shared__ uint32_t s_prefix;
   constexpr uint32_t BC_TILE = 256 * BC_ITEMS;
   __shared__ uint32_t s_off[BC_TILE];
   __shared__ uint64_t s_words[BC_TILE]; // the tile's words for the word-per-wave expansion (a wave re-reading them from memory one by
                                            // one runs at memory latency: 512 dependent round trips per wave)
   if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1ull) - ticket_base;
   __syncthreads();
   const uint64_t tile = s_tile;
   const uint64_t word0 = tile * BC_TILE;
   // word w of the tile is handled by thread w % 256 in round w / 256 (coalesced loads); offsets go through LDS
   uint64_t m[BC_ITEMS];
   uint32_t cnt[BC_ITEMS];
   uint32_t sum = 0;
#pragma unroll
   for (int k = 0; k < BC_ITEMS; k++) {
      const uint64_t w = word0 + (uint64_t) k * 256 + threadIdx.x;
      m[k] = w < n_words ? bitmap[w] : 0;
      cnt[k] = (uint32_t) __popcll(m[k]);
      s_off[k * 256 + threadIdx.x] = cnt[k];
      s_words[k * 256 + threadIdx.x] = m[k];
   }
   __syncthreads();
   // thread t scans words [8t, 8t + 8) of the tile (consecutive words → consecutive output positions)
   uint32_t own[BC_ITEMS];
#pragma unroll
   for (int k = 0; k < BC_ITEMS; k++) {
      own[k] = s_off[threadIdx.x * BC_ITEMS + k];
      sum += own[k];
   }
   uint32_t agg;
   uint32_t excl = d_block_scan256<uint32_t>(sum, s_wave, &agg);
   if (threadIdx.x < 64) {
      const uint32_t prefix = d_chain_prefix<uint32_t, 32>(status, tile, epoch, agg, threadIdx.x);
      if (threadIdx.x == 0) s_prefix = prefix;
   }
   __syncthreads();
   const uint32_t tile_base = s_prefix;
   excl += tile_base;
#pragma unroll
   for (int k = 0; k < BC_ITEMS; k++) {
      s_off[threadIdx.x * BC_ITEMS + k] = excl;
      excl += own[k];
   }
   if (tile == n_tiles - 1 && threadIdx.x == 0) {
      if (total) *total = (unsigned long long) tile_base + agg;
   }
   __syncthreads();
   if (agg * 8u < BC_TILE * 64u) { // fewer than one set bit in eight: a word per lane
#pragma unroll
      for (int k = 0; k < BC_ITEMS; k++) {
         uint64_t mm = m[k];
         if (!mm) continue;
         uint32_t at = s_off[k * 256 + threadIdx.x];
         const uint32_t base = (uint32_t) ((word0 + (uint64_t) k * 256 + threadIdx.x) * 64);
         while (mm) {
            const uint32_t row = base + (uint32_t) __builtin_ctzll(mm);
            if (at < cap) { // (cap < the real count only under a wrong replayed count: never write past the buffer)
               out[at] = row;
               if (match) second[at] = match[row];
            }
            at++;
            mm &= mm - 1;
         }
      }
   } else { // a word per wave iteration
      const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (uint32_t w = wave; w < BC_TILE; w += 4) {
         if (word0 + w >= n_words) break;
         const uint64_t mm = s_words[w]; // wave-uniform
         if ((mm >> lane) & 1) {
            const uint32_t at = s_off[w] + d_rank_in(mm);
            const uint32_t row = (uint32_t) ((word0 + w) * 64 + lane);
            if (at < cap) {
               out[at] = row;
               if (match) second[at] = match[row];
            }
         }
      }
   }
   if (tile == n_tiles - 1) { // the tail behind the real count (see above)
      const uint64_t end = (uint64_t) tile_base + agg;
      for (uint64_t i = end + threadIdx.x; i < cap; i += 256) {
         out[i] = 0;
         if (match) second[i] = 0;
      }
   }
}
int32_t ldb_bitmap_compact(ldb_ctx* ctx, const uint64_t* bitmap, int64_t n_words, uint32_t* out, uint64_t cap, const uint32_t* match, uint32_t* second, uint64_t* d_total) {
   if (n_words <= 0) {
      if (d_total) LDB_HIP(hipMemsetAsync(d_total, 0, 8, ctx->stream));
      return LDB_OK;
   }
   const int64_t wide = ldb_option("compact_wide_tiles", 1); // 0 = two words per thread always, 1 = the default floor of 2 048 tiles, n > 1 = that floor
   const int64_t floor_tiles = wide > 1 ? wide : 2048;
   const int64_t most = ldb_option("compact_max_words", 8);
   const int items = !wide ? 2 : n_words >= floor_tiles * 256 * 16 && most >= 16 ? 16 : n_words >= floor_tiles * 256 * 8 ? 8 : n_words >= floor_tiles * 256 * 4 ? 4 : 2;
   const uint64_t n_tiles = ((uint64_t) n_words + 256u * items - 1) / (256u * items);
   ChainCall c;
   LDB_TRY(ldb_chain_begin(ctx, n_tiles, false, &c));
   LdbProf prof_(ctx, "k_bitmap_compact");
#define LDB_BC_LAUNCH(I) \
   hipLaunchKernelGGL(k_bitmap_compact<I>, dim3((unsigned) n_tiles), dim3(256), 0, ctx->stream, bitmap, (uint64_t) n_words, n_tiles, out, cap, match, second, (unsigned long long*) d_total, c.status, \
                      c.ticket, c.ticket_base, c.epoch)
   if (items == 16) LDB_BC_LAUNCH(16);
   else if (items == 8) LDB_BC_LAUNCH(8);
   else if (items == 4) LDB_BC_LAUNCH(4);
   else LDB_BC_LAUNCH(2);
#undef LDB_BC_LAUNCH
   if (hipGetLastError() != hipSuccess) return ldb_chain_failed(ctx);
   return LDB_OK;
}
