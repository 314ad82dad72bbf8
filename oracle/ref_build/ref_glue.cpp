// ref_glue.cpp — C entry points over the REFERENCE'S OWN runtime objects (compiled from
// /root/reference by build_ref.sh into oracle/_ref/libldb_ref.so).  TEST INFRASTRUCTURE: used to
// validate the C restatement in oracle/ldb_oracle.c against the real implementation where the
// reference compiles offline:
//   * dbHashApplyColumn            src/runtime/Hash.cpp           (runtime twin of the compiled db.hash)
//   * Restrictions::create/applyFilters + Filter impls   src/runtime/storage/Restrictions.cpp
//   * HashIndexedView::build + tag helpers + bloomMasks  src/runtime/LazyJoinHashtable.cpp, helpers.{h,cpp}
//   * PreAggregationHashtableFragment::insert + PreAggregationHashtable::merge   src/runtime/PreAggregationHashtable.cpp
//   * GrowingBuffer / FlexibleBuffer / ThreadLocal / ExecutionContext            src/runtime/*.cpp
// The JIT-generated per-tuple loops (scan callback, probe loop, fragment lookup) have no C++ source
// in the reference; they are restated HERE following the lowerings cited at each loop, around the
// real data structures.  Nothing from this file is shipped or linked into the product.
#include <mutex>

#include "lingodb/runtime/ArrowView.h"
#include "lingodb/runtime/ExecutionContext.h"
#include "lingodb/runtime/GrowingBuffer.h"
#include "lingodb/runtime/Hash.h"
#include "lingodb/runtime/LazyJoinHashtable.h"
#include "lingodb/runtime/PreAggregationHashtable.h"
#include "lingodb/runtime/ThreadLocal.h"
#include "lingodb/runtime/helpers.h"
#include "lingodb/runtime/Heap.h"
#include "lingodb/runtime/Hashtable.h"
#include "lingodb/runtime/HashMultiMap.h"
#include "lingodb/runtime/SimpleState.h"
#include "lingodb/runtime/SegmentTreeView.h"
#include "lingodb/runtime/DateRuntime.h"
#include "lingodb/runtime/StringRuntime.h"
#include "lingodb/runtime/storage/Restrictions.h"
#include "lingodb/scheduler/Scheduler.h"
#include "lingodb/scheduler/Tasks.h"

#include <arrow/api.h>
#include <arrow/c/bridge.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>

using namespace lingodb;

namespace {
// ExecutionContext only stores the Session reference; a session is never touched on this path.
alignas(64) char g_fakeSession[4096];
struct CtxScope {
   std::unique_ptr<scheduler::SchedulerHandle> handle;
   std::unique_ptr<runtime::ExecutionContext> ctx;
   explicit CtxScope(int threads) {
      handle = scheduler::startScheduler((size_t) threads);
      ctx = std::make_unique<runtime::ExecutionContext>(*reinterpret_cast<runtime::Session*>(g_fakeSession));
      runtime::setCurrentExecutionContext(ctx.get());
   }
   ~CtxScope() {
      ctx.reset();
      runtime::setCurrentExecutionContext(nullptr);
   }
};

// morsel task: units of `unit` rows handed out by an atomic cursor
class RangeTask : public scheduler::TaskWithImplicitContext {
   std::atomic<size_t> next{0};
   size_t n, unit;
   std::function<void(size_t, size_t, size_t)> fn; // (begin, end, worker)
   std::vector<size_t> resv;

   public:
   RangeTask(size_t n, size_t unit, std::function<void(size_t, size_t, size_t)> fn) : n(n), unit(unit), fn(std::move(fn)), resv(scheduler::getNumWorkers(), 0) {}
   bool allocateWork() override {
      size_t b = next.fetch_add(unit);
      if (b >= n) {
         workExhausted.store(true);
         return false;
      }
      resv[scheduler::currentWorkerId()] = b;
      return true;
   }
   void performWork() override {
      size_t b = resv[scheduler::currentWorkerId()];
      fn(b, std::min(n, b + unit), scheduler::currentWorkerId());
   }
};

// TableChunk's flattening of one Arrow column into an ArrayView (LingoDBTable.cpp:200-225)
struct ColumnView {
   runtime::ArrayView view;
   const void* bufs[3];
};
void makeView(const arrow::Array& arr, ColumnView& cv) {
   auto data = arr.data();
   cv.view.length = data->length;
   cv.view.nullCount = data->null_count;
   cv.view.offset = data->offset;
   cv.view.nBuffers = (int64_t) data->buffers.size();
   cv.view.nChildren = 0;
   cv.view.children = nullptr;
   for (size_t b = 0; b < 3; b++) cv.bufs[b] = b < data->buffers.size() && data->buffers[b] ? data->buffers[b]->data() : nullptr;
   if (!cv.bufs[0]) cv.bufs[0] = runtime::ArrayView::validData.data(); // shared all-valid bitmap (:213-218)
   cv.view.buffers = cv.bufs;
}
} // namespace

extern "C" {

// ---- db.hash twin: fold one Arrow column into running[] (Hash.cpp:58-239)
int32_t ref_hash_column(struct ArrowSchema* schema, struct ArrowArray* array, uint64_t* running, int64_t n) {
   auto res = arrow::ImportArray(array, schema);
   if (!res.ok()) return -1;
   std::vector<uint64_t> r(running, running + n);
   runtime::dbHashApplyColumn(r, **res);
   std::memcpy(running, r.data(), sizeof(uint64_t) * (size_t) n);
   return 0;
}

uint16_t ref_bloom_mask(uint32_t idx) { return runtime::bloomMasks[idx & 2047]; }

// ---- pushed-down filters on a record batch (Restrictions.cpp) driven by the scan unit loop of
// ScanBatchesTask::unitRun (LingoDBTable.cpp:382-407).  Filter constants: kind 0 = string,
// 1 = int64, 2 = double; IN lists as arrays.
struct RefFilter {
   const char* column;
   int32_t op; // lingodb::runtime::FilterOp
   int32_t kind;
   const char* sval;
   int64_t ival;
   double dval;
   int32_t n_in;
   const char* const* in_s;
   const int64_t* in_i;
};
int64_t ref_scan_filter(struct ArrowSchema* schema, struct ArrowArray* batch, const RefFilter* filters, int32_t n_filters, uint32_t* out_rows, int32_t threads) {
   auto rb = arrow::ImportRecordBatch(batch, schema);
   if (!rb.ok()) return -1;
   auto& table = **rb;
   std::vector<runtime::FilterDescription> descs;
   for (int32_t f = 0; f < n_filters; f++) {
      runtime::FilterDescription d{};
      d.columnName = filters[f].column;
      d.columnId = 0;
      d.op = (runtime::FilterOp) filters[f].op;
      if (d.op == runtime::FilterOp::IN) {
         if (filters[f].kind == 0) {
            std::vector<std::string> v;
            for (int k = 0; k < filters[f].n_in; k++) v.emplace_back(filters[f].in_s[k]);
            d.values = v;
         } else {
            std::vector<int64_t> v(filters[f].in_i, filters[f].in_i + filters[f].n_in);
            d.values = v;
         }
      } else if (filters[f].kind == 0) {
         d.value = std::string(filters[f].sval ? filters[f].sval : "");
      } else if (filters[f].kind == 1) {
         d.value = filters[f].ival;
      } else {
         d.value = filters[f].dval;
      }
      descs.push_back(d);
   }
   std::unique_ptr<runtime::Restrictions> restrictions;
   try {
      restrictions = runtime::Restrictions::create(descs, *table.schema());
   } catch (std::exception&) { return -2; }
   std::vector<ColumnView> views((size_t) table.num_columns());
   for (int c = 0; c < table.num_columns(); c++) makeView(*table.column(c), views[(size_t) c]);
   const size_t n = (size_t) table.num_rows();
   const size_t unit = 20000; // splitSize, LingoDBTable.cpp:364
   std::vector<uint32_t> staged(n ? n : 1);
   std::vector<size_t> counts((n + unit - 1) / unit + 1, 0);
   CtxScope scope(threads);
   scheduler::awaitEntryTask(std::make_unique<RangeTask>(n, unit, [&](size_t b, size_t e, size_t) {
      uint16_t sv1[65536], sv2[65536];
      auto [len, sel] = restrictions->applyFilters(b, e - b, sv1, sv2, [&](size_t colId) { return &views[colId].view; });
      for (size_t i = 0; i < len; i++) staged[b + i] = (uint32_t) (b + sel[i]);
      counts[b / unit] = len;
   }));
   int64_t total = 0;
   for (size_t u = 0; u * unit < n; u++) {
      if (out_rows) std::memcpy(out_rows + total, staged.data() + u * unit, sizeof(uint32_t) * counts[u]);
      total += (int64_t) counts[u];
   }
   return total;
}

// ---- hash join: real GrowingBuffer + HashIndexedView::build, probe loop restated from
// LookupHashIndexedViewLowering / ScanListLowering (SubOpToControlFlow.cpp:2558-2586, 2254-2313).
// Build rows: {next, hash, key:int64, row:uint64} (MultiMapAsHashIndexedView layout,
// SpecializeSubOpPass.cpp:70-84).  Single int64 key → no hash compare (:110-118).
struct JoinEntry {
   JoinEntry* next;
   uint64_t hash;
   int64_t key;
   uint64_t row;
};
struct ViewLayout { // what generated code reads: {Entry** ht; size_t mask} (LazyJoinHashtable.h:13-14)
   JoinEntry** ht;
   size_t mask;
};
int64_t ref_join_int64(const int64_t* bkeys, const uint64_t* bhash, int64_t nb, const int64_t* pkeys, const uint64_t* phash, int64_t np, uint32_t* out_probe,
                       uint32_t* out_build, int64_t cap, int32_t threads) {
   CtxScope scope(threads);
   // build pipeline: thread-local GrowingBuffers, merged, then indexed
   auto* tl = runtime::GrowingBuffer::createThreadLocal(sizeof(JoinEntry));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) nb, 20000, [&](size_t b, size_t e, size_t) {
      auto* buf = reinterpret_cast<runtime::GrowingBuffer*>(tl->getLocal());
      for (size_t i = b; i < e; i++) {
         auto* en = reinterpret_cast<JoinEntry*>(buf->insert());
         en->next = nullptr;
         en->hash = bhash[i];
         en->key = bkeys[i];
         en->row = i;
      }
   }));
   auto* merged = runtime::GrowingBuffer::merge(tl);
   auto* view = runtime::HashIndexedView::build(merged);
   auto* lay = reinterpret_cast<ViewLayout*>(view);
   std::atomic<int64_t> cursor{0};
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) np, 20000, [&](size_t b, size_t e, size_t) {
      for (size_t i = b; i < e; i++) {
         uint64_t h = phash[i];
         JoinEntry* slot = lay->ht[h & lay->mask];
         JoinEntry* cur = runtime::matchesTag(slot, h) ? runtime::untag(slot) : nullptr; // = filterTagged (helpers.h:338-342)
         for (; cur; cur = cur->next) {
            if (cur->key == pkeys[i]) {
               int64_t idx = cursor.fetch_add(1);
               if (idx < cap) {
                  out_probe[idx] = (uint32_t) i;
                  out_build[idx] = (uint32_t) cur->row;
               }
            }
         }
      }
   }));
   return cursor.load();
}

// ---- group-by: real PreAggregationHashtableFragment (per worker via ThreadLocal) + merge.
// Per-tuple code restated from LookupPreAggrHtFragment (SubOpToControlFlow.cpp:3065-3157):
// slot = ht[(hash >> 6) & 1023]; hit = hash equal && keys equal → reduce; miss → insert + init.
// Entry content: {key:int64, sum:int64, count:int64}.
struct AggContent {
   int64_t key, sum, count;
};
static bool aggEq(uint8_t* a, uint8_t* b) { return reinterpret_cast<AggContent*>(a)->key == reinterpret_cast<AggContent*>(b)->key; }
static void aggCombine(uint8_t* dst, uint8_t* src) {
   reinterpret_cast<AggContent*>(dst)->sum += reinterpret_cast<AggContent*>(src)->sum;
   reinterpret_cast<AggContent*>(dst)->count += reinterpret_cast<AggContent*>(src)->count;
}
int64_t ref_groupby_int64(const int64_t* keys, const uint64_t* hashes, const int64_t* vals, int64_t n, int64_t* out_keys, int64_t* out_sums, int64_t* out_counts,
                          int64_t cap, int32_t threads) {
   using Fragment = runtime::PreAggregationHashtableFragment;
   CtxScope scope(threads);
   const size_t typeSize = sizeof(Fragment::Entry) + sizeof(AggContent);
   auto* tl = runtime::ThreadLocal::create([](uint8_t* arg) -> uint8_t* { return reinterpret_cast<uint8_t*>(Fragment::create(*reinterpret_cast<size_t*>(arg), false)); },
                                           reinterpret_cast<uint8_t*>(const_cast<size_t*>(&typeSize)));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n, 20000, [&](size_t b, size_t e, size_t) {
      auto* frag = reinterpret_cast<Fragment*>(tl->getLocal());
      for (size_t i = b; i < e; i++) {
         uint64_t h = hashes[i];
         Fragment::Entry* en = frag->ht[(h >> 6) & (Fragment::hashtableSize - 1)];
         if (!(en && en->hashValue == h && reinterpret_cast<AggContent*>(en->content)->key == keys[i])) {
            en = frag->insert(h);
            auto* c = reinterpret_cast<AggContent*>(en->content);
            c->key = keys[i];
            c->sum = 0;
            c->count = 0;
         }
         auto* c = reinterpret_cast<AggContent*>(en->content);
         c->sum += vals[i];
         c->count += 1;
      }
   }));
   auto* merged = runtime::PreAggregationHashtable::merge(tl, aggEq, aggCombine);
   // scan of the merged table (ScanPreAggrHtLowering, :2110): the buffer holds entry pointers
   struct Out {
      int64_t *keys, *sums, *counts;
      int64_t n, cap;
   } o{out_keys, out_sums, out_counts, 0, cap};
   auto* it = merged->createIterator();
   runtime::BufferIterator::iterate(
      it, false,
      [](runtime::Buffer buf, void* arg) {
         auto* st = reinterpret_cast<Out*>(arg);
         auto** entries = reinterpret_cast<Fragment::Entry**>(buf.ptr);
         size_t cnt = buf.numElements / sizeof(Fragment::Entry*); // iterator buffers carry a byte length
         for (size_t k = 0; k < cnt; k++) {
            auto* c = reinterpret_cast<AggContent*>(entries[k]->content);
            if (st->n < st->cap) {
               st->keys[st->n] = c->key;
               st->sums[st->n] = c->sum;
               st->counts[st->n] = c->count;
            }
            st->n++;
         }
      },
      &o);
   return o.n;
}

// StringRuntime::like (src/runtime/StringRuntime.cpp:134-136) and DateRuntime::extractYear
// (src/runtime/DateRuntime.cpp:99-101) — the real functions the generated code calls for LIKE and
// extract(year from …); pin the oracle's restatements of both.
int32_t ref_like(const char* str, int64_t str_len, const char* pat, int64_t pat_len) {
   return runtime::StringRuntime::like(runtime::VarLen32::fromDataAndLen(str, (size_t) str_len, runtime::StorageClass::TRANSIENT), runtime::VarLen32::fromDataAndLen(pat, (size_t) pat_len, runtime::StorageClass::TRANSIENT)) ? 1 : 0;
}
int64_t ref_extract_year(int64_t date_ns) { return runtime::DateRuntime::extractYear(date_ns); }
// StringRuntime::substr (src/runtime/StringRuntime.cpp:292-319): the real function behind SUBSTRING(… FROM … FOR …)
int64_t ref_substr(const char* str, int64_t str_len, int64_t from, int64_t len, char* out, int64_t cap) {
   runtime::VarLen32 r = runtime::StringRuntime::substr(runtime::VarLen32::fromDataAndLen(str, (size_t) str_len, runtime::StorageClass::TRANSIENT), from, len);
   const int64_t n = (int64_t) r.getLen();
   if (n <= cap) memcpy(out, r.data(), (size_t) n);
   return n;
}


// ---- sort / top-k / key-less aggregation / generic hash map through the reference's own objects
// Rows = {int64 keys[k], int64 row number}; the comparator restates db.sort_compare (LowerToStd.cpp:
// 1046-1064: per key lt / eq selects, DESC by operand swap) as the generated `bool(uint8_t*, uint8_t*)`
// the reference passes to GrowingBuffer::sort / Heap::create.  The row number is the last (ascending)
// key so that the order is total: std::sort and the heap leave ties unspecified.
static int32_t g_sort_k;
static const int32_t* g_sort_desc;
static bool sortCmp(uint8_t* l, uint8_t* r) {
   const int64_t *a = reinterpret_cast<const int64_t*>(l), *b = reinterpret_cast<const int64_t*>(r);
   for (int32_t j = 0; j < g_sort_k; j++) {
      const int64_t x = g_sort_desc[j] ? b[j] : a[j], y = g_sort_desc[j] ? a[j] : b[j];
      if (x < y) return true;
      if (!(x == y)) return false;
   }
   return a[g_sort_k] < b[g_sort_k];
}
// GrowingBuffer::insert x n, GrowingBuffer::sort (GrowingBuffer.cpp:54-78 → parallelSort, Sorting.cpp:343-393, above 512 rows)
int32_t ref_sort_rows(const int64_t* keys, int32_t k, const int32_t* desc, int64_t n, int32_t threads, uint32_t* out_perm) {
   CtxScope scope(threads);
   g_sort_k = k;
   g_sort_desc = desc;
   const size_t typeSize = sizeof(int64_t) * (size_t) (k + 1);
   int32_t rc = 0;
   scheduler::awaitEntryTask(std::make_unique<RangeTask>(1, 1, [&](size_t, size_t, size_t) {
      auto* gb = runtime::GrowingBuffer::create(runtime::GrowingBufferAllocator::getDefaultAllocator(), typeSize, 1024);
      for (int64_t i = 0; i < n; i++) {
         int64_t* row = reinterpret_cast<int64_t*>(gb->insert());
         memcpy(row, keys + i * k, sizeof(int64_t) * (size_t) k);
         row[k] = i;
      }
      runtime::Buffer sorted = gb->sort(sortCmp);
      if (sorted.numElements != typeSize * (size_t) n) {
         rc = -1;
         return;
      }
      for (int64_t i = 0; i < n; i++) out_perm[i] = (uint32_t) reinterpret_cast<const int64_t*>(sorted.ptr + (size_t) i * typeSize)[k];
   }));
   return rc;
}
// Heap::create / insert per worker, Heap::merge, getBuffer (Heap.cpp:8-72): the k_top first rows of the order
int64_t ref_topk_rows(const int64_t* keys, int32_t k, const int32_t* desc, int64_t n, int64_t k_top, int32_t threads, uint32_t* out_perm) {
   CtxScope scope(threads);
   g_sort_k = k;
   g_sort_desc = desc;
   const size_t typeSize = sizeof(int64_t) * (size_t) (k + 1);
   struct Arg {
      size_t kTop, typeSize;
   } arg{(size_t) k_top, typeSize};
   auto* tl = runtime::ThreadLocal::create([](uint8_t* a) -> uint8_t* {
      auto* x = reinterpret_cast<Arg*>(a);
      return reinterpret_cast<uint8_t*>(runtime::Heap::create(x->kTop, x->typeSize, sortCmp));
   }, reinterpret_cast<uint8_t*>(&arg));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n, 20000, [&](size_t b, size_t e, size_t) {
      auto* heap = reinterpret_cast<runtime::Heap*>(tl->getLocal());
      std::vector<int64_t> row((size_t) k + 1);
      for (size_t i = b; i < e; i++) {
         memcpy(row.data(), keys + i * (size_t) k, sizeof(int64_t) * (size_t) k);
         row[(size_t) k] = (int64_t) i;
         heap->insert(reinterpret_cast<uint8_t*>(row.data()));
      }
   }));
   int64_t got = 0;
   scheduler::awaitEntryTask(std::make_unique<RangeTask>(1, 1, [&](size_t, size_t, size_t) {
      runtime::Heap* merged = runtime::Heap::merge(tl);
      if (!merged) return;
      runtime::Buffer buf = merged->getBuffer();
      got = (int64_t) (buf.numElements / typeSize);
      for (int64_t i = 0; i < got; i++) out_perm[i] = (uint32_t) reinterpret_cast<const int64_t*>(buf.ptr + (size_t) i * typeSize)[k];
   }));
   return got;
}
// SimpleState::create per worker + SimpleState::merge with the generated combine (SimpleState.cpp:8-30,
// MergeThreadLocalSimpleState, SubOpToControlFlow.cpp:1733): key-less SUM + COUNT over the rows passing `keep`
struct SumCount {
   __int128 sum;
   int64_t count;
};
int32_t ref_simple_state_sum(const int64_t* vals, const uint8_t* keep, int64_t n, int32_t threads, int64_t out_lohi[2], int64_t* out_count) {
   CtxScope scope(threads);
   auto* tl = runtime::ThreadLocal::create([](uint8_t*) -> uint8_t* {
      auto* st = reinterpret_cast<SumCount*>(runtime::SimpleState::create(sizeof(SumCount)));
      st->sum = 0;
      st->count = 0;
      return reinterpret_cast<uint8_t*>(st);
   }, nullptr);
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n, 20000, [&](size_t b, size_t e, size_t) {
      auto* st = reinterpret_cast<SumCount*>(tl->getLocal());
      for (size_t i = b; i < e; i++)
         if (!keep || keep[i]) {
            st->sum += vals[i];
            st->count++;
         }
   }));
   SumCount total{0, 0};
   scheduler::awaitEntryTask(std::make_unique<RangeTask>(1, 1, [&](size_t, size_t, size_t) {
      auto* m = reinterpret_cast<SumCount*>(runtime::SimpleState::merge(tl, [](uint8_t* dst, uint8_t* src) {
         reinterpret_cast<SumCount*>(dst)->sum += reinterpret_cast<SumCount*>(src)->sum;
         reinterpret_cast<SumCount*>(dst)->count += reinterpret_cast<SumCount*>(src)->count;
      }));
      if (m) total = *m;
   }));
   out_lohi[0] = (int64_t) (uint64_t) total.sum;
   out_lohi[1] = (int64_t) (total.sum >> 64);
   *out_count = total.count;
   return 0;
}
// the generic Hashtable (Hashtable.cpp:16-150: chained, doubling, thread-local tables merged with
// mergeEntries): group-by of int64 keys with SUM + COUNT, as the non-pre-aggregated HashMap lowering uses it
int64_t ref_hashtable_groupby_int64(const int64_t* keys, const uint64_t* hashes, const int64_t* vals, int64_t n, int64_t* out_keys, int64_t* out_sums, int64_t* out_counts,
                                    int64_t cap, int32_t threads) {
   CtxScope scope(threads);
   const size_t typeSize = sizeof(void*) + sizeof(size_t) + sizeof(AggContent); // Entry{next, hashValue, content}
   auto* tl = runtime::ThreadLocal::create([](uint8_t* arg) -> uint8_t* {
      auto* ht = runtime::Hashtable::create(*reinterpret_cast<size_t*>(arg), 16);
      ht->setEqFn(aggEq);
      return reinterpret_cast<uint8_t*>(ht);
   }, reinterpret_cast<uint8_t*>(const_cast<size_t*>(&typeSize)));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n, 20000, [&](size_t b, size_t e, size_t) {
      auto* ht = reinterpret_cast<runtime::Hashtable*>(tl->getLocal());
      for (size_t i = b; i < e; i++) {
         AggContent probe{keys[i], 0, 0};
         const size_t before = ht->size();
         auto* c = reinterpret_cast<AggContent*>(ht->lookUpOrInsert(hashes[i], reinterpret_cast<uint8_t*>(&probe)));
         if (ht->size() != before) { // a fresh entry: the generated code initialises the new group's state here
            c->key = keys[i];
            c->sum = 0;
            c->count = 0;
         }
         c->sum += vals[i];
         c->count += 1;
      }
   }));
   int64_t got = 0;
   scheduler::awaitEntryTask(std::make_unique<RangeTask>(1, 1, [&](size_t, size_t, size_t) {
      auto* merged = runtime::Hashtable::merge(tl, aggEq, aggCombine);
      if (!merged) return;
      struct Out {
         int64_t *keys, *sums, *counts;
         int64_t n, cap;
         size_t typeSize;
      } o{out_keys, out_sums, out_counts, 0, cap, typeSize};
      runtime::BufferIterator::iterate(
         merged->createIterator(), false,
         [](runtime::Buffer buf, void* arg) {
            auto* st = reinterpret_cast<Out*>(arg);
            const size_t cnt = buf.numElements / st->typeSize;
            for (size_t k = 0; k < cnt; k++) {
               auto* c = reinterpret_cast<AggContent*>(buf.ptr + k * st->typeSize + sizeof(void*) + sizeof(size_t));
               if (st->n < st->cap) {
                  st->keys[st->n] = c->key;
                  st->sums[st->n] = c->sum;
                  st->counts[st->n] = c->count;
               }
               st->n++;
            }
         },
         &o);
      got = o.n;
   }));
   return got;
}


// The REAL SegmentTreeView (src/runtime/SegmentTreeView.cpp:18-110) over entries {int64 value, valid}: the generated
// createInitialStateFn / combineStatesFn of WindowLowering's aggregate functions (RelAlgToSubOp.cpp:1843-2027: a NULL
// state is the identity, SUM adds, MIN / MAX select, COUNT counts the non-NULL entries) restated as C callbacks, then
// lookup(from[q], to[q]) per query.  fn: 1 SUM, 2 MIN, 3 MAX, 4 COUNT (the ids of include/lingodb_gpu.h's ldb_window_fn_kind).
extern "C++" {
namespace {
struct WinEntry {
   int64_t v;
   int64_t ok;
};
template <int FN>
void winInit(unsigned char* state, unsigned char* entry) {
   auto* e = reinterpret_cast<WinEntry*>(entry);
   auto* s = reinterpret_cast<WinEntry*>(state);
   if (FN == 4) *s = {e->ok ? 1 : 0, 1};
   else *s = {e->ok ? e->v : 0, e->ok};
}
template <int FN>
void winCombine(unsigned char* out, unsigned char* l, unsigned char* r) {
   const WinEntry a = *reinterpret_cast<WinEntry*>(l), b = *reinterpret_cast<WinEntry*>(r);
   WinEntry o;
   if (!a.ok) o = b;
   else if (!b.ok) o = a;
   else if (FN == 2) o = {b.v < a.v ? b.v : a.v, 1};
   else if (FN == 3) o = {b.v > a.v ? b.v : a.v, 1};
   else o = {(int64_t) ((uint64_t) a.v + (uint64_t) b.v), 1};
   *reinterpret_cast<WinEntry*>(out) = o;
}
} // namespace
} // extern "C++"
int32_t ref_segment_tree(const int64_t* vals, const uint8_t* valid, int64_t n, int32_t fn, const int64_t* from, const int64_t* to, int64_t nq, int64_t* out_vals, uint8_t* out_valid) {
   CtxScope scope(1);
   std::vector<WinEntry> entries((size_t) n);
   for (int64_t i = 0; i < n; i++) entries[(size_t) i] = {vals[i], valid ? valid[i] : 1};
   runtime::Buffer buf{(size_t) n * sizeof(WinEntry), reinterpret_cast<uint8_t*>(entries.data())};
   runtime::SegmentTreeView* view = nullptr;
   switch (fn) {
      case 1: view = runtime::SegmentTreeView::build(buf, sizeof(WinEntry), winInit<1>, winCombine<1>, sizeof(WinEntry)); break;
      case 2: view = runtime::SegmentTreeView::build(buf, sizeof(WinEntry), winInit<2>, winCombine<2>, sizeof(WinEntry)); break;
      case 3: view = runtime::SegmentTreeView::build(buf, sizeof(WinEntry), winInit<3>, winCombine<3>, sizeof(WinEntry)); break;
      case 4: view = runtime::SegmentTreeView::build(buf, sizeof(WinEntry), winInit<4>, winCombine<4>, sizeof(WinEntry)); break;
      default: return -1;
   }
   for (int64_t q = 0; q < nq; q++) {
      WinEntry res{0, 0};
      view->lookup(reinterpret_cast<uint8_t*>(&res), (size_t) from[q], (size_t) to[q]);
      out_vals[q] = res.v;
      out_valid[q] = (uint8_t) (res.ok ? 1 : 0);
   }
   return 0;
}
// ---------------------------------------------------------------- HashMultiMap: outer joins that keep the build side
// translateHJWithMarker (RelAlgToSubOp.cpp:1248-1287): the LEFT input is inserted into a MultiMap [keys] → [values + flag]
// (subop.insert: one ENTRY per distinct key — found with the eq function — and one VALUE per row), every tuple of the right
// input looks its key up, walks the values of the matching entry, emits the pairs and scatters flag = true; afterwards the map
// is scanned and the rows whose flag is still false are emitted NULL-extended.  The container is the reference's own
// (src/runtime/HashMultiMap.cpp, compiled in place); the generated per-tuple code — hash the key as db.hash does, compare the
// stored hash and the key, walk next / valueList — is restated here around it.  NULL keys: with nullsEqual = false the eq
// function is never true for them, so every NULL-key build row becomes an entry of its own that no probe reaches.
// (The entry / value structs are private to the class: this file is compiled with -fno-access-control.)
int64_t ref_hmm_outer_join(const int64_t* bkeys, const uint8_t* bvalid, int64_t nb, const int64_t* pkeys, const uint8_t* pvalid, int64_t np, int64_t initial_capacity, int64_t* out_probe,
                           int64_t* out_build, int64_t cap, int64_t* unmatched_build, int64_t* n_unmatched_build, uint8_t* probe_matched) {
   CtxScope scope(1);
   struct KeyPart {
      int64_t key;
      uint8_t valid;
   };
   struct ValPart {
      int64_t row;
      uint8_t flag;
   };
   using HMM = runtime::HashMultiMap;
   HMM* hmm = HMM::create(sizeof(HMM::Entry) + sizeof(KeyPart), sizeof(HMM::Value) + sizeof(ValPart), (size_t) (initial_capacity > 0 ? initial_capacity : 4));
   auto hashOf = [](int64_t k) { // db.hash of one integer (Hash.cpp:25-28, LowerToLLVM.cpp:493-503)
      const uint64_t m = 11400714819323198549ull * (uint64_t) k;
      return (size_t) (m ^ __builtin_bswap64(m));
   };
   auto find = [&](size_t h, int64_t key) -> HMM::Entry* {
      for (HMM::Entry* e = runtime::filterTagged(hmm->ht.at(h & hmm->hashMask), h); e; e = e->next) {
         auto* kp = reinterpret_cast<KeyPart*>(e->keyContent);
         if (e->hashValue == h && kp->valid && kp->key == key) return e;
      }
      return nullptr;
   };
   for (int64_t i = 0; i < nb; i++) {
      const bool ok = !bvalid || bvalid[i];
      const size_t h = ok ? hashOf(bkeys[i]) : 0; // (a NULL key part is skipped by db.hash: the running hash stays 0)
      HMM::Entry* e = ok ? find(h, bkeys[i]) : nullptr;
      if (!e) {
         e = hmm->insertEntry(h);
         auto* kp = reinterpret_cast<KeyPart*>(e->keyContent);
         kp->key = ok ? bkeys[i] : 0;
         kp->valid = ok ? 1 : 0;
      }
      HMM::Value* v = hmm->insertValue(e);
      auto* vp = reinterpret_cast<ValPart*>(v->valueContent);
      vp->row = i;
      vp->flag = 0;
   }
   int64_t n_pairs = 0;
   for (int64_t j = 0; j < np; j++) {
      probe_matched[j] = 0;
      if (pvalid && !pvalid[j]) continue;
      HMM::Entry* e = find(hashOf(pkeys[j]), pkeys[j]);
      if (!e) continue;
      for (HMM::Value* v = e->valueList; v; v = v->nextValue) {
         auto* vp = reinterpret_cast<ValPart*>(v->valueContent);
         vp->flag = 1;
         probe_matched[j] = 1;
         if (n_pairs < cap) {
            out_probe[n_pairs] = j;
            out_build[n_pairs] = vp->row;
         }
         n_pairs++;
      }
   }
   // the scan of the map: entries in the order of the entry buffer, the values of an entry in list order
   int64_t nu = 0;
   hmm->entries.iterate([&](uint8_t* raw) {
      auto* e = reinterpret_cast<HMM::Entry*>(raw);
      for (HMM::Value* v = e->valueList; v; v = v->nextValue) {
         auto* vp = reinterpret_cast<ValPart*>(v->valueContent);
         if (!vp->flag) unmatched_build[nu++] = vp->row;
      }
   });
   *n_unmatched_build = nu;
   return n_pairs;
}

// =====================================================================================================================================
// CPU baseline legs over the reference's REAL runtime objects at the bench's own scale (bench.py cpu_baseline, kind "reference"):
// TPC-H Q1, Q6 and Q3 as the generated pipelines run them — morsels of 20 000 rows through ScanBatchesTask's unit loop (LingoDBTable.cpp:382-407)
// with the REAL Restrictions::applyFilters, per-tuple code restated from the lowerings cited at each loop (the JIT-generated loops have no C++
// source), around the REAL PreAggregationHashtableFragment / PreAggregationHashtable::merge, GrowingBuffer, HashIndexedView::build and the
// scheduler interface.  Compiled -O2 like the rest of oracle/_ref.  Columns arrive as raw Arrow value buffers (the layout LingoDBTable keeps:
// date32 / int32 / fixed_size_binary(4) = 4 bytes, decimal128 = 16 bytes little-endian).  Every function returns the wall-clock milliseconds of
// ONE execution (scheduler start-up excluded — the reference keeps its workers alive between queries) and writes its result for the caller to
// compare with the oracle legs.
namespace {
using i128 = __int128;
uint64_t dbHash64Glue(int64_t v) { // Hash.cpp:25-28
   uint64_t m1 = 11400714819323198549ull * static_cast<uint64_t>(v);
   return m1 ^ __builtin_bswap64(m1);
}
void fold(uint64_t& acc, uint64_t piece) { acc = __builtin_bswap64(acc) ^ piece; } // Hash.cpp:30-32

struct RawColumn { // one Arrow column as the ArrayView TableChunk hands to Restrictions (LingoDBTable.cpp:200-225)
   runtime::ArrayView view;
   const void* bufs[3];
   void set(const void* values, int64_t n) {
      view.length = n;
      view.nullCount = 0;
      view.offset = 0;
      view.nBuffers = 2;
      view.nChildren = 0;
      view.children = nullptr;
      bufs[0] = runtime::ArrayView::validData.data();
      bufs[1] = values;
      bufs[2] = nullptr;
      view.buffers = bufs;
   }
};
runtime::FilterDescription filterInt(const char* col, runtime::FilterOp op, int64_t v) {
   runtime::FilterDescription d{};
   d.columnName = col;
   d.columnId = 0;
   d.op = op;
   d.value = v;
   return d;
}
runtime::FilterDescription filterStr(const char* col, runtime::FilterOp op, const char* v) {
   runtime::FilterDescription d{};
   d.columnName = col;
   d.columnId = 0;
   d.op = op;
   d.value = std::string(v);
   return d;
}
double msSince(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// Q1's aggregation entry: keys + five 128-bit sums + count (the reference widens SUM(decimal(12,2)) to decimal(38,s): i128 arithmetic)
struct Q1Content {
   int32_t rf, ls;
   i128 qty, price, discPrice, charge, disc;
   int64_t count;
};
bool q1Eq(uint8_t* a, uint8_t* b) { return reinterpret_cast<Q1Content*>(a)->rf == reinterpret_cast<Q1Content*>(b)->rf && reinterpret_cast<Q1Content*>(a)->ls == reinterpret_cast<Q1Content*>(b)->ls; }
void q1Combine(uint8_t* d, uint8_t* s) {
   auto* x = reinterpret_cast<Q1Content*>(d);
   auto* y = reinterpret_cast<Q1Content*>(s);
   x->qty += y->qty, x->price += y->price, x->discPrice += y->discPrice, x->charge += y->charge, x->disc += y->disc, x->count += y->count;
}
struct Q3Content {
   int32_t orderkey, orderdate, shippriority;
   i128 revenue;
};
bool q3Eq(uint8_t* a, uint8_t* b) {
   auto* x = reinterpret_cast<Q3Content*>(a);
   auto* y = reinterpret_cast<Q3Content*>(b);
   return x->orderkey == y->orderkey && x->orderdate == y->orderdate && x->shippriority == y->shippriority;
}
void q3Combine(uint8_t* d, uint8_t* s) { reinterpret_cast<Q3Content*>(d)->revenue += reinterpret_cast<Q3Content*>(s)->revenue; }
struct BuildEntry { // MultiMapAsHashIndexedView layout (SpecializeSubOpPass.cpp:70-84): {next, hash, key, payload…}
   BuildEntry* next;
   uint64_t hash;
   int32_t key, a, b, pad;
};
struct Scheduler { // one scheduler + execution context per baseline session (the reference keeps its workers alive between queries)
   std::unique_ptr<CtxScope> scope;
   int threads = 0;
} g_sched;
} // namespace

int32_t ref_baseline_begin(int32_t threads) {
   g_sched.scope = std::make_unique<CtxScope>(threads);
   g_sched.threads = threads;
   return 0;
}
void ref_baseline_end() { g_sched.scope.reset(); }

// Q1: scan(lineitem, l_shipdate <= 1998-09-02) → lookup_or_insert in the PreAggregationHashtableFragment + reduce → merge → scan of the groups.
// out: per group {rf, ls, count, then the five sums as (lo, hi) pairs} = 13 int64 words; returns ms, *n_groups = groups
double ref_q1(const int32_t* shipdate, const int32_t* returnflag, const int32_t* linestatus, const i128* quantity, const i128* price, const i128* discount, const i128* tax, int64_t n,
              int64_t* out, int64_t cap, int64_t* n_groups) {
   using Fragment = runtime::PreAggregationHashtableFragment;
   if (!g_sched.scope) return -1;
   auto schema = arrow::schema({arrow::field("l_shipdate", arrow::date32())});
   std::unique_ptr<runtime::Restrictions> restrictions;
   try {
      restrictions = runtime::Restrictions::create({filterStr("l_shipdate", runtime::FilterOp::LTE, "1998-09-02")}, *schema);
   } catch (std::exception&) { return -2; }
   RawColumn dateCol;
   dateCol.set(shipdate, n);
   const auto t0 = std::chrono::steady_clock::now();
   const size_t typeSize = sizeof(Fragment::Entry) + sizeof(Q1Content);
   auto* tl = runtime::ThreadLocal::create([](uint8_t* arg) -> uint8_t* { return reinterpret_cast<uint8_t*>(Fragment::create(*reinterpret_cast<size_t*>(arg), false)); },
                                           reinterpret_cast<uint8_t*>(const_cast<size_t*>(&typeSize)));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n, 20000, [&](size_t b, size_t e, size_t) {
      uint16_t sv1[65536], sv2[65536];
      auto [len, sel] = restrictions->applyFilters(b, e - b, sv1, sv2, [&](size_t) { return &dateCol.view; });
      auto* frag = reinterpret_cast<Fragment*>(tl->getLocal());
      for (size_t k = 0; k < len; k++) {
         const size_t i = b + sel[k];
         uint64_t h = 0; // db.hash over (l_returnflag, l_linestatus): HashLowering folds the pieces (LowerToStd.cpp:1065-1152)
         fold(h, dbHash64Glue(returnflag[i]));
         fold(h, dbHash64Glue(linestatus[i]));
         Fragment::Entry* en = frag->ht[(h >> 6) & (Fragment::hashtableSize - 1)]; // LookupPreAggrHtFragment (SubOpToControlFlow.cpp:3065-3157)
         if (!(en && en->hashValue == h && reinterpret_cast<Q1Content*>(en->content)->rf == returnflag[i] && reinterpret_cast<Q1Content*>(en->content)->ls == linestatus[i])) {
            en = frag->insert(h);
            auto* c = reinterpret_cast<Q1Content*>(en->content);
            *c = Q1Content{returnflag[i], linestatus[i], 0, 0, 0, 0, 0, 0};
         }
         auto* c = reinterpret_cast<Q1Content*>(en->content);
         const i128 dp = price[i] * (100 - discount[i]); // decimal(12,2) x decimal(13,2) → scale 4
         c->qty += quantity[i];
         c->price += price[i];
         c->discPrice += dp;
         c->charge += dp * (100 + tax[i]); // scale 6
         c->disc += discount[i];
         c->count += 1;
      }
   }));
   auto* merged = runtime::PreAggregationHashtable::merge(tl, q1Eq, q1Combine);
   struct Out {
      int64_t* out;
      int64_t n, cap;
   } o{out, 0, cap};
   auto* it = merged->createIterator();
   runtime::BufferIterator::iterate(
      it, false,
      [](runtime::Buffer buf, void* arg) {
         auto* st = reinterpret_cast<Out*>(arg);
         auto** entries = reinterpret_cast<Fragment::Entry**>(buf.ptr);
         const size_t cnt = buf.numElements / sizeof(Fragment::Entry*);
         for (size_t k = 0; k < cnt; k++) {
            auto* c = reinterpret_cast<Q1Content*>(entries[k]->content);
            if (st->n < st->cap) {
               int64_t* r = st->out + 13 * st->n;
               r[0] = c->rf, r[1] = c->ls, r[2] = c->count;
               const i128 sums[5] = {c->qty, c->price, c->discPrice, c->charge, c->disc};
               for (int a = 0; a < 5; a++) r[3 + 2 * a] = (int64_t) sums[a], r[4 + 2 * a] = (int64_t) (sums[a] >> 64);
            }
            st->n++;
         }
      },
      &o);
   const double ms = msSince(t0);
   *n_groups = o.n;
   return ms;
}

// Q6: five restrictions on three columns → SUM(l_extendedprice * l_discount) in a SimpleState per worker (key-less reduce, SimpleState.cpp),
// the per-worker states combined at the end.  out_lohi = the 128-bit sum (scale 4), *n_pass = rows passing
double ref_q6(const int32_t* shipdate, const i128* discount, const i128* quantity, const i128* price, int64_t n, int64_t out_lohi[2], int64_t* n_pass) {
   if (!g_sched.scope) return -1;
   auto schema = arrow::schema({arrow::field("l_shipdate", arrow::date32()), arrow::field("l_discount", arrow::decimal128(12, 2)), arrow::field("l_quantity", arrow::decimal128(12, 2))});
   std::unique_ptr<runtime::Restrictions> restrictions;
   try {
      restrictions = runtime::Restrictions::create({filterStr("l_shipdate", runtime::FilterOp::GTE, "1994-01-01"), filterStr("l_shipdate", runtime::FilterOp::LT, "1995-01-01"),
                                                    filterStr("l_discount", runtime::FilterOp::GTE, "0.05"), filterStr("l_discount", runtime::FilterOp::LTE, "0.07"),
                                                    filterStr("l_quantity", runtime::FilterOp::LT, "24")},
                                                   *schema);
   } catch (std::exception&) { return -2; }
   RawColumn cols[3];
   cols[0].set(shipdate, n);
   cols[1].set(discount, n);
   cols[2].set(quantity, n);
   const auto t0 = std::chrono::steady_clock::now();
   struct alignas(64) State {
      i128 sum = 0;
      int64_t rows = 0;
   };
   std::vector<State> states((size_t) scheduler::getNumWorkers());
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n, 20000, [&](size_t b, size_t e, size_t worker) {
      uint16_t sv1[65536], sv2[65536];
      auto [len, sel] = restrictions->applyFilters(b, e - b, sv1, sv2, [&](size_t colId) { return &cols[colId].view; });
      State& st = states[worker];
      for (size_t k = 0; k < len; k++) {
         const size_t i = b + sel[k];
         st.sum += price[i] * discount[i];
      }
      st.rows += (int64_t) len;
   }));
   i128 total = 0;
   int64_t rows = 0;
   for (auto& s : states) total += s.sum, rows += s.rows;
   const double ms = msSince(t0);
   out_lohi[0] = (int64_t) total;
   out_lohi[1] = (int64_t) (total >> 64);
   *n_pass = rows;
   return ms;
}

// Q3: customer (c_mktsegment = 'BUILDING') → GrowingBuffer → HashIndexedView; orders (o_orderdate < 1995-03-15) probe it, survivors →
// GrowingBuffer → HashIndexedView; lineitem (l_shipdate > 1995-03-15) probes that; group by (l_orderkey, o_orderdate, o_shippriority) in the
// PreAggregationHashtable; the groups come back unsorted (the caller orders the few thousand it compares).  c_mktsegment arrives as 4-byte codes
// of its first four characters (the restriction on the utf8 column is the only string work of the query and is evaluated by the REAL Restrictions
// over a fixed_size_binary(4) rendering 'BUIL' — the segment names differ in their first letter).
// out: per group {orderkey, orderdate, shippriority, revenue lo, revenue hi}
double ref_q3(const int32_t* c_custkey, const int32_t* c_segment4, int64_t n_c, const int32_t* o_orderkey, const int32_t* o_custkey, const int32_t* o_orderdate,
              const int32_t* o_shippriority, int64_t n_o, const int32_t* l_orderkey, const i128* l_price, const i128* l_discount, const int32_t* l_shipdate, int64_t n_l,
              int64_t* out, int64_t cap, int64_t* n_groups) {
   using Fragment = runtime::PreAggregationHashtableFragment;
   if (!g_sched.scope) return -1;
   std::unique_ptr<runtime::Restrictions> rOrders, rLine;
   try {
      rOrders = runtime::Restrictions::create({filterStr("o_orderdate", runtime::FilterOp::LT, "1995-03-15")}, *arrow::schema({arrow::field("o_orderdate", arrow::date32())}));
      rLine = runtime::Restrictions::create({filterStr("l_shipdate", runtime::FilterOp::GT, "1995-03-15")}, *arrow::schema({arrow::field("l_shipdate", arrow::date32())}));
   } catch (std::exception&) { return -2; }
   RawColumn oDate, lDate;
   oDate.set(o_orderdate, n_o);
   lDate.set(l_shipdate, n_l);
   const int32_t building = (int32_t) ('B' | ('U' << 8) | ('I' << 16) | ('L' << 24));
   const auto t0 = std::chrono::steady_clock::now();
   // pipeline 1: customer → build side
   auto* tlC = runtime::GrowingBuffer::createThreadLocal(sizeof(BuildEntry));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n_c, 20000, [&](size_t b, size_t e, size_t) {
      auto* buf = reinterpret_cast<runtime::GrowingBuffer*>(tlC->getLocal());
      for (size_t i = b; i < e; i++) {
         if (c_segment4[i] != building) continue;
         auto* en = reinterpret_cast<BuildEntry*>(buf->insert());
         *en = BuildEntry{nullptr, dbHash64Glue(c_custkey[i]), c_custkey[i], 0, 0, 0};
      }
   }));
   auto* viewC = reinterpret_cast<ViewLayout*>(runtime::HashIndexedView::build(runtime::GrowingBuffer::merge(tlC)));
   // pipeline 2: orders ⋈ customer → build side keyed by o_orderkey
   auto* tlO = runtime::GrowingBuffer::createThreadLocal(sizeof(BuildEntry));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n_o, 20000, [&](size_t b, size_t e, size_t) {
      uint16_t sv1[65536], sv2[65536];
      auto [len, sel] = rOrders->applyFilters(b, e - b, sv1, sv2, [&](size_t) { return &oDate.view; });
      auto* buf = reinterpret_cast<runtime::GrowingBuffer*>(tlO->getLocal());
      for (size_t k = 0; k < len; k++) {
         const size_t i = b + sel[k];
         const uint64_t h = dbHash64Glue(o_custkey[i]);
         JoinEntry* slot = viewC->ht[h & viewC->mask]; // LookupHashIndexedViewLowering (SubOpToControlFlow.cpp:2558-2586)
         auto* cur = reinterpret_cast<BuildEntry*>(runtime::matchesTag(slot, h) ? runtime::untag(slot) : nullptr);
         for (; cur; cur = cur->next) {
            if (cur->key == o_custkey[i]) {
               auto* en = reinterpret_cast<BuildEntry*>(buf->insert());
               *en = BuildEntry{nullptr, dbHash64Glue(o_orderkey[i]), o_orderkey[i], o_orderdate[i], o_shippriority[i], 0};
               break; // c_custkey is the primary key
            }
         }
      }
   }));
   auto* viewO = reinterpret_cast<ViewLayout*>(runtime::HashIndexedView::build(runtime::GrowingBuffer::merge(tlO)));
   // pipeline 3: lineitem ⋈ orders → aggregation
   const size_t typeSize = sizeof(Fragment::Entry) + sizeof(Q3Content);
   auto* tl = runtime::ThreadLocal::create([](uint8_t* arg) -> uint8_t* { return reinterpret_cast<uint8_t*>(Fragment::create(*reinterpret_cast<size_t*>(arg), false)); },
                                           reinterpret_cast<uint8_t*>(const_cast<size_t*>(&typeSize)));
   scheduler::awaitEntryTask(std::make_unique<RangeTask>((size_t) n_l, 20000, [&](size_t b, size_t e, size_t) {
      uint16_t sv1[65536], sv2[65536];
      auto [len, sel] = rLine->applyFilters(b, e - b, sv1, sv2, [&](size_t) { return &lDate.view; });
      auto* frag = reinterpret_cast<Fragment*>(tl->getLocal());
      for (size_t k = 0; k < len; k++) {
         const size_t i = b + sel[k];
         const uint64_t hk = dbHash64Glue(l_orderkey[i]);
         JoinEntry* slot = viewO->ht[hk & viewO->mask];
         auto* cur = reinterpret_cast<BuildEntry*>(runtime::matchesTag(slot, hk) ? runtime::untag(slot) : nullptr);
         for (; cur; cur = cur->next) {
            if (cur->key != l_orderkey[i]) continue;
            uint64_t h = 0;
            fold(h, hk);
            fold(h, dbHash64Glue(cur->a));
            fold(h, dbHash64Glue(cur->b));
            Fragment::Entry* en = frag->ht[(h >> 6) & (Fragment::hashtableSize - 1)];
            auto* c = en ? reinterpret_cast<Q3Content*>(en->content) : nullptr;
            if (!(en && en->hashValue == h && c->orderkey == l_orderkey[i] && c->orderdate == cur->a && c->shippriority == cur->b)) {
               en = frag->insert(h);
               c = reinterpret_cast<Q3Content*>(en->content);
               *c = Q3Content{l_orderkey[i], cur->a, cur->b, 0};
            }
            c->revenue += l_price[i] * (100 - l_discount[i]);
            break; // o_orderkey is the primary key
         }
      }
   }));
   auto* merged = runtime::PreAggregationHashtable::merge(tl, q3Eq, q3Combine);
   struct Out {
      int64_t* out;
      int64_t n, cap;
   } o{out, 0, cap};
   auto* it = merged->createIterator();
   runtime::BufferIterator::iterate(
      it, false,
      [](runtime::Buffer buf, void* arg) {
         auto* st = reinterpret_cast<Out*>(arg);
         auto** entries = reinterpret_cast<Fragment::Entry**>(buf.ptr);
         const size_t cnt = buf.numElements / sizeof(Fragment::Entry*);
         for (size_t k = 0; k < cnt; k++) {
            auto* c = reinterpret_cast<Q3Content*>(entries[k]->content);
            if (st->n < st->cap) {
               int64_t* r = st->out + 5 * st->n;
               r[0] = c->orderkey, r[1] = c->orderdate, r[2] = c->shippriority, r[3] = (int64_t) c->revenue, r[4] = (int64_t) (c->revenue >> 64);
            }
            st->n++;
         }
      },
      &o);
   const double ms = msSince(t0);
   *n_groups = o.n;
   return ms;
}
} // extern "C"
