// ldb_scan.hip — columnar scan + pushed-down predicate evaluation → selection (row-id vector).
// Replaces (reference): ScanBatchesTask::unitRun (src/runtime/storage/LingoDBTable.cpp:382-407)
// and Restrictions::applyFilters with its Filter impls (src/runtime/storage/Restrictions.cpp:67-390).
//
// MI355X design: the CPU ping-pongs uint16 selection vectors per 20 000-row morsel, one pass per
// predicate.  Here one pass evaluates the whole conjunction per row (later conjuncts only load
// their column for lanes still alive), a wave turns its 64 verdicts into one 64-bit ballot word,
// and the selection is produced from that bitmap (1 bit/row of extra traffic) in ascending row
// order: k_scan_bitmap → block-count scan → k_scan_expand (mbcnt rank → coalesced id writes).
#include "ldb_internal.h"
#include "ldb_device.h"
#include <memory>

#include "ldb_scan_kernel.h"
#include "ldb_jit.h"

// generic ahead-of-time kernels (descriptor read from memory)
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_bitmap(const DScan* __restrict__ d, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ block_counts) {
   __shared__ uint32_t s_cnt[SCAN_BLOCK / LDB_WAVE];
   scan_bitmap_body(*d, d, bitmap, block_counts, s_cnt);
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_count(const DScan* __restrict__ d, unsigned long long* __restrict__ total) { scan_count_body(*d, d, total); }

// run-time specialised variants of the two kernels above (hiprtc; ldb_jit.hip)
static const char* SCAN_SPEC_SRC =
   "extern \"C\" __global__ __launch_bounds__(SCAN_BLOCK) void k_scan_bitmap_spec(const DScan* __restrict__ d, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts) {\n"
   "   __shared__ uint32_t s_cnt[SCAN_BLOCK / LDB_WAVE];\n"
   "   scan_bitmap_body(LDB_META, d, bitmap, counts, s_cnt);\n"
   "}\n"
   "extern \"C\" __global__ __launch_bounds__(SCAN_BLOCK) void k_scan_count_spec(const DScan* __restrict__ d, unsigned long long* __restrict__ total) { scan_count_body(LDB_META, d, total); }\n";
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_bitmap_dnf(const DScanDnf* __restrict__ d, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ block_counts) {
   __shared__ uint32_t s_cnt[SCAN_BLOCK / LDB_WAVE];
   scan_bitmap_dnf_body(*d, d, bitmap, block_counts, s_cnt);
}
static const char* DNF_SPEC_SRC =
   "extern \"C\" __global__ __launch_bounds__(SCAN_BLOCK) void k_scan_bitmap_dnf_spec(const DScanDnf* __restrict__ d, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ counts) {\n"
   "   __shared__ uint32_t s_cnt[SCAN_BLOCK / LDB_WAVE];\n"
   "   scan_bitmap_dnf_body(LDB_META, d, bitmap, counts, s_cnt);\n"
   "}\n";
static void scan_meta(const DScan* h, DScan* m) {
   memcpy(m, h, sizeof(DScan));
   m->n_rows = 0;
   for (int p = 0; p < LDB_MAX_PREDS; p++) ldb_jit_strip_pred(m->preds[p]);
}

bool ldb_scan_jit_check(std::string* log) {
   DScan m;
   memset(&m, 0, sizeof(m));
   m.n_preds = 2;
   m.preds[0].col.type = LDB_T_DATE32;
   m.preds[0].col.width = 4;
   m.preds[0].op = LDB_F_GTE;
   m.preds[1].col.type = LDB_T_UTF8;
   m.preds[1].col.offsets = 1;
   m.preds[1].op = LDB_F_EQ;
   m.preds[1].rhs_kind = LDB_RHS_STRING;
   m.preds[1].str_len = 8;
   memcpy(m.preds[1].str, "BUILDING", 8);
   // a "simple" LIKE pattern: the position-parallel matcher with the segments as compile-time constants
   m.n_preds = 3;
   m.preds[2].col.type = LDB_T_UTF8;
   m.preds[2].col.offsets = 1;
   m.preds[2].op = LDB_F_NOT_LIKE;
   m.preds[2].rhs_kind = LDB_RHS_STRING;
   m.preds[2].str_len = 18;
   memcpy(m.preds[2].str, "%special%requests%", 18);
   ldb_like_plan(&m.preds[2]);
   if (m.preds[2].n_in != 2) {
      if (log) *log = "scan jit check: the LIKE pattern was not planned as two segments";
      return false;
   }
   if (!ldb_jit_compile_only("ldb_scan_kernel.h", "DScan", SCAN_SPEC_SRC, &m, sizeof(m), log)) return false;
   // a two-clause disjunction over the same conjuncts
   auto dn = std::make_unique<DScanDnf>();
   memset(dn.get(), 0, sizeof(DScanDnf));
   dn->n_clauses = 2;
   dn->preds[0] = m.preds[0];
   dn->preds[1] = m.preds[1];
   dn->preds[2] = m.preds[0];
   dn->preds[2].op = LDB_F_LT;
   dn->clause_end[0] = 2;
   dn->clause_end[1] = 3;
   return ldb_jit_compile_only("ldb_scan_kernel.h", "DScanDnf", DNF_SPEC_SRC, dn.get(), sizeof(DScanDnf), log);
}

// Expand the bitmap into ascending row ids.  Each wave walks its words; the lane whose bit is
// set writes its row id at block_offset + (#set bits in earlier words) + rank within the word.
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_expand(const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ block_offsets,
                                                            uint32_t* __restrict__ out_rows, uint64_t n_rows, uint64_t cap, uint32_t parts) {
   __shared__ uint32_t s_pop[SCAN_WORDS_PER_BLOCK];
   const uint64_t word0 = (uint64_t) blockIdx.x * SCAN_WORDS_PER_BLOCK;
   const uint64_t n_words = (n_rows + 63) / 64;
   uint64_t my = word0 + threadIdx.x < n_words ? bitmap[word0 + threadIdx.x] : 0;
   s_pop[threadIdx.x] = (uint32_t) __popcll(my);
   __syncthreads();
   // exclusive scan over the 256 word popcounts (Hillis-Steele in LDS)
   uint32_t own = s_pop[threadIdx.x];
   for (int off = 1; off < SCAN_WORDS_PER_BLOCK; off <<= 1) {
      uint32_t t = threadIdx.x >= (unsigned) off ? s_pop[threadIdx.x - off] : 0;
      __syncthreads();
      s_pop[threadIdx.x] += t;
      __syncthreads();
   }
   uint32_t excl = s_pop[threadIdx.x] - own;
   const uint32_t block_total = s_pop[SCAN_WORDS_PER_BLOCK - 1];
   __syncthreads();
   s_pop[threadIdx.x] = excl;
   __syncthreads();
   const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const uint32_t base = block_offsets[blockIdx.x * parts]; // (the scan kernels count per part of a zone: the first part's offset is the zone's)
   // entries behind the real count (cap is larger only when it came from a replayed count that no longer holds) must
   // still be row numbers: the consumer is already queued
   if (blockIdx.x == gridDim.x - 1)
      for (uint64_t i = (uint64_t) base + block_total + threadIdx.x; i < cap; i += SCAN_BLOCK) out_rows[i] = 0;
   for (uint32_t w = wave; w < SCAN_WORDS_PER_BLOCK; w += SCAN_BLOCK / LDB_WAVE) {
      if (word0 + w >= n_words) break;
      uint64_t m = bitmap[word0 + w]; // wave-uniform
      const uint64_t at = (uint64_t) base + s_pop[w] + d_rank_in(m);
      if (((m >> lane) & 1) && at < cap) out_rows[at] = (uint32_t) ((word0 + w) * 64 + lane); // (cap = the host's count; see ldb_readback)
   }
}

// compose: new_rowids[j] = old_rowids[sel[j]] (sides that already had row ids)
__global__ void k_compose(const uint32_t* __restrict__ old_rows, const uint32_t* __restrict__ sel, uint32_t* __restrict__ out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = old_rows[sel[i]];
}

static int32_t build_scan_desc(ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds, DScan* h) {
   if (n_preds < 0 || n_preds > LDB_MAX_PREDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "scan: %d predicates (max %d)", n_preds, LDB_MAX_PREDS);
   memset(h, 0, sizeof(*h));
   h->n_rows = (uint64_t) in->n_rows;
   h->n_preds = n_preds;
   for (int32_t p = 0; p < n_preds; p++) LDB_TRY(ldb_make_dpred(in, &preds[p], &h->preds[p]));
   ldb_order_preds(h->preds, n_preds); // cheap conjuncts first, same-column neighbours marked
   return LDB_OK;
}

// Restrict relation `in` to the logical rows listed in `sel` (device, n_sel entries, ascending).
// Takes ownership of `sel`.
int32_t ldb_rel_select(ldb_ctx* ctx, ldb_rel* in, uint32_t* sel, int64_t n_sel, ldb_rel** out) {
   ldb_rel* r = ldb_rel_new(ctx);
   r->n_rows = n_sel;
   bool sel_used = false;
   std::vector<LdbComposeJob> jobs; // all sides that need a vector of their own: one launch
   for (auto& s : in->sides) {
      ldb_rel_side ns;
      ns.table = s.table;
      ns.may_null = s.may_null;
      if (!s.rowids && !sel_used) {
         ns.rowids = sel;
         ns.owned = true;
         sel_used = true;
      } else { // (a second identity side over the same logical rows gets a copy of the selection: ids == NULL)
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, sizeof(uint32_t) * (size_t) (n_sel ? n_sel : 1)));
         ns.owned = true;
         jobs.push_back({(const uint32_t*) s.rowids, ns.rowids, 0});
      }
      r->sides.push_back(ns);
   }
   LDB_TRY(ldb_compose_rowids(ctx, sel, nullptr, jobs.data(), (int) jobs.size(), (uint64_t) n_sel));
   if (!sel_used) ldb_dev_free(ctx, sel);
   *out = r;
   return LDB_OK;
}

// bitmap → ascending row numbers (device, owned by caller).  `launch(bitmap, block_counts)` runs the
// kernel that evaluates the predicate: one ballot word per 64 rows + passing rows per 16384-row block.
template <typename LAUNCH>
static int32_t scan_run_with(ldb_ctx* ctx, int64_t n, LAUNCH launch, uint32_t** sel_out, uint64_t* total_out) {
   const int64_t n_words = (n + 63) / 64;
   const int64_t n_blocks = (n_words + SCAN_WORDS_PER_BLOCK - 1) / SCAN_WORDS_PER_BLOCK;
   uint32_t* sel;
   if (n == 0) {
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &sel, 16));
      *sel_out = sel;
      *total_out = 0;
      return LDB_OK;
   }
   // a short input is cut finer than one workgroup per 16 384 rows: 2 … 16 workgroups share a zone (the kernels read the split from gridDim.y)
   unsigned parts = 1;
   if (ldb_option("scan_split", 1) != 0)
      while (parts < 16 && n_blocks * parts < 2048) parts *= 2;
   uint64_t* bitmap;
   uint32_t *counts, *offsets;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &bitmap, sizeof(uint64_t) * (size_t) n_words));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &counts, sizeof(uint32_t) * (size_t) n_blocks * parts));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &offsets, sizeof(uint32_t) * (size_t) n_blocks * parts));
   LDB_TRY(launch(bitmap, counts, (unsigned) n_blocks, parts));
   LDB_HIP(hipGetLastError());
   uint64_t* d_total;
   LDB_TRY(ldb_counters(ctx, 1, &d_total));
   LDB_TRY(ldb_exclusive_scan_u32(ctx, counts, offsets, n_blocks * parts, d_total));
   uint64_t total = 0;
   // (the read-back's identity includes the scanned row count: a scan that ran in the recorded execution and is skipped now — a dictionary
   // predicate whose code set is cached by then — must not hand ITS count to the next scan of the plan; a different identity makes the replay
   // see that the execution took another path, check what it replayed and record from there, instead of replaying a wrong count and losing
   // the whole execution at the final comparison: 7 of the 22 queries paid that once during warm-up)
   LDB_TRY(ldb_read_u64_at(ctx, d_total, &total, ldb_site_derived(LDB_SITE, (uint32_t) n), 0));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &sel, sizeof(uint32_t) * (size_t) (total ? total : 1)));
   if (total) hipLaunchKernelGGL(k_scan_expand, dim3((unsigned) n_blocks), dim3(SCAN_BLOCK), 0, ctx->stream, bitmap, offsets, sel, (uint64_t) n, total, (uint32_t) parts);
   LDB_HIP(hipGetLastError());
   ldb_dev_free(ctx, bitmap);
   ldb_dev_free(ctx, counts);
   ldb_dev_free(ctx, offsets);
   *sel_out = sel;
   *total_out = total;
   return LDB_OK;
}

// run the conjunction over the dense base rows of `in` → ascending row numbers (device, owned by caller)
static int32_t scan_run(ldb_ctx* ctx, ldb_rel* in, const DScan& h, uint32_t** sel_out, uint64_t* total_out) {
   const int64_t n = in->n_rows;
   LdbDesc<DScan> d_desc(ctx);
   if (n) LDB_TRY(d_desc.upload(&h, sizeof(h)));
   DScan* d = d_desc.p;
   const int32_t st = scan_run_with(
      ctx, n,
      [&](uint64_t* bitmap, uint32_t* counts, unsigned n_blocks, unsigned parts) -> int32_t {
         hipFunction_t spec = nullptr;
         // a LIKE is specialised from far fewer rows on (round 6): with its pattern as compile-time constants the matcher is ~50 x the generic one per
         // row — Q16's '%Customer%Complaints%' over 1 M supplier comments took 1.13 ms generic, the 150 M comments of Q13 take 3.3 ms specialised
         bool has_like = false;
         for (int p = 0; p < h.n_preds; p++) has_like = has_like || h.preds[p].op == LDB_F_LIKE || h.preds[p].op == LDB_F_NOT_LIKE;
         if (ldb_jit_wanted(n) || (has_like && ldb_option("jit", 1) != 0 && n >= ldb_option("jit_min_rows_like", 65536))) {
            DScan meta;
            scan_meta(&h, &meta);
            meta.n_rows = parts > 1 ? 1 : 0; // (which of the two loop forms this specialisation is compiled for: ldb_scan_kernel.h)
            std::string why;
            spec = ldb_jit_kernel(ctx->device, "ldb_scan_kernel.h", "DScan", SCAN_SPEC_SRC, "k_scan_bitmap_spec", &meta, sizeof(meta), &why);
         }
         LdbProf prof_(ctx, "k_scan_bitmap");
         if (spec) {
            void* params[] = {(void*) &d, (void*) &bitmap, (void*) &counts};
            LDB_HIP(hipModuleLaunchKernel(spec, n_blocks, parts, 1, SCAN_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
         } else {
            hipLaunchKernelGGL(k_scan_bitmap, dim3(n_blocks, parts), dim3(SCAN_BLOCK), 0, ctx->stream, d, bitmap, counts);
         }
         return LDB_OK;
      },
      sel_out, total_out);
   d_desc.release();
   return st;
}

// ---------------------------------------------------------------- disjunctive normal form
// OR of up to DNF_MAX_CLAUSES conjunctions (TPC-H Q19's three brand/container/quantity/size
// alternatives).  The reference keeps such a predicate as generated residual code — db.or over
// db.and trees after the optimiser has pulled the common conjuncts out — and
// evaluates it per tuple; here one pass evaluates clause after clause per row (a row that already
// passed skips the remaining clauses) into the same ballot bitmap the conjunctive scan produces.
extern "C" int32_t ldb_gpu_scan_filter_dnf(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, const int32_t* clause_sizes, int32_t n_clauses, ldb_rel** out) {
   if (!ctx || !in || !out || !preds || !clause_sizes) LDB_FAIL(LDB_ERR_INVALID, "scan_filter_dnf: NULL argument");
   if (n_clauses < 1 || n_clauses > DNF_MAX_CLAUSES) LDB_FAIL(LDB_ERR_UNSUPPORTED, "scan_filter_dnf: %d clauses (max %d)", n_clauses, DNF_MAX_CLAUSES);
   LDB_TRY(ldb_rel_force(ctx, in));
   auto hp = std::make_unique<DScanDnf>();
   memset(hp.get(), 0, sizeof(DScanDnf));
   hp->n_rows = (uint64_t) in->n_rows;
   hp->n_clauses = n_clauses;
   int32_t total_preds = 0;
   for (int32_t c = 0; c < n_clauses; c++) {
      if (clause_sizes[c] == 0) LDB_FAIL(LDB_ERR_INVALID, "scan_filter_dnf: clause %d is empty", c);
      if (clause_sizes[c] < 0 || total_preds + clause_sizes[c] > DNF_MAX_PREDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "scan_filter_dnf: more than %d conjuncts in all", DNF_MAX_PREDS);
      for (int32_t p = 0; p < clause_sizes[c]; p++, total_preds++) LDB_TRY(ldb_make_dpred(in, &preds[total_preds], &hp->preds[total_preds]));
      hp->clause_end[c] = total_preds;
   }
   LdbDesc<DScanDnf> d_desc(ctx);
   if (in->n_rows) LDB_TRY(d_desc.upload(hp.get(), sizeof(DScanDnf)));
   DScanDnf* d = d_desc.p;
   uint32_t* sel;
   uint64_t total;
   const int32_t st = scan_run_with(
      ctx, in->n_rows,
      [&](uint64_t* bitmap, uint32_t* counts, unsigned n_blocks, unsigned parts) -> int32_t {
         hipFunction_t spec = nullptr;
         if (ldb_jit_wanted(in->n_rows)) { // the clause structure and every conjunct's type / operator / constant as compile-time constants
            auto meta = std::make_unique<DScanDnf>();
            memcpy(meta.get(), hp.get(), sizeof(DScanDnf));
            meta->n_rows = parts > 1 ? 1 : 0; // (the loop form, as in scan_run)
            for (int p = 0; p < DNF_MAX_PREDS; p++) ldb_jit_strip_pred(meta->preds[p]);
            std::string why;
            spec = ldb_jit_kernel(ctx->device, "ldb_scan_kernel.h", "DScanDnf", DNF_SPEC_SRC, "k_scan_bitmap_dnf_spec", meta.get(), sizeof(DScanDnf), &why);
         }
         LdbProf prof_(ctx, "k_scan_bitmap_dnf");
         if (spec) {
            void* params[] = {(void*) &d, (void*) &bitmap, (void*) &counts};
            LDB_HIP(hipModuleLaunchKernel(spec, n_blocks, parts, 1, SCAN_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
         } else {
            hipLaunchKernelGGL(k_scan_bitmap_dnf, dim3(n_blocks, parts), dim3(SCAN_BLOCK), 0, ctx->stream, (const DScanDnf*) d, bitmap, counts);
         }
         return LDB_OK;
      },
      &sel, &total);
   d_desc.release();
   LDB_TRY(st);
   return ldb_rel_select(ctx, in, sel, (int64_t) total, out);
}

// lazy filters: on by default for dense relations of >= LDB_LAZY_MIN_ROWS rows (default 1 M);
// LDB_LAZY_FILTER=0 materialises every filter immediately
static bool lazy_wanted(const ldb_rel* in) {
   if (ldb_option("lazy_filter", 1) == 0 || in->n_rows < ldb_option("lazy_min_rows", 1 << 20)) return false;
   for (auto& s : in->sides)
      if (s.rowids) return false;
   return true;
}

int32_t ldb_rel_force(ldb_ctx* ctx, ldb_rel* r) {
   if (!r || r->pending.empty()) return LDB_OK;
   DScan h;
   memset(&h, 0, sizeof(h));
   h.n_rows = (uint64_t) r->n_rows;
   h.n_preds = (int32_t) r->pending.size();
   for (size_t p = 0; p < r->pending.size(); p++) h.preds[p] = r->pending[p];
   ldb_order_preds(h.preds, h.n_preds);
   uint32_t* sel;
   uint64_t total;
   LDB_TRY(scan_run(ctx, r, h, &sel, &total));
   r->pending.clear();
   ldb_rel* m = nullptr;
   LDB_TRY(ldb_rel_select(ctx, r, sel, (int64_t) total, &m));
   r->n_rows = m->n_rows;
   r->sides.swap(m->sides); // r was dense: m's old sides (after the swap) own nothing
   ldb_gpu_rel_release(ctx, m);
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_scan_filter(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds, ldb_rel** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "scan_filter: NULL argument");
   if (n_preds < 0 || n_preds > LDB_MAX_PREDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "scan: %d predicates (max %d)", n_preds, LDB_MAX_PREDS);
   if (!in->pending.empty() && in->pending.size() + (size_t) n_preds > LDB_MAX_PREDS) LDB_TRY(ldb_rel_force(ctx, in));
   // LIKE conjuncts are evaluated by the scan kernel, whose waves stage their 64 strings in LDS and
   // match by position (d_like_simple_wave); fused into a consumer they would run the row-wise
   // matcher (Q13's o_comment filter inside the group-by kernel: 84 ms instead of 4 + 10)
   bool has_like = false;
   for (int32_t p = 0; p < n_preds; p++) has_like = has_like || preds[p].op == LDB_F_LIKE || preds[p].op == LDB_F_NOT_LIKE;
   if (has_like && !in->pending.empty()) LDB_TRY(ldb_rel_force(ctx, in));
   if (!has_like && (!in->pending.empty() || lazy_wanted(in))) {
      // stay lazy: the conjuncts (compiled against the dense sides) travel with the relation
      std::vector<DPred> all = in->pending;
      for (int32_t p = 0; p < n_preds; p++) {
         DPred dp;
         LDB_TRY(ldb_make_dpred(in, &preds[p], &dp));
         all.push_back(dp);
      }
      ldb_rel* r = ldb_rel_new(ctx);
      r->n_rows = in->n_rows;
      for (auto& s : in->sides) r->sides.push_back(ldb_rel_side{s.table, nullptr, false});
      r->pending.swap(all);
      *out = r;
      return LDB_OK;
   }
   DScan h;
   LDB_TRY(build_scan_desc(in, preds, n_preds, &h));
   uint32_t* sel;
   uint64_t total;
   LDB_TRY(scan_run(ctx, in, h, &sel, &total));
   return ldb_rel_select(ctx, in, sel, (int64_t) total, out);
}

extern "C" int32_t ldb_gpu_scan_count(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds, int64_t* count) {
   if (!ctx || !in || !count) LDB_FAIL(LDB_ERR_INVALID, "scan_count: NULL argument");
   if (in->pending.size() + (size_t) (n_preds > 0 ? n_preds : 0) > LDB_MAX_PREDS) LDB_TRY(ldb_rel_force(ctx, in));
   DScan h;
   LDB_TRY(build_scan_desc(in, preds, n_preds, &h));
   for (auto& dp : in->pending) h.preds[h.n_preds++] = dp; // a lazy input's own conjuncts, fused
   ldb_order_preds(h.preds, h.n_preds);
   LdbDesc<DScan> d_desc(ctx);
   LDB_TRY(d_desc.upload(&h, sizeof(h)));
   DScan* d = d_desc.p;
   uint64_t* d_cnt; // (a zeroed arena word: no clear of its own)
   LDB_TRY(ldb_counters(ctx, 1, &d_cnt));
   if (in->n_rows) {
      hipFunction_t spec = nullptr;
      if (ldb_jit_wanted(in->n_rows)) {
         DScan meta;
         scan_meta(&h, &meta);
         std::string why;
         spec = ldb_jit_kernel(ctx->device, "ldb_scan_kernel.h", "DScan", SCAN_SPEC_SRC, "k_scan_count_spec", &meta, sizeof(meta), &why);
      }
      unsigned long long* total = (unsigned long long*) d_cnt;
      const int grid = ldb_grid_for(ctx, in->n_rows, SCAN_BLOCK, 8);
      LdbProf prof_(ctx, "k_scan_count");
      if (spec) {
         void* params[] = {(void*) &d, (void*) &total};
         LDB_HIP(hipModuleLaunchKernel(spec, (unsigned) grid, 1, 1, SCAN_BLOCK, 1, 1, 0, ctx->stream, params, nullptr));
      } else {
         hipLaunchKernelGGL(k_scan_count, dim3(grid), dim3(SCAN_BLOCK), 0, ctx->stream, d, total);
      }
   }
   LDB_HIP(hipGetLastError());
   uint64_t total = 0;
   LDB_TRY(ldb_read_u64(ctx, d_cnt, &total));
   d_desc.release();
   *count = (int64_t) total;
   return LDB_OK;
}
