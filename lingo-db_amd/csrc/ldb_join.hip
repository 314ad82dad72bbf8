// ldb_join.hip — hash-join build and probe: host side, AOT kernels, run-time specialisation hook.
// Device code: ldb_join_kernel.h.
// Replaces (reference): build-side materialisation + HashIndexedView::build
// (src/runtime/GrowingBuffer.cpp:44, src/runtime/LazyJoinHashtable.cpp:12-34) and the generated
// probe (LookupHashIndexedViewLowering / ScanListLowering,
// src/compiler/Conversion/SubOpToControlFlow/SubOpToControlFlow.cpp:2558-2586, 2254-2313).
//
// MI355X design: no row store, no pointer chains, no bloom-tagged pointers.  The table is a flat
// open-addressing array of 64-bit words claimed with ONE global CAS each:
//   KEY32 mode (one integer key of <= 32 bit — every TPC-H join):  word = key32 << 32 | (row + 1)
//            → a probe needs exactly one random 8-byte access per visited slot, no key re-check
//            (the CPU path drops the hash compare in the same case, SpecializeSubOpPass.cpp:110-118);
//   TAG mode (anything else, incl. strings / composite keys):      word = hash[63:32] << 32 | (row + 1)
//            → tag match is verified on the build row's key columns (late materialised).
// Capacity nextPow2(2n) (load factor <= 0.5) instead of the CPU's chained nextPow2(1.25n).
// Duplicate build keys occupy separate slots; a probe walks until the first empty slot.
// Unique build sides (verified while building) produce pairs through a dense match vector +
// ordered ballot-bitmap compaction (no output-cursor atomics); duplicated build keys use
// wave-aggregated appends; SEMI/ANTI/MARK results come out in ascending probe order.
#include "ldb_internal.h"
#include "ldb_join_kernel.h"
#include "ldb_jit.h"
#include <cmath>
#include "ldb_chain.h"
#include <algorithm>
#include <memory>

struct ldb_hashtable {
   ldb_ctx* ctx = nullptr;
   ldb_rel* build = nullptr; // referenced (kept alive by the caller until release)
   std::vector<ldb_colref> keys;
   uint64_t* slots = nullptr;
   uint64_t cap = 0;
   int32_t key32 = 0;
   int32_t unique = 0;
   int32_t ordered_slots = 0; // KEY32: slots follow the key order (DJoin::ordered_slots)
   int64_t kmin = 0, kmax = -1;
   uint64_t kmult = 0;
   uint32_t kmult32 = 0, ksh = 0; // DJoin::slot32
   int32_t slot32 = 0;
   uint32_t* key_bits = nullptr; // one bit per key value of [kmin, kmax] (DJoin::has_key_bits), or NULL
   int32_t chained = 0; // one slot per distinct key, rows linked through next[] (DJoin::chained)
   uint32_t* next = nullptr;
   int32_t direct = 0; // 1: slots = uint32_t[kmax - kmin + 1] indexed by key - kmin; 2: rank-bitmap words (DJoin::direct)
   int32_t rank_sorted = 0;
   uint32_t* coarse = nullptr; // one bit per 2^coarse_shift key values (DJoin::has_coarse), or NULL
   uint32_t coarse_shift = 6;
   uint32_t coarse_words = 0; // direct == 2: a key's rank IS its build row (ascending keys without NULLs); else next[] = rank → row
   size_t slot_bytes = 0; // bytes of the slot array (cap x 8, or cap x 4 when direct)
   int32_t pair32 = 0; // two 4-byte keys: slots are pairs of words (DJoin::pair32), cap x 16 bytes
   // the table owns its device buffers: an early error return from the build frees them with the object
   ~ldb_hashtable() {
      if (!ctx) return;
      ldb_dev_free(ctx, slots);
      ldb_dev_free(ctx, key_bits);
      ldb_dev_free(ctx, next);
      ldb_dev_free(ctx, coarse);
   }
};

// ---------------------------------------------------------------- ahead-of-time (generic) kernels
// generic kernels: one translation unit each (ldb_join_gk_*.hip)
__global__ void k_join_build(const DJoin* __restrict__ d);
__global__ void k_join_key_range(const DJoin* __restrict__ d, long long* __restrict__ out);
__global__ void k_join_key_bits(const DJoin* __restrict__ d);
__global__ void k_join_rank_bits(const DJoin* __restrict__ d);
__global__ void k_join_rank_perm(const DJoin* __restrict__ d);
// coarse bit b ⇔ some build key among the 2^shift key values b << shift … (shift = 6: rank words 2b, 2b + 1; 5: word b; 4 … 0: a 16 … 1-bit piece of one word's
// presence half — at 0 the filter IS the presence bitmap)
__global__ void k_rank_coarse(const uint64_t* __restrict__ tab, uint64_t n_words, uint32_t* __restrict__ coarse, uint32_t coarse_words, uint32_t shift) {
   for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < coarse_words; c += gridDim.x * blockDim.x) {
      uint32_t m = 0;
      for (uint32_t b = 0; b < 32; b++) {
         const uint64_t bit = (uint64_t) c * 32 + b; // covers key values [bit << shift, (bit + 1) << shift); a rank word's low half = 32 presence bits
         uint32_t any;
         if (shift == 6) {
            const uint64_t w = bit * 2;
            any = (w < n_words ? (uint32_t) tab[w] : 0u) | (w + 1 < n_words ? (uint32_t) tab[w + 1] : 0u);
         } else if (shift == 5) {
            any = bit < n_words ? (uint32_t) tab[bit] : 0u;
         } else {
            const uint64_t w = bit >> (5u - shift); // 2^(5 - shift) coarse bits per rank word
            const uint32_t piece = (uint32_t) (bit & ((1u << (5u - shift)) - 1u)), width = 1u << shift;
            any = w < n_words ? ((uint32_t) tab[w] >> (piece * width)) & ((1u << width) - 1u) : 0u;
         }
         if (any) m |= 1u << b;
      }
      coarse[c] = m;
   }
}
// rank-bitmap build, pass 2 (after k_join_rank_bits set the presence halves): popcounts → scan → prefix halves as ONE chained launch (ldb_chain.h): tab[w] = presence bits | (number of
// build keys in the words before w) << 32; *total = number of distinct build keys
__global__ __launch_bounds__(256) void k_rank_prefix_chain(uint64_t* __restrict__ tab, uint64_t n_words, uint64_t n_tiles, unsigned long long* __restrict__ total,
                                                           unsigned long long* __restrict__ status, unsigned long long* __restrict__ ticket, unsigned long long ticket_base,
                                                           unsigned long long epoch) {
   __shared__ unsigned long long s_tile;
   __shared__ uint32_t s_wave[4];
   __shared__ uint32_t s_prefix;
   if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1ull) - ticket_base;
   __syncthreads();
   const uint64_t tile = s_tile;
   const uint64_t base = tile * CHAIN_TILE + (uint64_t) threadIdx.x * CHAIN_ITEMS;
   uint32_t bits[CHAIN_ITEMS];
   uint32_t sum = 0;
#pragma unroll
   for (int k = 0; k < CHAIN_ITEMS; k++) {
      bits[k] = base + k < n_words ? (uint32_t) tab[base + k] : 0u;
      sum += (uint32_t) __popc(bits[k]);
   }
   uint32_t agg;
   uint32_t excl = d_block_scan256<uint32_t>(sum, s_wave, &agg);
   if (threadIdx.x < 64) {
      const uint32_t prefix = d_chain_prefix<uint32_t, 32>(status, tile, epoch, agg, threadIdx.x);
      if (threadIdx.x == 0) s_prefix = prefix;
   }
   __syncthreads();
   excl += s_prefix;
   if (total && tile == n_tiles - 1 && threadIdx.x == 0) *total = (unsigned long long) s_prefix + agg;
#pragma unroll
   for (int k = 0; k < CHAIN_ITEMS; k++) {
      if (base + k < n_words) tab[base + k] = (uint64_t) bits[k] | ((uint64_t) excl << 32);
      excl += (uint32_t) __popc(bits[k]);
   }
}
__global__ void k_join_probe_pairs(const DJoin* __restrict__ d);
__global__ void k_join_probe_pairs_count(const DJoin* __restrict__ d);
__global__ void k_join_probe_count(const DJoin* __restrict__ d);
__global__ void k_join_probe_exists(const DJoin* __restrict__ d);
__global__ void k_join_probe_unique(const DJoin* __restrict__ d);
__global__ void k_join_probe_markbuild(const DJoin* __restrict__ d);
__global__ void k_join_flags_bitmap(const uint8_t* __restrict__ flags, uint64_t n, int anti, uint64_t* __restrict__ bitmap, unsigned long long* __restrict__ counter) {
   join_flags_bitmap_body(flags, n, anti, bitmap, counter);
}
__global__ void k_join_flags_bitmap2(const uint8_t* __restrict__ flags, const uint8_t* __restrict__ not_flags, uint64_t n, uint64_t* __restrict__ bitmap, unsigned long long* __restrict__ counter) {
   join_flags_bitmap_body(flags, n, 0, bitmap, counter, not_flags);
}

// run-time specialised variants (hiprtc; ldb_jit.hip)
static const char* JOIN_SPEC_SRC =
   "extern \"C\" __global__ void k_join_build_spec(const DJoin* __restrict__ d) { join_build_body(LDB_META, d); }\n"
   "extern \"C\" __global__ void k_join_probe_pairs_spec(const DJoin* __restrict__ d) { join_probe_pairs_body(LDB_META, d); }\n"
   "extern \"C\" __global__ void k_join_probe_pairs_count_spec(const DJoin* __restrict__ d) { join_probe_pairs_count_body(LDB_META, d); }\n"
   "extern \"C\" __global__ void k_join_probe_count_spec(const DJoin* __restrict__ d) { join_probe_count_body(LDB_META, d); }\n"
   "extern \"C\" __global__ void k_join_probe_exists_spec(const DJoin* __restrict__ d) { join_probe_exists_body(LDB_META, d); }\n"
   "extern \"C\" __global__ void k_join_probe_unique_spec(const DJoin* __restrict__ d) { join_probe_unique_body(LDB_META, d); }\n"
   "extern \"C\" __global__ void k_join_probe_markbuild_spec(const DJoin* __restrict__ d) { join_probe_markbuild_body(LDB_META, d); }\n";

typedef void (*join_kernel_t)(const DJoin*);
// launch `generic` or its specialised twin `<name>_spec` on the ctx stream
static int32_t launch_join(ldb_ctx* ctx, const DJoin* h, const DJoin* d, int grid, const char* prof_name, const char* spec_name, join_kernel_t generic) {
   hipFunction_t spec = nullptr;
   if (ldb_jit_wanted((int64_t) h->n_rows)) {
      auto meta = std::make_unique<DJoin>();
      memcpy(meta.get(), h, sizeof(DJoin));
      meta->n_rows = meta->cap = meta->slots = meta->out_probe = meta->out_build = meta->out_cap = 0;
      meta->counter = meta->bitmap = meta->mark = meta->mark2 = meta->match = meta->flags = 0;
      meta->kmin = meta->kmax = 0;
      meta->kmult = 0;
      meta->kmult32 = meta->ksh = 0;
      meta->key_bits = 0;
      meta->next = 0;
      meta->coarse = 0;
      meta->coarse_words = 0;
      ldb_jit_strip_keys(meta->bkeys);
      ldb_jit_strip_keys(meta->pkeys);
      for (int p = 0; p < LDB_MAX_PREDS; p++) ldb_jit_strip_pred(meta->ppreds[p]);
      for (int p = 0; p < LDB_MAX_M2PREDS; p++) ldb_jit_strip_pred(meta->m2preds[p]);
      for (int k = 0; k < LDB_MAX_RESID; k++) {
         ldb_jit_strip_col(meta->resid[k].pcol);
         ldb_jit_strip_col(meta->resid[k].bcol);
      }
      std::string why;
      // only the wanted kernel's wrapper goes into the translation unit: compiling all seven for every
      // descriptor made a new probe shape cost 7x its share of hiprtc time
      std::string one;
      for (const char* line = JOIN_SPEC_SRC; *line;) {
         const char* nl = strchr(line, '\n');
         const size_t len = nl ? (size_t) (nl - line) + 1 : strlen(line);
         std::string l(line, len);
         if (l.find(std::string(" ") + spec_name + "(") != std::string::npos) one = l;
         line += len;
      }
      spec = ldb_jit_kernel(ctx->device, "ldb_join_kernel.h", "DJoin", one.empty() ? JOIN_SPEC_SRC : one.c_str(), spec_name, meta.get(), sizeof(DJoin), &why);
   }
   // a launch with a coarse key bitmap: 512-thread workgroups, each staging the bitmap (<= 40 KB) in dynamic LDS — four of
   // them share a CU's 160 KB, i.e. the same 32 waves per CU as the 256-thread launches
   const unsigned lds = h->has_coarse ? 4u * h->coarse_words : 0u;
   const bool big_lds = lds > 40u * 1024u; // the fine filter (<= 156 KB): one 1024-thread workgroup per CU
   const bool tiles = h->n_ppreds > 0; // the tile kernels are written for 256-thread workgroups: their launch shape stays, the filter only adds dynamic LDS
   const unsigned block = (h->has_coarse && !tiles) ? (big_lds ? 1024u : 512u) : 256u;
   if (h->has_coarse && !tiles) grid = (int) std::max<int64_t>(1, std::min<int64_t>(((int64_t) h->n_rows + block - 1) / block, (int64_t) ctx->cus * (big_lds ? 1 : 4)));
   if (big_lds) { // more than 64 KB of dynamic LDS per workgroup has to be asked for
      if (spec) (void) hipFuncSetAttribute((const void*) spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
      else (void) hipFuncSetAttribute((const void*) generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
      (void) hipGetLastError();
   }
   LdbProf prof_(ctx, prof_name);
   if (spec) {
      void* params[] = {(void*) &d};
      LDB_HIP(hipModuleLaunchKernel(spec, (unsigned) grid, 1, 1, block, 1, 1, lds, ctx->stream, params, nullptr));
   } else {
      hipLaunchKernelGGL(generic, dim3(grid), dim3(block), lds, ctx->stream, d);
   }
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}

bool ldb_join_jit_check(std::string* log) {
   // representative shape: a fact table with a fused two-conjunct date filter probing a unique,
   // ordered KEY32 table that also keeps key bits (TPC-H Q7's lineitem → supplier probe)
   auto m = std::make_unique<DJoin>();
   memset(m.get(), 0, sizeof(DJoin));
   m->key32 = 1;
   m->kind = LDB_JOIN_INNER;
   m->has_bitmap = 1;
   m->ordered_slots = 1;
   m->has_key_bits = 1;
   m->build_unique = 1;
   m->slot32 = 1;
   m->bkeys.n_keys = m->pkeys.n_keys = 1;
   m->bkeys.cols[0].type = m->pkeys.cols[0].type = LDB_T_INT32;
   m->bkeys.cols[0].width = m->pkeys.cols[0].width = 4;
   m->n_ppreds = 2;
   for (int p = 0; p < 2; p++) {
      m->ppreds[p].col.type = LDB_T_DATE32;
      m->ppreds[p].col.width = 4;
      m->ppreds[p].op = p ? LDB_F_LTE : LDB_F_GTE;
      m->ppreds[p].lo = p ? 9861 : 9131;
      m->ppreds[p].same_col = p;
   }
   if (const char* shape = getenv("LDB_JIT_CHECK_SHAPE")) { // offline ISA inspection of other shapes
      if (!strcmp(shape, "fk_count")) { // the FK probe micro-benchmark: no filter, key range too wide for key bits
         m->n_ppreds = 0;
         m->has_key_bits = 0;
         m->has_bitmap = 0;
      }
   }
   if (!ldb_jit_compile_only("ldb_join_kernel.h", "DJoin", JOIN_SPEC_SRC, m.get(), sizeof(DJoin), log)) return false;
   // second shape: the FK probe into a direct-addressed table (no key bits: the two-deep pipeline of DProbePipe)
   m->n_ppreds = 0;
   m->ordered_slots = 0;
   m->slot32 = 0;
   m->has_key_bits = 0;
   m->direct = 1;
   if (!ldb_jit_compile_only("ldb_join_kernel.h", "DJoin", JOIN_SPEC_SRC, m.get(), sizeof(DJoin), log)) return false;
   // third shape: the rank-bitmap table with ascending build keys (pipelined too), then with a rank → row permutation
   m->direct = 2;
   m->rank_sorted = 1;
   if (!ldb_jit_compile_only("ldb_join_kernel.h", "DJoin", JOIN_SPEC_SRC, m.get(), sizeof(DJoin), log)) return false;
   m->rank_sorted = 0;
   m->kind = LDB_JOIN_SEMI;
   if (!ldb_jit_compile_only("ldb_join_kernel.h", "DJoin", JOIN_SPEC_SRC, m.get(), sizeof(DJoin), log)) return false;
   // … and with the LDS-resident coarse key bitmap in front of it (512-thread workgroups)
   m->rank_sorted = 1;
   m->has_coarse = 32 | 6;
   return ldb_jit_compile_only("ldb_join_kernel.h", "DJoin", JOIN_SPEC_SRC, m.get(), sizeof(DJoin), log);
}

// ---------------------------------------------------------------- small helper kernels
__global__ void k_word_pop(const uint64_t* __restrict__ bitmap, uint32_t* __restrict__ pop, uint64_t n_words) {
   for (uint64_t w = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; w < n_words; w += (uint64_t) gridDim.x * blockDim.x) pop[w] = (uint32_t) __popcll(bitmap[w]);
}
// bitmap → ascending row ids: one wave per 64-bit word, offsets from a device scan of the word popcounts
__global__ void k_bitmap_expand(const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ word_off, uint32_t* __restrict__ out, uint64_t n_words) {
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   for (uint64_t w = wave; w < n_words; w += n_waves) {
      uint64_t m = bitmap[w];
      if ((m >> lane) & 1) out[word_off[w] + d_rank_in(m)] = (uint32_t) (w * 64 + lane);
   }
}
// the same for SPARSE bitmaps: one LANE per word — a wave reads 64 words with one coalesced load and only the lanes whose
// word is non-zero write (a wave per word spends a whole iteration on every empty word: 9.4 M iterations for a 600 M-row
// probe of which 95 % produce nothing)
__global__ void k_bitmap_expand_sparse(const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ word_off, uint32_t* __restrict__ out, uint64_t n_words) {
   for (uint64_t w = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; w < n_words; w += (uint64_t) gridDim.x * blockDim.x) {
      uint64_t m = bitmap[w];
      if (!m) continue;
      uint32_t at = word_off[w];
      const uint32_t base = (uint32_t) (w * 64);
      while (m) {
         out[at++] = base + (uint32_t) __builtin_ctzll(m);
         m &= m - 1;
      }
   }
}
// picks the scheme by density: fewer than one set bit in eight → lane per word
static void launch_bitmap_expand(ldb_ctx* ctx, const uint64_t* bitmap, const uint32_t* off, uint32_t* out, int64_t n_words, uint64_t total) {
   if (total * 8 < (uint64_t) n_words * 64) hipLaunchKernelGGL(k_bitmap_expand_sparse, dim3(ldb_grid_for(ctx, n_words, 256, 8)), dim3(256), 0, ctx->stream, bitmap, off, out, (uint64_t) n_words);
   else hipLaunchKernelGGL(k_bitmap_expand, dim3(ldb_grid_for(ctx, n_words * 64, 256, 8)), dim3(256), 0, ctx->stream, bitmap, off, out, (uint64_t) n_words);
}
__global__ void k_iota_u32j(uint32_t* out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = (uint32_t) i;
}
// out[j] = ids[sel[j]] with LDB_NULL_ROW passthrough
// all-match shortcut of a replayed unique probe: match[] of a row whose bitmap bit is clear was never written
__global__ void k_unmatched_to_zero(const uint64_t* __restrict__ bitmap, uint32_t* __restrict__ match, uint64_t n) {
   const uint64_t n_words = (n + 63) / 64;
   for (uint64_t w = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; w < n_words; w += (uint64_t) gridDim.x * blockDim.x) {
      uint64_t missing = ~bitmap[w];
      if (w == n_words - 1 && (n & 63)) missing &= (1ull << (n & 63)) - 1ull;
      while (missing) {
         const int b = __ffsll((unsigned long long) missing) - 1;
         match[w * 64 + (uint64_t) b] = 0u;
         missing &= missing - 1;
      }
   }
}
__global__ void k_compose_null(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ sel, uint32_t* __restrict__ out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint32_t s = sel[i];
      out[i] = s == LDB_NULL_ROW ? LDB_NULL_ROW : (ids ? ids[s] : s);
   }
}

// first[0, na) then second[0, nb) (identity when NULL), or na rows followed by nb NULL rows (pad_nulls)
__global__ void k_concat_rowids(const uint32_t* __restrict__ first, uint64_t na, const uint32_t* __restrict__ second, uint64_t nb, int pad_nulls, uint32_t* __restrict__ out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < na + nb; i += (uint64_t) gridDim.x * blockDim.x)
      out[i] = i < na ? (first ? first[i] : (uint32_t) i) : (pad_nulls ? LDB_NULL_ROW : (second ? second[i - na] : (uint32_t) (i - na)));
}

// debug_check option: largest row id of a selection vector (LDB_NULL_ROW ignored)
__global__ void k_max_rowid(const uint32_t* __restrict__ ids, uint64_t n, unsigned int* __restrict__ out) {
   unsigned int mx = 0;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t v = ids[i];
      if (v != LDB_NULL_ROW && v > mx) mx = v;
   }
   if (mx) atomicMax(out, mx);
}
static int32_t debug_check_ids(ldb_ctx* ctx, const char* what, const uint32_t* ids, uint64_t n, int64_t limit, const DJoin* h) {
   if (!ids || !n) return LDB_OK;
   unsigned int* d = (unsigned int*) (ctx->d_scratch + 40);
   LDB_HIP(hipMemsetAsync(d, 0, 8, ctx->stream));
   hipLaunchKernelGGL(k_max_rowid, dim3(ldb_grid_for(ctx, (int64_t) n, 256, 4)), dim3(256), 0, ctx->stream, ids, n, d);
   uint64_t mx = 0;
   LDB_TRY(ldb_read_u64(ctx, d, &mx));
   mx &= 0xFFFFFFFFull;
   if ((int64_t) mx >= limit)
      LDB_FAIL(LDB_ERR_INVALID, "debug_check: %s row id %llu >= %lld (kind %d key32 %d ordered %d key_bits %d chained %d unique %d n_ppreds %d n_rows %llu cap %llu)", what, (unsigned long long) mx,
               (long long) limit, h->kind, h->key32, h->ordered_slots, h->has_key_bits, h->chained, h->build_unique, h->n_ppreds, (unsigned long long) h->n_rows, (unsigned long long) h->cap);
   return LDB_OK;
}


// ---------------------------------------------------------------- radix clustering of the probe side
// An ordered KEY32 table is walked almost sequentially by a probe side that is clustered on the key
// (lineitem → orders).  An UNCLUSTERED probe side (random keys into a multi-GB slot array) pays one
// random DRAM access per row: 600 M probes into a 4.3 GB table run at 33 Grows/s against 190 for
// clustered keys.  The radix path first partitions the probe rows by SLOT RANGE — partition p holds
// the rows whose slot falls into [p, p + 1) * cap / P, P chosen so that one range is ~1 MB — with a
// histogram pass and a scatter pass (workgroup-private LDS cursors, (key, row) tuples written to
// contiguous per-(workgroup, partition) runs), and then probes the partitioned keys: every
// partition's slot range stays resident in the L2 / Infinity Cache while its rows are processed,
// and the ordinary probe kernels run on a dense key column again.  The reference has no
// counterpart (its chained table takes the cache misses, LazyJoinHashtable.cpp:12-34); this is the
// radix-partitioned join of the north star with the partitions held in cache instead of LDS.
#define RX_BLOCK 256
#define RX_MAX_PARTS 4096
struct DRadix {
   uint64_t n;
   uint64_t values; // probe key column (4-byte integers)
   uint64_t rowids; // its relation side's row ids or 0
   int64_t kmin, kmax;
   uint64_t kmult, mask;
   uint32_t kmult32, ksh, slot32, pshift, nparts, grid;
   uint64_t rows_per_wg;
   uint32_t direct, pad; // the table's layout (ldb_hashtable::direct): 1 = one slot per key value, 2 = one word per 32 key values
};
__device__ __forceinline__ uint32_t d_radix_part(const DRadix& d, uint32_t key) {
   const uint32_t r = key - (uint32_t) d.kmin;
   if (r > (uint32_t) (d.kmax - d.kmin)) return 0; // no slot: matches nothing, any partition will do
   const uint64_t pos = d.direct ? (uint64_t) (d.direct == 2 ? r >> 5 : r) : d.slot32 ? (uint64_t) __umulhi(r << d.ksh, d.kmult32) : ((((uint64_t) r * d.kmult) >> 32) & d.mask);
   return (uint32_t) (pos >> d.pshift);
}
__device__ __forceinline__ uint32_t d_radix_key(const DRadix& d, uint64_t i) {
   const uint32_t row = d.rowids ? gptr<uint32_t>(d.rowids)[i] : (uint32_t) i;
   return row == LDB_NULL_ROW ? 0x80000000u : (uint32_t) gptr<int32_t>(d.values)[row];
}
__global__ __launch_bounds__(RX_BLOCK) void k_radix_hist(DRadix d, uint32_t* __restrict__ hist) {
   __shared__ uint32_t h[RX_MAX_PARTS];
   for (uint32_t p = threadIdx.x; p < d.nparts; p += RX_BLOCK) h[p] = 0;
   __syncthreads();
   const uint64_t b = blockIdx.x * d.rows_per_wg, e = b + d.rows_per_wg < d.n ? b + d.rows_per_wg : d.n;
   for (uint64_t i = b + threadIdx.x; i < e; i += RX_BLOCK) atomicAdd(&h[d_radix_part(d, d_radix_key(d, i))], 1u);
   __syncthreads();
   for (uint32_t p = threadIdx.x; p < d.nparts; p += RX_BLOCK) hist[(uint64_t) p * d.grid + blockIdx.x] = h[p];
}
__global__ __launch_bounds__(RX_BLOCK) void k_radix_scatter(DRadix d, const uint32_t* __restrict__ offs, uint32_t* __restrict__ key_out, uint32_t* __restrict__ perm_out) {
   __shared__ uint32_t cur[RX_MAX_PARTS];
   for (uint32_t p = threadIdx.x; p < d.nparts; p += RX_BLOCK) cur[p] = offs[(uint64_t) p * d.grid + blockIdx.x];
   __syncthreads();
   const uint64_t b = blockIdx.x * d.rows_per_wg, e = b + d.rows_per_wg < d.n ? b + d.rows_per_wg : d.n;
   for (uint64_t i = b + threadIdx.x; i < e; i += RX_BLOCK) {
      const uint32_t key = d_radix_key(d, i);
      const uint32_t at = atomicAdd(&cur[d_radix_part(d, key)], 1u);
      key_out[at] = key;
      perm_out[at] = (uint32_t) i;
   }
}
// how local is the probe order already?  Share of sampled neighbouring rows whose slots lie within 64 KB
__global__ void k_radix_locality(DRadix d, uint64_t stride, unsigned int* __restrict__ out) {
   const uint64_t t = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x;
   const uint64_t i = t * stride;
   unsigned int near = 0, valid = 0;
   if (i + 1 < d.n) {
      const uint32_t a = d_radix_key(d, i) - (uint32_t) d.kmin, c = d_radix_key(d, i + 1) - (uint32_t) d.kmin;
      const uint64_t pa = d.direct ? (uint64_t) (d.direct == 2 ? a >> 5 : a) : d.slot32 ? (uint64_t) __umulhi(a << d.ksh, d.kmult32) : ((((uint64_t) a * d.kmult) >> 32) & d.mask);
      const uint64_t pc = d.direct ? (uint64_t) (d.direct == 2 ? c >> 5 : c) : d.slot32 ? (uint64_t) __umulhi(c << d.ksh, d.kmult32) : ((((uint64_t) c * d.kmult) >> 32) & d.mask);
      valid = 1;
      near = (pa > pc ? pa - pc : pc - pa) < 8192 ? 1 : 0;
   }
   const unsigned long long mn = __ballot(near), mv = __ballot(valid);
   if ((threadIdx.x & 63) == 0) {
      atomicAdd(out, (unsigned int) __popcll(mn));
      atomicAdd(out + 1, (unsigned int) __popcll(mv));
   }
}

struct RadixProbe { // the clustered stand-in for a probe relation
   ldb_rel* rel = nullptr; // sides: [keys table (identity)] + the probe's sides through the permutation
   ldb_table* keys = nullptr;
   // rank-table probes whose partitions are LDS-sized: where partition q begins (part_offs[q * chunks]) and what it covers
   uint32_t* part_offs = nullptr;
   uint32_t chunks = 0, nparts = 0, shift = 0;
};
static void radix_release(ldb_ctx* ctx, RadixProbe& rp) {
   if (rp.rel) ldb_gpu_rel_release(ctx, rp.rel);
   if (rp.keys) ldb_gpu_table_release(ctx, rp.keys);
   ldb_dev_free(ctx, rp.part_offs);
   rp.rel = nullptr;
   rp.keys = nullptr;
   rp.part_offs = nullptr;
}

// ---------------------------------------------------------------- LDS-staged probe of a radix-partitioned probe side
// North star: "radix-partitioned hash-join build/probe … LDS-staged hash buckets".  After the write-combining partition
// (ldb_wc.hip) the probe keys of partition q all fall into ONE slice of the rank table (2^shift key values = 2^(shift-5)
// 8-byte words, at most 64 KB).  One workgroup of 1 024 threads per partition copies the slice into LDS once and probes its
// keys there: a probe is an LDS read (32 banks) instead of one L2 line request per lane — the L1 / TA pipe handles 64
// distinct lines per wave instruction at ~1 line per cycle, which is what bounds the cache-resident probe (3.4 ms for
// 600 M keys).  A 64-row chunk belongs to the workgroup whose partition holds the chunk's first row; the few rows of a chunk
// that belong to the next partition read their word from memory.  Outputs are those of the ordinary unique probe (dense
// match[] + ballot bitmap + counter) so everything downstream is unchanged.
struct DPartProbe {
   uint64_t n, tab_words;
   const int32_t* keys;
   const uint64_t* tab;
   const uint32_t* perm; // rank → build row (NULL: the rank is the row)
   const uint32_t* part_offs;
   uint32_t chunks, nparts, words_per_part, range;
   int64_t kmin;
   uint32_t* match;
   uint64_t* bitmap;
   unsigned long long* counter;
   int32_t mode, has_bitmap; // mode 0: count matches (counter[1]); 1: unique pairs (match, bitmap, counter[0])
};
__global__ __launch_bounds__(1024) void k_join_probe_lds(DPartProbe d) {
   extern __shared__ uint64_t s_tab[];
   const uint32_t p = blockIdx.x;
   const uint64_t b = d.part_offs[(uint64_t) p * d.chunks], e = p + 1 < d.nparts ? (uint64_t) d.part_offs[(uint64_t) (p + 1) * d.chunks] : d.n;
   const uint64_t cb = (b + 63) / 64, ce = (e + 63) / 64; // the 64-row chunks whose first row lies in [b, e)
   if (cb >= ce) return;
   const uint64_t w0 = (uint64_t) p * d.words_per_part;
   for (uint32_t k = threadIdx.x; k < d.words_per_part; k += blockDim.x) s_tab[k] = w0 + k < d.tab_words ? d.tab[w0 + k] : 0ull;
   __syncthreads();
   const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
   unsigned long long local = 0;
   for (uint64_t c = cb + wave; c < ce; c += n_waves) {
      const uint64_t i = c * 64 + lane;
      uint32_t hw = 0; // rank + 1, or 0: no such key
      if (i < d.n) {
         const uint32_t r = (uint32_t) d.keys[i] - (uint32_t) d.kmin;
         if (r <= d.range) {
            const uint64_t wi = (uint64_t) (r >> 5);
            const uint64_t w = (wi >= w0 && wi < w0 + d.words_per_part) ? s_tab[wi - w0] : d.tab[wi];
            hw = d_rank_word(w, r);
         }
      }
      const uint64_t mm = __ballot(hw != 0);
      if (d.mode == 1) {
         const uint32_t brow = hw ? (d.perm ? d.perm[hw - 1u] : hw - 1u) : LDB_NULL_ROW;
         if (i < d.n && (hw || !d.has_bitmap)) d.match[i] = brow;
         if (lane == 0 && d.has_bitmap) d.bitmap[c] = mm;
      }
      local += (unsigned long long) __popcll(mm);
   }
   if (lane == 0 && local) atomicAdd(d.counter + (d.mode == 0 ? 1 : 0), local);
}
// the LDS-staged probe is possible: a rank table, partitions known, a slice fits 64 KB, plain key equality
static bool part_probe_ok(const ldb_hashtable* ht, const RadixProbe* rp, int32_t n_resid) {
   return rp && rp->part_offs && ht->direct == 2 && n_resid == 0 && rp->shift >= 5 && (8ull << (rp->shift - 5)) <= (64u << 10) && ldb_option("join_radix_lds", 1) != 0;
}
static int32_t launch_part_probe(ldb_ctx* ctx, ldb_hashtable* ht, const RadixProbe* rp, int64_t n, int mode, uint32_t* match, uint64_t* bitmap, unsigned long long* counter) {
   DPartProbe d;
   memset(&d, 0, sizeof(d));
   d.n = (uint64_t) n;
   d.tab_words = (uint64_t) ((ht->kmax - ht->kmin) / 32 + 1);
   d.keys = (const int32_t*) rp->keys->cols[0].values;
   d.tab = ht->slots;
   d.perm = ht->rank_sorted ? nullptr : ht->next;
   d.part_offs = rp->part_offs;
   d.chunks = rp->chunks;
   d.nparts = rp->nparts;
   d.words_per_part = 1u << (rp->shift - 5);
   d.range = (uint32_t) (ht->kmax - ht->kmin);
   d.kmin = ht->kmin;
   d.match = match;
   d.bitmap = bitmap;
   d.counter = counter;
   d.mode = mode;
   d.has_bitmap = bitmap ? 1 : 0;
   LdbProf prof_(ctx, mode == 0 ? "k_join_probe_count" : "k_join_probe_unique");
   hipLaunchKernelGGL(k_join_probe_lds, dim3(rp->nparts), dim3(1024), (size_t) d.words_per_part * 8, ctx->stream, d);
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}
// decides whether to cluster and, if so, builds the clustered relation; rp.rel stays NULL otherwise
static int32_t radix_prepare(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int32_t kind, RadixProbe& rp) {
   // 0 off, 1 whenever possible, -1 auto (default since round 4): a dense, unfiltered probe column of >= 16 M rows into a
   // direct / rank table of >= 64 MB whose keys a locality sample finds unclustered is partitioned with the write-combining
   // scatter (ldb_wc.hip) into LDS-sized table slices and probed from LDS: 600 M random FK probes 10.9 → 7.6 ms (DESIGN.md §2
   // Join).  Clustered probe sides (every large TPC-H probe) keep the direct probe; row-id or lazily filtered probe sides too
   // (the one-pass cursor scatter they would need loses to the direct probe).
   const int64_t mode = ldb_option("join_radix", -1);
   if (mode == 0 || n_keys != 1 || !ht->key32 || !(ht->ordered_slots || ht->direct) || ht->chained || probe->n_rows < 2) return LDB_OK;
   if (kind == LDB_JOIN_MARK || kind == LDB_JOIN_SEMI || kind == LDB_JOIN_ANTI) return LDB_OK; // results promised in probe order
   if (probe->sides.size() + 1 + ht->build->sides.size() > LDB_MAX_SIDES) return LDB_OK;
   DCol kc;
   LDB_TRY(ldb_make_dcol(probe, keys[0], &kc));
   if (kc.width != 4 || kc.validity || kc.type == LDB_T_FLOAT32) return LDB_OK;
   // slots of the table and their size: a rank table has one 8-byte word per 32 key values
   const uint64_t units = ht->direct == 2 ? (ht->cap + 31) / 32 : ht->cap;
   const uint64_t unit_bytes = ht->direct == 1 ? 4 : 8;
   if (mode < 0 && (units * unit_bytes < (uint64_t) ldb_option("join_radix_min_table_bytes", 64ll << 20) || probe->n_rows < ldb_option("join_radix_min_rows", 16ll << 20))) return LDB_OK;
   if (mode < 0 && (!probe->pending.empty() || kc.rowids || !ht->direct || ldb_option("join_radix_wc", 1) == 0)) return LDB_OK; // auto: only where the write-combining path applies
   LDB_TRY(ldb_rel_force(ctx, probe));
   LDB_TRY(ldb_make_dcol(probe, keys[0], &kc));
   const int64_t n = probe->n_rows;
   DRadix d;
   memset(&d, 0, sizeof(d));
   d.n = (uint64_t) n;
   d.values = kc.values;
   d.rowids = kc.rowids;
   d.kmin = ht->kmin;
   d.kmax = ht->kmax;
   d.kmult = ht->kmult;
   d.mask = ht->cap - 1;
   d.kmult32 = ht->kmult32;
   d.ksh = ht->ksh;
   d.slot32 = (uint32_t) ht->slot32;
   d.direct = (uint32_t) ht->direct;
   if (mode < 0) { // already clustered on the key?  then the table is walked sequentially as it is
      unsigned int* dl = (unsigned int*) (ctx->d_scratch + 40);
      LDB_HIP(hipMemsetAsync(dl, 0, 8, ctx->stream));
      const uint64_t samples = 1 << 16, stride = std::max<uint64_t>(1, (uint64_t) n / samples);
      hipLaunchKernelGGL(k_radix_locality, dim3((unsigned) (samples / 256)), dim3(256), 0, ctx->stream, d, stride, dl);
      uint64_t both = 0;
      LDB_TRY(ldb_read_u64(ctx, dl, &both));
      const uint64_t near = both & 0xFFFFFFFFull, valid = both >> 32;
      if (valid == 0 || near * 2 >= valid) return LDB_OK;
   }
   // partitions of ~1 MB of slots each
   uint32_t nparts = 16;
   const uint64_t part_bytes = (uint64_t) ldb_option("join_radix_part_bytes", ht->direct == 2 ? (64 << 10) : (1 << 20)); // rank table: slices that fit LDS
   while (nparts < RX_MAX_PARTS && (units * unit_bytes) / nparts > part_bytes) nparts <<= 1;
   uint32_t lg = 0;
   while ((1ull << lg) < units) lg++;
   uint32_t lp = 0;
   while ((1u << lp) < nparts) lp++;
   if (lg < lp) return LDB_OK;
   d.nparts = nparts;
   d.pshift = lg - lp;
   d.grid = (uint32_t) std::min<int64_t>(ctx->cus * 8, (n + 4095) / 4096);
   d.rows_per_wg = ((uint64_t) n + d.grid - 1) / d.grid;
   uint32_t *hist = nullptr, *offs = nullptr, *perm;
   const size_t hn = (size_t) nparts * d.grid;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &perm, 4 * (size_t) n));
   ldb_coltype kt = {LDB_T_INT32, 0, 0, 0};
   kt.type = kc.type;
   const char* nm = "radix_key";
   LDB_TRY(ldb_gpu_table_alloc(ctx, "radix_keys", 1, &kt, &nm, n, nullptr, 0, &rp.keys));
   if (ldb_option("join_radix_wc", 1) != 0 && ht->direct && !kc.rowids && nparts > 16 && ht->kmax - ht->kmin <= (int64_t) 0xFFFFFFFFll) {
      // write-combining partition (ldb_wc.hip): tile-sort in LDS, full-line runs, two passes above 64 partitions — the
      // scatter below keeps `nparts` open 4-byte streams per workgroup and loses to the direct probe beyond ~16 partitions
      const uint32_t shift = d.pshift + (ht->direct == 2 ? 5u : 0u);
      const int32_t st = ldb_wc_partition(ctx, (const uint32_t*) kc.values, nullptr, (uint64_t) n, (uint32_t) (int32_t) ht->kmin, (uint32_t) (ht->kmax - ht->kmin), shift, nparts,
                                          (uint32_t*) rp.keys->cols[0].values, perm, &rp.part_offs, &rp.chunks, "k_radix_hist", "k_radix_scatter");
      if (st != LDB_OK) {
         ldb_dev_free(ctx, perm);
         return st;
      }
      rp.nparts = nparts;
      rp.shift = shift;
   } else {
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &hist, 4 * hn));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &offs, 4 * hn));
   {
      LdbProf prof_(ctx, "k_radix_hist");
      hipLaunchKernelGGL(k_radix_hist, dim3(d.grid), dim3(RX_BLOCK), 0, ctx->stream, d, hist);
   }
   LDB_TRY(ldb_exclusive_scan_u32(ctx, hist, offs, (int64_t) hn, nullptr));
   {
      LdbProf prof_(ctx, "k_radix_scatter");
      hipLaunchKernelGGL(k_radix_scatter, dim3(d.grid), dim3(RX_BLOCK), 0, ctx->stream, d, (const uint32_t*) offs, (uint32_t*) rp.keys->cols[0].values, perm);
   }
   LDB_HIP(hipGetLastError());
   }
   ldb_dev_free(ctx, hist);
   ldb_dev_free(ctx, offs);
   ldb_rel* r = ldb_rel_new(ctx);
   r->n_rows = n;
   r->sides.push_back(ldb_rel_side{rp.keys, nullptr, false});
   bool perm_taken = false;
   const int cg = ldb_grid_for(ctx, n, 256, 8);
   for (auto& s : probe->sides) {
      ldb_rel_side ns{s.table, nullptr, true, s.may_null};
      if (!s.rowids && !perm_taken) {
         ns.rowids = perm;
         perm_taken = true;
      } else {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, 4 * (size_t) n));
         hipLaunchKernelGGL(k_compose_null, dim3(cg), dim3(256), 0, ctx->stream, (const uint32_t*) s.rowids, (const uint32_t*) perm, ns.rowids, (uint64_t) n);
      }
      r->sides.push_back(ns);
   }
   if (!perm_taken) ldb_dev_free(ctx, perm);
   LDB_HIP(hipGetLastError());
   rp.rel = r;
   return LDB_OK;
}
// drop the key-table side a clustered probe put in front of the probe's own sides
static void radix_strip(ldb_ctx* ctx, ldb_rel* r) {
   if (r->sides.empty()) return;
   if (r->sides[0].owned) ldb_dev_free(ctx, r->sides[0].rowids);
   r->sides.erase(r->sides.begin());
}

static uint64_t next_pow2_u64(uint64_t v) {
   uint64_t p = 1;
   while (p < v) p <<= 1;
   return p;
}

int32_t ldb_rel_select(ldb_ctx* ctx, ldb_rel* in, uint32_t* sel, int64_t n_sel, ldb_rel** out);

// ---------------------------------------------------------------- build
extern "C" int32_t ldb_gpu_join_build(ldb_ctx* ctx, ldb_rel* build, const ldb_colref* keys, int32_t n_keys, int32_t build_unique, ldb_hashtable** out) {
   if (!ctx || !build || !out || n_keys < 1) LDB_FAIL(LDB_ERR_INVALID, "join_build: bad argument");
   LDB_TRY(ldb_rel_force(ctx, build)); // the table is sized from the exact row count
   auto ht = std::make_unique<ldb_hashtable>();
   ht->ctx = ctx;
   ht->build = build;
   ht->keys.assign(keys, keys + n_keys);
   ht->unique = build_unique;
   auto hp = std::make_unique<DJoin>();
   DJoin* h = hp.get();
   memset(h, 0, sizeof(*h));
   LDB_TRY(ldb_make_dkeys(build, keys, n_keys, &h->bkeys));
   const DCol& k0 = h->bkeys.cols[0];
   ht->key32 = (n_keys == 1 && (k0.type == LDB_T_INT32 || k0.type == LDB_T_DATE32 || k0.type == LDB_T_CHAR4 || k0.type == LDB_T_INT16 || k0.type == LDB_T_INT8)) ? 1 : 0;
   ht->cap = std::max<uint64_t>(64, next_pow2_u64((uint64_t) build->n_rows * 2));
   h->n_rows = (uint64_t) build->n_rows;
   h->key32 = ht->key32;
   // flags / counters: zeroed words of the context's arena (ldb_counters) — fresh ones for every pass instead of a clear
   uint32_t* dflags;
   LDB_TRY(ldb_counters(ctx, 1, (uint64_t**) &dflags));
   h->flags = (uint64_t) dflags;
   h->has_flags = 1;
   // KEY32: slots in key order (see DJoin::ordered_slots) — needs the build key range first
   const bool ordered_enabled = ldb_option("join_ordered", 1) != 0;
   if (ht->key32 && ordered_enabled && build->n_rows > 0) {
      long long got[2];
      // the key column's cached statistic (a superset of the build rows' range, free after its first
      // use) when it has one; otherwise one pass over the build keys
      const ldb_table* kt = build->sides[(size_t) keys[0].side].table;
      int64_t clo = 0, chi = -1;
      if (!kt->cols[(size_t) keys[0].col].validity && ldb_column_range(ctx, kt, keys[0].col, &clo, &chi) == LDB_OK && clo <= chi) {
         got[0] = clo;
         got[1] = chi;
      } else {
         long long* range = (long long*) (ctx->d_scratch + 32);
         const long long init[2] = {INT64_MAX, INT64_MIN};
         LDB_TRY(ldb_h2d_small(ctx, range, init, 16));
         LdbDesc<DJoin> dr_desc(ctx);
         LDB_TRY(dr_desc.upload(h, sizeof(*h)));
         DJoin* dr = dr_desc.p;
         hipLaunchKernelGGL(k_join_key_range, dim3(ldb_grid_for(ctx, build->n_rows, 256, 8)), dim3(256), 0, ctx->stream, dr, range);
         LDB_TRY(LDB_READBACK(ctx, got, range, 16));
         dr_desc.release();
      }
      // DIRECT addressing when the key range is at most a few times the build rows (primary keys, and
      // filtered subsets of one): the table is then no larger than the open-addressing array it replaces
      // (4 B x range against 8 B x nextPow2(2n) in [16n, 32n) bytes) and a probe is one 4-byte load
      const unsigned __int128 range0 = got[0] <= got[1] ? (unsigned __int128) ((__int128) got[1] - got[0]) + 1 : 0;
      // RANK-BITMAP table for (promised) unique keys over a range of at most 64 values per build row — or any build side over
      // a range of at most 2^26 values (a 16 MB table: a selective filter on a dimension table, Q17's 20 k of 20 M parts):
      // range / 4 bytes, one 8-byte load per probe, no collisions, nothing to clear but the words (DJoin::direct == 2)
      bool ranked = false;
      if (range0 > 0 && build_unique && ldb_option("join_direct", 1) != 0 && ldb_option("join_rank", 1) != 0 && range0 <= ((unsigned __int128) 1 << 32) &&
          (range0 <= (unsigned __int128) std::max<int64_t>(4096, 64 * build->n_rows) || range0 <= ((unsigned __int128) 1 << 26)) && got[0] >= INT32_MIN && got[1] <= INT32_MAX && build->n_rows < (int64_t) LDB_NULL_ROW) {
         const uint64_t n_words = (uint64_t) (range0 / 32) + 1;
         uint64_t* tab = nullptr;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &tab, 8 * (size_t) n_words));
         ht->slots = tab; // (owned by the table object from here on: freed on every error return)
         LDB_HIP(hipMemsetAsync(tab, 0, 8 * (size_t) n_words, ctx->stream));
         unsigned long long* counter;
         LDB_TRY(ldb_counters(ctx, 3, (uint64_t**) &counter));
         uint64_t* d_total = (uint64_t*) (counter + 2);
         h->direct = 2;
         h->kmin = got[0];
         h->kmax = got[1];
         h->slots = (uint64_t) tab;
         h->counter = (uint64_t) counter;
         LdbDesc<DJoin> dr_desc(ctx);
         LDB_TRY(dr_desc.upload(h, sizeof(*h)));
         DJoin* dr = dr_desc.p;
         {
            LdbProf prof_(ctx, "k_join_build");
            hipLaunchKernelGGL(k_join_rank_bits, dim3(ldb_grid_for(ctx, build->n_rows, 256, 8)), dim3(256), 0, ctx->stream, dr);
            // popcounts → scan → prefix halves, one launch
            const uint64_t n_tiles = (n_words + CHAIN_TILE - 1) / CHAIN_TILE;
            ChainCall c;
            LDB_TRY(ldb_chain_begin(ctx, n_tiles, false, &c));
            hipLaunchKernelGGL(k_rank_prefix_chain, dim3((unsigned) n_tiles), dim3(256), 0, ctx->stream, tab, n_words, n_tiles, (unsigned long long*) d_total, c.status, c.ticket, c.ticket_base, c.epoch);
            if (hipGetLastError() != hipSuccess) return ldb_chain_failed(ctx);
         }
         uint64_t back[3] = {0, 0, 0}; // non-NULL keys, —, distinct keys
         uint32_t fl[2] = {0, 0};
         LDB_TRY(LDB_READBACK(ctx, back, counter, 24));
         LDB_TRY(LDB_READBACK(ctx, fl, dflags, 8));
         if (back[0] == back[2]) { // every non-NULL key set a bit of its own: unique
            ranked = true;
            ht->direct = 2;
            ht->kmin = got[0];
            ht->kmax = got[1];
            ht->cap = (uint64_t) range0;
            ht->slot_bytes = 8 * (size_t) n_words;
            ht->rank_sorted = (fl[0] & 4u) ? 0 : 1;
            // the LDS-resident coarse filter for selective builds over a small key range (DJoin::has_coarse): at most 40 KB
            // (four 512-thread workgroups per CU) and worth it when most 64-key blocks are empty
            // Round 6: one bit per SIXTEEN key values where that pays.  p(g) = 1 - (1 - density)^g is the share of probes a g-key filter lets
            // through to the L2.  Q9's green parts (5.4 % of the key range): p(64) = 0.97 — useless — p(16) = 0.59; Q8's part filter (0.67 %):
            // 0.35 against 0.10.  The fine filter of a 20 M key range is 156 KB: ONE 1024-thread workgroup per CU instead of four of 512 — taken
            // when the coarse one would pass more than a quarter of the probes; where the fine filter itself fits 40 KB it simply replaces the
            // coarse one.  Option join_coarse_fine (default 1).
            uint32_t shift = 6;
            uint64_t cwords = (uint64_t) (range0 / 64 / 32) + 1;
            const double density = range0 > 0 ? (double) back[2] / (double) range0 : 1.0;
            const double pass64 = 1.0 - pow(1.0 - density, 64.0), pass16 = 1.0 - pow(1.0 - density, 16.0);
            const bool coarse_on = ldb_option("join_coarse", 1) != 0;
            bool want = coarse_on && cwords * 4 <= 40 * 1024 && (unsigned __int128) back[2] * 64 * 2 <= range0;
            if (coarse_on && ldb_option("join_coarse_fine", 1) != 0 && pass16 <= 0.7) {
               const uint64_t fine_words = (uint64_t) (range0 / 16 / 32) + 1;
               const bool small = fine_words * 4 <= 40 * 1024;
               if (small || (fine_words * 4 <= 156 * 1024 && (!want || pass64 > 0.25))) {
                  shift = 4;
                  cwords = fine_words;
                  want = true;
               }
            }
            // … and where a still finer one fits beside the tile kernels' queue (<= 16 KB: a supplier-sized key range), the finest that does: Q21's 4 %
            // of the 1 M supplier keys pass 48 % of the probes at 16 keys per bit, 28 % at 8; Q7's 8 %: 74 → 49 %.  (0 = the presence bits themselves.)
            if (coarse_on && ldb_option("join_coarse_finest", 1) != 0 && want && shift == 4 && cwords * 4 <= 16 * 1024) {
               while (shift > 0 && ((uint64_t) (range0 >> (shift - 1)) / 32 + 1) * 4 <= 16 * 1024 && 1.0 - pow(1.0 - density, (double) (1u << shift)) > 0.1) shift--;
               cwords = (uint64_t) (range0 >> shift) / 32 + 1;
            }
            if (want) {
               LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->coarse, 4 * (size_t) cwords));
               ht->coarse_words = (uint32_t) cwords;
               ht->coarse_shift = shift;
               hipLaunchKernelGGL(k_rank_coarse, dim3((unsigned) ((cwords + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t*) tab, n_words, ht->coarse, (uint32_t) cwords, shift);
               LDB_HIP(hipGetLastError());
            }
            if (!ht->rank_sorted) {
               LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->next, 4 * (size_t) (build->n_rows ? build->n_rows : 1)));
               h->next = (uint64_t) ht->next;
               LdbDesc<DJoin> dp_desc(ctx);
               LDB_TRY(dp_desc.upload(h, sizeof(*h)));
               DJoin* dp = dp_desc.p;
               hipLaunchKernelGGL(k_join_rank_perm, dim3(ldb_grid_for(ctx, build->n_rows, 256, 8)), dim3(256), 0, ctx->stream, dp);
               LDB_HIP(hipGetLastError());
               dp_desc.release();
            }
         } else { // duplicate keys: the promise does not hold — the general layouts below
            ldb_dev_free(ctx, tab);
            ht->slots = nullptr;
            h->direct = 0;
            h->slots = 0;
            h->counter = 0;
            ht->unique = 0;
            build_unique = 0;
         }
         dr_desc.release();
         LDB_TRY(ldb_counters(ctx, 1, (uint64_t**) &dflags)); // (the passes below start from clean flags)
         h->flags = (uint64_t) dflags;
      }
      if (ranked) {
         *out = ht.release();
         return LDB_OK;
      }
      if (range0 > 0 && ldb_option("join_direct", 1) != 0 && range0 <= (unsigned __int128) std::max<int64_t>(1024, 8 * build->n_rows) && range0 < ((unsigned __int128) 1 << 31) &&
          got[0] >= INT32_MIN && got[1] <= INT32_MAX) {
         ht->direct = 1;
         ht->kmin = got[0];
         ht->kmax = got[1];
         ht->cap = (uint64_t) range0;
         if (range0 <= ((unsigned __int128) 1 << 27) && range0 >= 4096) { // key bits: 32x smaller than the table, L2-resident for selective builds
            const size_t words = (size_t) ((range0 + 31) / 32);
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->key_bits, 4 * words));
            LDB_HIP(hipMemsetAsync(ht->key_bits, 0, 4 * words, ctx->stream));
         }
      } else if (got[0] <= got[1]) { // at least one non-NULL key
         ht->ordered_slots = 1;
         ht->kmin = got[0];
         ht->kmax = got[1];
         ht->kmult = (uint64_t) ((((unsigned __int128) ht->cap) << 32) / ((unsigned __int128) (got[1] - got[0]) + 1));
         {
            // 32-bit slot arithmetic: slot = mulhi32((key - kmin) << ksh, kmult32) with (range << ksh) in [cap, 2 cap)
            const unsigned __int128 range128 = (unsigned __int128) ((__int128) got[1] - got[0]) + 1;
            if (ht->cap <= (1ull << 31) && range128 <= ((unsigned __int128) 1 << 32) && got[0] >= INT32_MIN && got[1] <= INT32_MAX) {
               uint64_t range = (uint64_t) range128;
               uint32_t ksh = 0;
               while ((range << ksh) < ht->cap) ksh++;
               const unsigned __int128 km = (((unsigned __int128) ht->cap) << 32) / (unsigned __int128) (range << ksh);
               ht->kmult32 = km > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) km;
               ht->ksh = ksh;
               ht->slot32 = 1;
            }
         }
         // one bit per key value when that fits the L2 comfortably (DJoin::has_key_bits)
         const unsigned __int128 range = (unsigned __int128) ((__int128) got[1] - got[0]) + 1;
         if (range <= ((unsigned __int128) 1 << 27)) { // <= 16 MB of bits
            const size_t words = (size_t) ((range + 31) / 32);
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->key_bits, 4 * words));
            LDB_HIP(hipMemsetAsync(ht->key_bits, 0, 4 * words, ctx->stream));
         }
      }
   }
   // up to three passes: ordered slots → hashed slots (skewed key range) → chained (a key repeats so
   // often that one slot per row gives long runs); each pass stops early when it sees such a run
   const bool force_chained = ldb_option("join_chained", 0) == 1; // tests
   if (force_chained && !ht->direct) {
      ht->ordered_slots = 0;
      ldb_dev_free(ctx, ht->key_bits);
      ht->key_bits = nullptr;
   }
   if (force_chained || (ht->direct && !build_unique)) { // a direct table without the promise of unique keys chains from the start
      ht->chained = 1;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->next, 4 * (size_t) (build->n_rows ? build->n_rows : 1)));
   }
   // two 4-byte integer keys, open addressing: the key values live in the slot (DJoin::pair32)
   if (!ht->direct && !ht->chained && n_keys == 2 && ldb_option("join_pair32", 1) != 0) {
      bool narrow = true;
      for (int k = 0; k < 2; k++) {
         const DCol& c = h->bkeys.cols[k];
         narrow = narrow && c.width == 4 && (c.type == LDB_T_INT32 || c.type == LDB_T_DATE32 || c.type == LDB_T_CHAR4);
      }
      ht->pair32 = narrow ? 1 : 0;
   }
   ht->slot_bytes = (ht->direct ? 4 : ht->pair32 ? 16 : 8) * (size_t) ht->cap;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->slots, ht->slot_bytes));
   LDB_HIP(hipMemsetAsync(ht->slots, 0, ht->slot_bytes, ctx->stream));
   h->cap = ht->cap;
   h->slots = (uint64_t) ht->slots;
   for (int attempt = 0; attempt < 3; attempt++) {
      h->direct = ht->direct;
      h->ordered_slots = ht->ordered_slots;
      h->kmin = ht->kmin;
      h->kmax = ht->kmax;
      h->kmult = ht->kmult;
      h->kmult32 = ht->kmult32;
      h->ksh = ht->ksh;
      h->slot32 = ht->ordered_slots ? ht->slot32 : 0;
      h->key_bits = (uint64_t) ht->key_bits;
      h->has_key_bits = ht->key_bits ? 1 : 0;
      h->chained = ht->chained;
      h->next = (uint64_t) ht->next;
      if (ht->chained) ht->pair32 = 0; // (a chained rebuild uses the first cap words of the same allocation)
      h->pair32 = ht->pair32;
      LdbDesc<DJoin> d_desc(ctx);
      LDB_TRY(d_desc.upload(h, sizeof(*h)));
      DJoin* d = d_desc.p;
      if (build->n_rows && h->has_key_bits && (h->ordered_slots || h->direct)) hipLaunchKernelGGL(k_join_key_bits, dim3(ldb_grid_for(ctx, build->n_rows, 256, 8)), dim3(256), 0, ctx->stream, d);
      if (build->n_rows) LDB_TRY(launch_join(ctx, h, d, ldb_grid_for(ctx, build->n_rows, 256, 8), "k_join_build", "k_join_build_spec", k_join_build));
      d_desc.release();
      uint64_t f = 0;
      // open addressing: which insertion meets a long run depends on the order the insertions happen in, but whether ANY does is in practice
      // a property of the keys (a run of 512 needs hundreds of equal or colliding keys: then some insertion walks it in every order).  The
      // flag is replayed like any count — round 4 read it for real in every execution, one stream wait in the middle of every plan with an
      // open-addressing build (Q18: 11.5 of 13.8 ms of host time blocked in this call) — and a run that does come out differently is caught
      // by the comparison at the trace's end like any other mis-speculation: the execution is void (an insertion that gave up dropped its row)
      // and is repeated.  Such repeats are counted apart (LDB_RB_ORDER_DEPENDENT → ldb_gpu_order_dependent_misses, bench.py's prepared_plans
      // block): a key distribution that sits on the run-length threshold shows there instead of hiding among data-dependent misses
      LDB_TRY(ldb_read_u64_at(ctx, dflags, &f, LDB_SITE, LDB_RB_ORDER_DEPENDENT));
      if (ht->direct && (f & 1) && !ht->chained) { // duplicate keys in a direct table: chain them
         ht->chained = 1;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->next, 4 * (size_t) build->n_rows));
         LDB_HIP(hipMemsetAsync(ht->slots, 0, ht->slot_bytes, ctx->stream));
         LDB_TRY(ldb_counters(ctx, 1, (uint64_t**) &dflags));
         h->flags = (uint64_t) dflags;
         continue;
      }
      if ((f & 2) && !ht->chained) {
         if (ht->ordered_slots) { // this key distribution needs hashed slots
            ht->ordered_slots = 0;
            ldb_dev_free(ctx, ht->key_bits);
            ht->key_bits = nullptr;
         } else { // hashed and still long runs: duplicates → chain them
            ht->chained = 1;
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &ht->next, 4 * (size_t) build->n_rows));
         }
         LDB_HIP(hipMemsetAsync(ht->slots, 0, ht->slot_bytes, ctx->stream));
         LDB_TRY(ldb_counters(ctx, 1, (uint64_t**) &dflags));
         h->flags = (uint64_t) dflags;
         continue;
      }
      // the caller's promise of unique keys is verified: duplicates fall back to the general probe
      if ((f & 1) || ht->chained) ht->unique = 0;
      break;
   }
   *out = ht.release();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_hashtable_release(ldb_ctx* ctx, ldb_hashtable* ht) {
   (void) ctx;
   delete ht; // ~ldb_hashtable frees slots / key bits / chains
   return LDB_OK;
}
extern "C" int64_t ldb_gpu_hashtable_slots(const ldb_hashtable* ht) { return ht ? (int64_t) ht->cap : -1; }
extern "C" int64_t ldb_gpu_hashtable_bytes(const ldb_hashtable* ht) { return ht ? (int64_t) ht->slot_bytes : -1; }

static int32_t make_probe_desc(ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, DJoin* h, const ldb_join_residual* resid = nullptr,
                               int32_t n_resid = 0) {
   if ((size_t) n_keys != ht->keys.size()) LDB_FAIL(LDB_ERR_INVALID, "join_probe: %d probe keys vs %zu build keys", n_keys, ht->keys.size());
   memset(h, 0, sizeof(*h));
   LDB_TRY(ldb_make_dkeys(ht->build, ht->keys.data(), n_keys, &h->bkeys));
   LDB_TRY(ldb_make_dkeys(probe, keys, n_keys, &h->pkeys));
   for (int k = 0; k < n_keys; k++) {
      bool sa = h->bkeys.cols[k].type == LDB_T_UTF8, sb = h->pkeys.cols[k].type == LDB_T_UTF8;
      bool fa = h->bkeys.cols[k].type == LDB_T_FLOAT64 || h->bkeys.cols[k].type == LDB_T_FLOAT32;
      bool fb = h->pkeys.cols[k].type == LDB_T_FLOAT64 || h->pkeys.cols[k].type == LDB_T_FLOAT32;
      if (sa != sb || fa != fb) LDB_FAIL(LDB_ERR_INVALID, "join_probe: key %d type class differs between build and probe", k);
      // the hash must agree on both sides: date32 hashes in ns, plain ints as is
      if ((h->bkeys.cols[k].type == LDB_T_DATE32) != (h->pkeys.cols[k].type == LDB_T_DATE32)) LDB_FAIL(LDB_ERR_INVALID, "join_probe: key %d date vs non-date", k);
      bool wa = h->bkeys.cols[k].type == LDB_T_DECIMAL128 && h->bkeys.cols[k].precision >= 19;
      bool wb = h->pkeys.cols[k].type == LDB_T_DECIMAL128 && h->pkeys.cols[k].precision >= 19;
      if (wa != wb) LDB_FAIL(LDB_ERR_INVALID, "join_probe: key %d decimal width class differs (cast to a common type first)", k);
   }
   h->n_rows = (uint64_t) probe->n_rows;
   h->cap = ht->cap;
   h->slots = (uint64_t) ht->slots;
   h->key32 = ht->key32;
   h->ordered_slots = ht->ordered_slots;
   h->direct = ht->direct;
   h->rank_sorted = ht->rank_sorted;
   h->kmin = ht->kmin;
   h->kmax = ht->kmax;
   h->kmult = ht->kmult;
   h->kmult32 = ht->kmult32;
   h->ksh = ht->ksh;
   h->slot32 = ht->ordered_slots ? ht->slot32 : 0;
   h->key_bits = (uint64_t) ht->key_bits;
   h->has_key_bits = ht->key_bits ? 1 : 0;
   h->chained = ht->chained;
   h->next = (uint64_t) ht->next;
   h->pair32 = 0;
   if (ht->pair32 && !ht->chained) { // verify from the slot when the probe's key columns are 4-byte integers too, else through the rows
      bool narrow = true;
      for (int k = 0; k < 2; k++) {
         const DCol& c = h->pkeys.cols[k];
         narrow = narrow && c.width == 4 && (c.type == LDB_T_INT32 || c.type == LDB_T_DATE32 || c.type == LDB_T_CHAR4);
      }
      h->pair32 = narrow ? 1 : 2;
   }
   h->build_unique = (ht->unique && !ht->chained) ? 1 : 0;
   // a lazy probe relation brings its filter along: evaluated inside the probe kernel
   h->n_ppreds = (int32_t) probe->pending.size();
   // (the staging costs every workgroup the filter's bytes in loads: large probes only.  A probe with a fused filter runs the tile kernels — 256-thread
   // workgroups with their queue in static LDS, ~8 per CU —: only a filter of <= 16 KB rides along there (round 6: Q21's 380 M filtered probes of
   // the 1 M supplier keys, 4 % of them present: half the L2 requests end in LDS))
   if (ht->coarse && probe->n_rows >= (1 << 22) && (probe->pending.empty() || (4u * ht->coarse_words <= 16u * 1024u && ldb_option("join_coarse_filtered", 1) != 0))) {
      h->coarse = (uint64_t) ht->coarse;
      h->coarse_words = ht->coarse_words;
      h->has_coarse = (int32_t) (32u | ht->coarse_shift); // (32 | log2 of the key values per bit — 6, 4 … 0: the kernels read the granularity from here)
   }
   for (size_t p = 0; p < probe->pending.size(); p++) h->ppreds[p] = probe->pending[p];
   ldb_order_preds(h->ppreds, h->n_ppreds);
   if (n_resid < 0 || n_resid > LDB_MAX_RESID || (n_resid && !resid)) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_probe: %d residual conjuncts (max %d)", n_resid, LDB_MAX_RESID);
   h->n_resid = n_resid;
   for (int32_t k = 0; k < n_resid; k++) {
      LDB_TRY(ldb_make_dcol(probe, resid[k].probe_col, &h->resid[k].pcol));
      LDB_TRY(ldb_make_dcol(ht->build, resid[k].build_col, &h->resid[k].bcol));
      const int op = resid[k].op;
      if (op < LDB_F_EQ || op > LDB_F_GTE) LDB_FAIL(LDB_ERR_INVALID, "join_probe: residual %d: comparison operator expected", k);
      for (const DCol* c : {&h->resid[k].pcol, &h->resid[k].bcol})
         if (c->type == LDB_T_UTF8 || c->type == LDB_T_FLOAT32 || c->type == LDB_T_FLOAT64) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_probe: residual %d: integer / decimal / date columns only", k);
      h->resid[k].op = op;
   }
   return LDB_OK;
}

static int32_t probe_count_impl(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int64_t* matches, bool radix_ok, const RadixProbe* part = nullptr);
extern "C" int32_t ldb_gpu_join_probe_count(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int64_t* matches) {
   return probe_count_impl(ctx, ht, probe, keys, n_keys, matches, true);
}
static int32_t probe_count_impl(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int64_t* matches, bool radix_ok, const RadixProbe* part) {
   if (!ctx || !ht || !probe || !matches) LDB_FAIL(LDB_ERR_INVALID, "join_probe_count: NULL argument");
   if (radix_ok) {
      RadixProbe rp;
      LDB_TRY(radix_prepare(ctx, ht, probe, keys, n_keys, LDB_JOIN_INNER, rp));
      if (rp.rel) { // an unclustered probe side: partitioned by slot range first
         const ldb_colref k0 = {0, 0};
         const int32_t st = probe_count_impl(ctx, ht, rp.rel, &k0, 1, matches, false, &rp);
         radix_release(ctx, rp);
         return st;
      }
   }
   auto hp = std::make_unique<DJoin>();
   LDB_TRY(make_probe_desc(ht, probe, keys, n_keys, hp.get()));
   hp->kind = LDB_JOIN_INNER;
   unsigned long long* counter;
   LDB_TRY(ldb_counters(ctx, 2, (uint64_t**) &counter));
   hp->counter = (uint64_t) counter;
   LdbDesc<DJoin> d_desc(ctx);
   LDB_TRY(d_desc.upload(hp.get(), sizeof(DJoin)));
   DJoin* d = d_desc.p;
   if (probe->n_rows && part_probe_ok(ht, part, 0)) LDB_TRY(launch_part_probe(ctx, ht, part, probe->n_rows, 0, nullptr, nullptr, counter)); // partitions staged in LDS
   else if (probe->n_rows) LDB_TRY(launch_join(ctx, hp.get(), d, ldb_grid_for(ctx, probe->n_rows, 256, 8), "k_join_probe_count", "k_join_probe_count_spec", k_join_probe_count));
   uint64_t m = 0;
   LDB_TRY(ldb_read_u64(ctx, counter + 1, &m));
   d_desc.release();
   *matches = (int64_t) m;
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_join_probe(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int32_t kind, ldb_rel** out,
                                      ldb_table** mark_out) {
   return ldb_gpu_join_probe_residual(ctx, ht, probe, keys, n_keys, kind, nullptr, 0, out, mark_out);
}

static int32_t probe_impl(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int32_t kind, const ldb_join_residual* resid, int32_t n_resid,
                          ldb_rel** out, ldb_table** mark_out, bool radix_ok, const RadixProbe* part = nullptr);
extern "C" int32_t ldb_gpu_join_probe_residual(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int32_t kind,
                                               const ldb_join_residual* resid, int32_t n_resid, ldb_rel** out, ldb_table** mark_out) {
   return probe_impl(ctx, ht, probe, keys, n_keys, kind, resid, n_resid, out, mark_out, true);
}
// Build-side semi join AND build-side anti join against the same table in one pass over the probe side: *out = the build rows that have a
// partner among the probe rows (key equality + residual conjuncts) but NO partner among the probe rows that also satisfy `anti_preds`.
// Equal to SEMI_BUILD, a table over its result, and ANTI_BUILD probed by `probe` filtered on anti_preds — TPC-H Q21's EXISTS / NOT EXISTS
// pair, which the reference runs as two marker joins (translateHJWithMarker, RelAlgToSubOp.cpp:1248-1287) — with one walk of each chain.
extern "C" int32_t ldb_gpu_join_probe_semi_anti_build(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, const ldb_join_residual* resid,
                                                      int32_t n_resid, const ldb_filter_desc* anti_preds, int32_t n_anti_preds, ldb_rel** out) {
   if (!ctx || !ht || !probe || !out || n_keys < 1 || !anti_preds || n_anti_preds < 1) LDB_FAIL(LDB_ERR_INVALID, "join_probe_semi_anti_build: bad argument");
   if (n_anti_preds > LDB_MAX_M2PREDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_probe_semi_anti_build: more than %d conjuncts on the anti side", LDB_MAX_M2PREDS);
   LDB_TRY(ldb_rel_force(ctx, probe)); // (the second marker's conjuncts are evaluated per matching pair; a lazy probe filter is applied first)
   auto hb = std::make_unique<DJoin>();
   LDB_TRY(make_probe_desc(ht, probe, keys, n_keys, hb.get(), resid, n_resid));
   hb->kind = LDB_JOIN_SEMI_BUILD;
   for (int32_t p = 0; p < n_anti_preds; p++) LDB_TRY(ldb_make_dpred(probe, &anti_preds[p], &hb->m2preds[p]));
   hb->n_m2preds = n_anti_preds;
   const int64_t nb = ht->build->n_rows, nbw = (nb + 63) / 64;
   LdbBufs tmp(ctx);
   uint8_t *flags, *flags2;
   uint64_t* bitmap;
   LDB_TRY(tmp.alloc(&flags, 2 * (size_t) (nb ? nb : 1)));
   flags2 = flags + (nb ? nb : 1);
   LDB_TRY(tmp.alloc(&bitmap, 8 * (size_t) (nbw ? nbw : 1)));
   LDB_HIP(hipMemsetAsync(flags, 0, 2 * (size_t) (nb ? nb : 1), ctx->stream));
   hb->mark = (uint64_t) flags;
   hb->mark2 = (uint64_t) flags2;
   hb->has_mark = 1;
   unsigned long long* cnt;
   LDB_TRY(ldb_counters(ctx, 2, (uint64_t**) &cnt));
   LdbDesc<DJoin> d_desc(ctx);
   LDB_TRY(d_desc.upload(hb.get(), sizeof(DJoin)));
   DJoin* d = d_desc.p;
   int32_t st = LDB_OK;
   if (probe->n_rows) st = launch_join(ctx, hb.get(), d, ldb_grid_for(ctx, probe->n_rows, 256, 8), "k_join_probe_markbuild", "k_join_probe_markbuild_spec", k_join_probe_markbuild);
   d_desc.release();
   LDB_TRY(st);
   if (nb) hipLaunchKernelGGL(k_join_flags_bitmap2, dim3(ldb_grid_for(ctx, nb, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) flags, (const uint8_t*) flags2, (uint64_t) nb, bitmap, cnt);
   LDB_HIP(hipGetLastError());
   uint64_t total = 0;
   LDB_TRY(ldb_read_u64(ctx, cnt, &total));
   uint32_t* sel;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &sel, 4 * (size_t) (total ? total : 1)));
   if (nbw && total) {
      st = ldb_bitmap_compact(ctx, bitmap, nbw, sel, total, nullptr, nullptr, nullptr);
      if (st != LDB_OK) {
         ldb_dev_free(ctx, sel);
         return st;
      }
   }
   return ldb_rel_select(ctx, ht->build, sel, (int64_t) total, out);
}

static int32_t probe_impl(ldb_ctx* ctx, ldb_hashtable* ht, ldb_rel* probe, const ldb_colref* keys, int32_t n_keys, int32_t kind, const ldb_join_residual* resid, int32_t n_resid,
                          ldb_rel** out, ldb_table** mark_out, bool radix_ok, const RadixProbe* part) {
   if (!ctx || !ht || !probe || !out) LDB_FAIL(LDB_ERR_INVALID, "join_probe: NULL argument");
   if (kind < LDB_JOIN_INNER || kind > LDB_JOIN_FULL_OUTER) LDB_FAIL(LDB_ERR_INVALID, "join_probe: bad kind %d", kind);
   if (kind == LDB_JOIN_RIGHT_OUTER || kind == LDB_JOIN_FULL_OUTER) {
      // the pairs of the inner / left-outer join, then the build rows whose marker no probe tuple set (the reference
      // scans its HashMultiMap for unmarked entries after the probe pipeline, translateHJWithMarker)
      struct RelHold {
         ldb_ctx* ctx;
         ldb_rel* r = nullptr;
         ~RelHold() {
            if (r) ldb_gpu_rel_release(ctx, r);
         }
      } pairs{ctx}, unm{ctx};
      LDB_TRY(probe_impl(ctx, ht, probe, keys, n_keys, kind == LDB_JOIN_RIGHT_OUTER ? LDB_JOIN_INNER : LDB_JOIN_LEFT_OUTER, resid, n_resid, &pairs.r, nullptr, radix_ok));
      LDB_TRY(probe_impl(ctx, ht, probe, keys, n_keys, LDB_JOIN_ANTI_BUILD, resid, n_resid, &unm.r, nullptr, radix_ok));
      const int64_t n1 = pairs.r->n_rows, n2 = unm.r->n_rows, nn = n1 + n2;
      if (nn >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_probe: %ld result rows exceed uint32 row ids", (long) nn);
      const size_t np = probe->sides.size(), nb = ht->build->sides.size();
      if (pairs.r->sides.size() != np + nb || unm.r->sides.size() != nb) LDB_FAIL(LDB_ERR_INVALID, "join_probe: outer join pieces do not line up");
      ldb_rel* r = ldb_rel_new(ctx);
      r->n_rows = nn;
      const int cg = ldb_grid_for(ctx, nn, 256, 8);
      for (size_t j = 0; j < np + nb; j++) {
         const ldb_rel_side& a = pairs.r->sides[j];
         ldb_rel_side ns{a.table, nullptr, true, true};
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, 4 * (size_t) (nn ? nn : 1)));
         const uint32_t* tail = j >= np ? unm.r->sides[j - np].rowids : nullptr;
         if (nn) hipLaunchKernelGGL(k_concat_rowids, dim3(cg), dim3(256), 0, ctx->stream, (const uint32_t*) a.rowids, (uint64_t) n1, tail, (uint64_t) n2, j < np ? 1 : 0, ns.rowids);
         if (j >= np) ns.may_null = a.may_null || unm.r->sides[j - np].may_null;
         r->sides.push_back(ns);
      }
      LDB_HIP(hipGetLastError());
      *out = r;
      return LDB_OK;
   }
   if (radix_ok) {
      RadixProbe rp;
      LDB_TRY(radix_prepare(ctx, ht, probe, keys, n_keys, kind, rp));
      if (rp.rel) { // an unclustered probe side: partitioned by slot range first, then the ordinary kernels
         const ldb_colref k0 = {0, 0};
         std::vector<ldb_join_residual> rs(resid, resid + (n_resid > 0 ? n_resid : 0));
         for (auto& x : rs) x.probe_col.side += 1; // the key table sits in front of the probe's sides
         const int32_t st = probe_impl(ctx, ht, rp.rel, &k0, 1, kind, rs.data(), n_resid, out, mark_out, false, &rp);
         if (st == LDB_OK && kind != LDB_JOIN_SEMI_BUILD && kind != LDB_JOIN_ANTI_BUILD) radix_strip(ctx, *out);
         radix_release(ctx, rp);
         return st;
      }
   }
   // kinds that emit a row for EVERY probe row (outer / single / mark) need the filtered row set itself
   if (kind == LDB_JOIN_LEFT_OUTER || kind == LDB_JOIN_SINGLE || kind == LDB_JOIN_MARK) LDB_TRY(ldb_rel_force(ctx, probe));
   // a long conjunction is applied by the scan kernel first (round 6): fused, the tile kernels' filter stage passes over ALL rows once per conjunct in
   // front of a queue that then holds almost nothing — Q12's five conjuncts keep 0.5 % of lineitem: 3.5 ms fused, 2.3 ms as a scan + a probe of the
   // 3 M survivors.  Up to three conjuncts (every other filtered probe of the 22 plans: one or two) stay fused
   if ((int64_t) probe->pending.size() > ldb_option("join_fuse_max_conjuncts", 3)) LDB_TRY(ldb_rel_force(ctx, probe));
   const bool pairs = kind == LDB_JOIN_INNER || kind == LDB_JOIN_LEFT_OUTER || kind == LDB_JOIN_SINGLE;
   if (kind == LDB_JOIN_SEMI_BUILD || kind == LDB_JOIN_ANTI_BUILD) {
      // flag the build rows that some probe row matches, then keep (SEMI) / drop (ANTI) them
      auto hb = std::make_unique<DJoin>();
      LDB_TRY(make_probe_desc(ht, probe, keys, n_keys, hb.get(), resid, n_resid));
      hb->kind = kind;
      const int64_t nb = ht->build->n_rows, nbw = (nb + 63) / 64;
      uint8_t* flags;
      uint64_t* bitmap;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &flags, (size_t) (nb ? nb : 1)));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &bitmap, 8 * (size_t) (nbw ? nbw : 1)));
      LDB_HIP(hipMemsetAsync(flags, 0, (size_t) (nb ? nb : 1), ctx->stream));
      hb->mark = (uint64_t) flags;
      hb->has_mark = 1;
      unsigned long long* cnt;
      LDB_TRY(ldb_counters(ctx, 2, (uint64_t**) &cnt));
      LdbDesc<DJoin> d_desc(ctx);
      LDB_TRY(d_desc.upload(hb.get(), sizeof(DJoin)));
      DJoin* d = d_desc.p;
      if (probe->n_rows) LDB_TRY(launch_join(ctx, hb.get(), d, ldb_grid_for(ctx, probe->n_rows, 256, 8), "k_join_probe_markbuild", "k_join_probe_markbuild_spec", k_join_probe_markbuild));
      d_desc.release();
      if (nb) hipLaunchKernelGGL(k_join_flags_bitmap, dim3(ldb_grid_for(ctx, nb, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) flags, (uint64_t) nb, kind == LDB_JOIN_ANTI_BUILD ? 1 : 0, bitmap, cnt);
      LDB_HIP(hipGetLastError());
      uint64_t total = 0;
      LDB_TRY(ldb_read_u64(ctx, cnt, &total));
      uint32_t* sel;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &sel, 4 * (size_t) (total ? total : 1)));
      if (nbw && total) LDB_TRY(ldb_bitmap_compact(ctx, bitmap, nbw, sel, total, nullptr, nullptr, nullptr));
      ldb_dev_free(ctx, flags);
      ldb_dev_free(ctx, bitmap);
      return ldb_rel_select(ctx, ht->build, sel, (int64_t) total, out);
   }
   if (pairs && probe->sides.size() + ht->build->sides.size() > LDB_MAX_SIDES)
      LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_probe: result would have more than %d sides (materialize first)", LDB_MAX_SIDES);
   auto hp = std::make_unique<DJoin>();
   DJoin* h = hp.get();
   LDB_TRY(make_probe_desc(ht, probe, keys, n_keys, h, resid, n_resid));
   h->kind = kind;
   unsigned long long* counter; // [0] rows produced, [1] matches (+ 3 debug words): zeroed arena words
   LDB_TRY(ldb_counters(ctx, 6, (uint64_t**) &counter));
   h->counter = (uint64_t) counter;
   const int64_t n = probe->n_rows;
   const int64_t n_words = (n + 63) / 64;
   const int grid = ldb_grid_for(ctx, n, 256, 8);

   if (!pairs) { // SEMI / ANTI / MARK
      uint64_t* bitmap;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &bitmap, 8 * (size_t) (n_words ? n_words : 1)));
      h->bitmap = (uint64_t) bitmap;
      h->has_bitmap = 1;
      ldb_table* mark = nullptr;
      if (kind == LDB_JOIN_MARK) {
         if (!mark_out) LDB_FAIL(LDB_ERR_INVALID, "join_probe: MARK needs mark_out");
         ldb_coltype t = {LDB_T_BOOL8, 0, 0, 0};
         const char* nm = "mark";
         LDB_TRY(ldb_gpu_table_alloc(ctx, "mark", 1, &t, &nm, n, nullptr, 0, &mark));
         h->mark = (uint64_t) mark->cols[0].values;
         h->has_mark = 1;
      }
      LdbDesc<DJoin> d_desc(ctx);
      LDB_TRY(d_desc.upload(h, sizeof(*h)));
      DJoin* d = d_desc.p;
      if (n) LDB_TRY(launch_join(ctx, h, d, grid, "k_join_probe_exists", "k_join_probe_exists_spec", k_join_probe_exists));
      d_desc.release();
      if (kind == LDB_JOIN_MARK) {
         ldb_dev_free(ctx, bitmap);
         *mark_out = mark;
         // all probe rows in input order
         ldb_rel* r = ldb_rel_new(ctx);
         r->n_rows = n;
         for (auto& s : probe->sides) {
            ldb_rel_side ns{s.table, nullptr, false, s.may_null};
            if (s.rowids) {
               LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, 4 * (size_t) (n ? n : 1)));
               if (n) LDB_HIP(hipMemcpyAsync(ns.rowids, s.rowids, 4 * (size_t) n, hipMemcpyDeviceToDevice, ctx->stream));
               ns.owned = true;
            }
            r->sides.push_back(ns);
         }
         *out = r;
         return LDB_OK;
      }
      uint64_t total = 0;
      LDB_TRY(ldb_read_u64(ctx, counter, &total));
      uint32_t* sel;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &sel, 4 * (size_t) (total ? total : 1)));
      if (n_words && total) LDB_TRY(ldb_bitmap_compact(ctx, bitmap, n_words, sel, total, nullptr, nullptr, nullptr));
      ldb_dev_free(ctx, bitmap);
      return ldb_rel_select(ctx, probe, sel, (int64_t) total, out);
   }

   uint32_t *op = nullptr, *ob = nullptr;
   uint64_t produced = 0;
   if (ht->unique) {
      // at most one match per probe row: dense match vector + ordered bitmap compaction
      uint32_t* match;
      uint64_t* bitmap = nullptr;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &match, 4 * (size_t) (n ? n : 1)));
      if (kind == LDB_JOIN_INNER) {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &bitmap, 8 * (size_t) (n_words ? n_words : 1)));
         h->has_bitmap = 1;
      }
      h->match = (uint64_t) match;
      h->bitmap = (uint64_t) bitmap;
      LdbDesc<DJoin> d_desc(ctx);
      LDB_TRY(d_desc.upload(h, sizeof(*h)));
      DJoin* d = d_desc.p;
      if (n && part_probe_ok(ht, part, n_resid) && probe->pending.empty()) LDB_TRY(launch_part_probe(ctx, ht, part, n, 1, match, bitmap, counter)); // partitions staged in LDS
      else if (n) LDB_TRY(launch_join(ctx, h, d, grid, "k_join_probe_unique", "k_join_probe_unique_spec", k_join_probe_unique));
      d_desc.release();
      if (getenv("LDB_DEBUG_COUNTS")) {
         uint64_t c[3];
         for (int k = 0; k < 3; k++) LDB_TRY(ldb_read_u64(ctx, counter + 2 + k, &c[k]));
         fprintf(stderr, "[debug counts] queued %llu key-bit hits %llu matches %llu\n", (unsigned long long) c[0], (unsigned long long) c[1], (unsigned long long) c[2]);
      }
      if (kind == LDB_JOIN_INNER) {
         LDB_TRY(ldb_read_u64(ctx, counter, &produced));
         if (produced == (uint64_t) n && n > 0 && ldb_option("join_all_match", 1) != 0) {
            // EVERY probe row found its partner (a foreign key probing its primary key: most joins of a TPC-H plan): the result has the probe
            // relation's rows in the probe relation's order, so its sides carry over as they are — shared, not copied — and match[] IS the
            // build side's selection.  No bitmap compaction, no composition of the probe sides (round 6; the reference never materialises
            // between the operators of a pipeline either, SubOpToControlFlow.cpp:1123-1202)
            if (ctx->trace_mode == 2 && n_words) // a replayed count: should a row be partner-less after all, its match[] word must not be garbage
               hipLaunchKernelGGL(k_unmatched_to_zero, dim3(ldb_grid_for(ctx, (int64_t) n_words, 256, 8)), dim3(256), 0, ctx->stream, (const uint64_t*) bitmap, match, (uint64_t) n);
            ldb_dev_free(ctx, bitmap);
            ldb_rel* r = ldb_rel_new(ctx);
            r->n_rows = n;
            for (auto& s : probe->sides) {
               ldb_rel_side ns{s.table, s.rowids, s.rowids != nullptr, s.may_null};
               if (s.rowids) ldb_dev_share(ctx, s.rowids);
               r->sides.push_back(ns);
            }
            std::vector<LdbComposeJob> bjobs;
            bool match_taken = false;
            for (auto& s : ht->build->sides) {
               ldb_rel_side ns{s.table, nullptr, true, s.may_null};
               if (!s.rowids && !match_taken) {
                  ns.rowids = match;
                  match_taken = true;
               } else {
                  LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, 4 * (size_t) n));
                  bjobs.push_back({(const uint32_t*) s.rowids, ns.rowids, 1});
               }
               r->sides.push_back(ns);
            }
            LDB_TRY(ldb_compose_rowids(ctx, match, match, bjobs.data(), (int) bjobs.size(), (uint64_t) n));
            if (!match_taken) ldb_dev_free(ctx, match);
            *out = r;
            return LDB_OK;
         }
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &op, 4 * (size_t) (produced ? produced : 1)));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &ob, 4 * (size_t) (produced ? produced : 1)));
         // matched probe rows in ascending order + the build row of each, in one launch
         if (n_words && produced) LDB_TRY(ldb_bitmap_compact(ctx, bitmap, n_words, op, produced, match, ob, nullptr));
         ldb_dev_free(ctx, bitmap);
         ldb_dev_free(ctx, match);
      } else { // LEFT_OUTER / SINGLE: exactly one output row per probe row
         produced = (uint64_t) n;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &op, 4 * (size_t) (n ? n : 1)));
         if (n) hipLaunchKernelGGL(k_iota_u32j, dim3(grid), dim3(256), 0, ctx->stream, op, (uint64_t) n);
         ob = match;
      }
   } else {
      // duplicated build keys: count per 64-row chunk → scan → emit (see join_probe_pairs_body)
      const int64_t n_chunks = (n + 63) / 64;
      uint32_t *chunk_cnt, *chunk_off;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &chunk_cnt, 4 * (size_t) (n_chunks ? n_chunks : 1)));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &chunk_off, 4 * (size_t) (n_chunks ? n_chunks : 1)));
      h->match = (uint64_t) chunk_cnt;
      LdbDesc<DJoin> d_desc(ctx);
      LDB_TRY(d_desc.upload(h, sizeof(*h)));
      DJoin* d = d_desc.p;
      if (n) {
         LDB_TRY(launch_join(ctx, h, d, grid, "k_join_probe_pairs_count", "k_join_probe_pairs_count_spec", k_join_probe_pairs_count));
         LDB_TRY(ldb_exclusive_scan_u32(ctx, chunk_cnt, chunk_off, n_chunks, nullptr));
         LDB_TRY(ldb_read_u64(ctx, counter, &produced));
      }
      d_desc.release();
      if (produced >= (uint64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_probe: %llu result rows exceed uint32 row ids", (unsigned long long) produced);
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &op, 4 * (size_t) (produced ? produced : 1)));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &ob, 4 * (size_t) (produced ? produced : 1)));
      if (produced) {
         if (ctx->trace_mode == 2) { // a replayed `produced` may exceed what the kernel really emits: no uninitialised row id behind the real tail
            LDB_HIP(hipMemsetAsync(op, 0, 4 * (size_t) produced, ctx->stream));
            LDB_HIP(hipMemsetAsync(ob, 0, 4 * (size_t) produced, ctx->stream));
         }
         h->match = (uint64_t) chunk_off;
         h->out_probe = (uint64_t) op;
         h->out_build = (uint64_t) ob;
         h->out_cap = produced;
         LDB_TRY(d_desc.upload(h, sizeof(*h)));
         d = d_desc.p;
         LDB_TRY(launch_join(ctx, h, d, grid, "k_join_probe_pairs", "k_join_probe_pairs_spec", k_join_probe_pairs));
         d_desc.release();
      }
      ldb_dev_free(ctx, chunk_cnt);
      ldb_dev_free(ctx, chunk_off);
   }
   if (ldb_option("debug_check", 0)) {
      LDB_TRY(debug_check_ids(ctx, "probe", op, produced, probe->n_rows, h));
      LDB_TRY(debug_check_ids(ctx, "build", ob, produced, ht->build->n_rows, h));
   }
   // result relation: probe sides composed with op, build sides composed with ob
   ldb_rel* r = ldb_rel_new(ctx);
   r->n_rows = (int64_t) produced;
   const int cg = ldb_grid_for(ctx, (int64_t) produced, 256, 8);
   // LEFT_OUTER / SINGLE pad the build sides of unmatched probe rows with LDB_NULL_ROW
   const bool pads = kind == LDB_JOIN_LEFT_OUTER || kind == LDB_JOIN_SINGLE;
   std::vector<LdbComposeJob> jobs; // every side that needs its own vector: composed in ONE launch below
   bool free_op = false, free_ob = false;
   auto add_sides = [&](ldb_rel* src, uint32_t* sel, int which, bool sel_may_null, bool* free_sel) -> int32_t {
      bool sel_taken = false; // the first identity side IS the selection vector: hand it over, no copy
      for (auto& s : src->sides) {
         ldb_rel_side ns{s.table, nullptr, true, s.may_null || sel_may_null};
         if (!s.rowids && !sel_taken) {
            ns.rowids = sel;
            sel_taken = true;
         } else {
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, 4 * (size_t) (produced ? produced : 1)));
            jobs.push_back({(const uint32_t*) s.rowids, ns.rowids, which});
         }
         r->sides.push_back(ns);
      }
      *free_sel = !sel_taken;
      return LDB_OK;
   };
   (void) cg;
   LDB_TRY(add_sides(probe, op, 0, false, &free_op));
   LDB_TRY(add_sides(ht->build, ob, 1, pads, &free_ob));
   LDB_TRY(ldb_compose_rowids(ctx, op, ob, jobs.data(), (int) jobs.size(), produced));
   if (free_op) ldb_dev_free(ctx, op);
   if (free_ob) ldb_dev_free(ctx, ob);
   *out = r;
   return LDB_OK;
}

// ---------------------------------------------------------------- nested-loop join
// A join WITHOUT key equality (translateNLJ, reference RelAlgToSubOp.cpp:948-1033: the right side is materialised into a
// buffer, every left tuple scans it and the join predicate filters the combinations).  Here: both sides get a constant key
// column, the build side becomes ONE chain of a chained table, and the probe walks that chain evaluating the residual
// conjuncts (column-vs-column comparisons between the sides, <= LDB_MAX_RESID) — the same walk every hash join with a residual
// predicate does, so every join kind is available.  O(|probe| x |build|) by nature: meant for small build sides (band joins
// against dimension tables, cross products of scalar subqueries).
static void strip_table_sides(ldb_ctx* ctx, ldb_rel* r, const ldb_table* t) {
   for (size_t k = 0; k < r->sides.size();) {
      if (r->sides[k].table == t) {
         if (r->sides[k].owned) ldb_dev_free(ctx, r->sides[k].rowids);
         r->sides.erase(r->sides.begin() + (long) k);
      } else {
         k++;
      }
   }
}
extern "C" int32_t ldb_gpu_join_nl(ldb_ctx* ctx, ldb_rel* probe, ldb_rel* build, int32_t kind, const ldb_join_residual* resid, int32_t n_resid, ldb_rel** out, ldb_table** mark_out) {
   if (!ctx || !probe || !build || !out) LDB_FAIL(LDB_ERR_INVALID, "join_nl: NULL argument");
   if (n_resid < 0 || n_resid > LDB_MAX_RESID || (n_resid && !resid)) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_nl: %d predicate conjuncts (0..%d column-vs-column comparisons)", n_resid, LDB_MAX_RESID);
   if (kind == LDB_JOIN_RIGHT_OUTER || kind == LDB_JOIN_FULL_OUTER) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_nl: right / full outer nested-loop joins");
   LDB_TRY(ldb_rel_force(ctx, probe));
   LDB_TRY(ldb_rel_force(ctx, build));
   if (probe->sides.size() + 1 > LDB_MAX_SIDES || build->sides.size() + 1 > LDB_MAX_SIDES) LDB_FAIL(LDB_ERR_UNSUPPORTED, "join_nl: more than %d sides (materialize first)", LDB_MAX_SIDES - 1);
   struct Tmp {
      ldb_ctx* ctx;
      ldb_table *kp = nullptr, *kb = nullptr;
      ldb_rel *p2 = nullptr, *b2 = nullptr;
      ldb_hashtable* ht = nullptr;
      ~Tmp() {
         if (ht) ldb_gpu_hashtable_release(ctx, ht);
         if (p2) ldb_gpu_rel_release(ctx, p2);
         if (b2) ldb_gpu_rel_release(ctx, b2);
         if (kp) ldb_gpu_table_release(ctx, kp);
         if (kb) ldb_gpu_table_release(ctx, kb);
      }
   } t{ctx};
   const ldb_coltype kt = {LDB_T_INT32, 0, 0, 0};
   const char* nm = "nl_key";
   LDB_TRY(ldb_gpu_table_alloc(ctx, "nl_probe_key", 1, &kt, &nm, probe->n_rows, nullptr, 0, &t.kp));
   LDB_TRY(ldb_gpu_table_alloc(ctx, "nl_build_key", 1, &kt, &nm, build->n_rows, nullptr, 0, &t.kb));
   if (probe->n_rows) LDB_HIP(hipMemsetAsync(t.kp->cols[0].values, 0, 4 * (size_t) probe->n_rows, ctx->stream));
   if (build->n_rows) LDB_HIP(hipMemsetAsync(t.kb->cols[0].values, 0, 4 * (size_t) build->n_rows, ctx->stream));
   LDB_TRY(ldb_gpu_rel_zip(ctx, probe, t.kp, &t.p2));
   LDB_TRY(ldb_gpu_rel_zip(ctx, build, t.kb, &t.b2));
   const ldb_colref bkey = {(int32_t) build->sides.size(), 0}, pkey = {(int32_t) probe->sides.size(), 0};
   LDB_TRY(ldb_gpu_join_build(ctx, t.b2, &bkey, 1, 0, &t.ht)); // one key value, not unique: one chain holding every build row
   ldb_rel* r = nullptr;
   LDB_TRY(ldb_gpu_join_probe_residual(ctx, t.ht, t.p2, &pkey, 1, kind, resid, n_resid, &r, mark_out));
   strip_table_sides(ctx, r, t.kp); // the constant key columns are not part of the result
   strip_table_sides(ctx, r, t.kb);
   *out = r;
   return LDB_OK;
}

