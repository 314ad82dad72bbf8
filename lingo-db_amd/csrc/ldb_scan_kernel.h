// ldb_scan_kernel.h — device code of the columnar scan + pushed-down predicate kernels.
// Compiled ahead of time (generic) and, for large inputs, at run time with the predicate list as
// a compile-time constant (ldb_jit.hip) — same source, see ldb_gb_kernel.h for the rationale.
// Replaces (reference): ScanBatchesTask::unitRun (src/runtime/storage/LingoDBTable.cpp:382-407) and
// Restrictions::applyFilters + Filter impls (src/runtime/storage/Restrictions.cpp:67-390).
#pragma once
#include "ldb_device.h"

#define SCAN_BLOCK 256
#define SCAN_WORDS_PER_BLOCK 256 // 64-bit words → 16384 rows per block

struct DScan {
   uint64_t n_rows; // run-time
   int32_t n_preds;
   int32_t pad;
   DPred preds[LDB_MAX_PREDS];
};

__device__ __forceinline__ bool d_eval_conj(const DScan& m, const DScan* __restrict__ d, uint64_t i) {
   bool pass = true;
   const int np = m.n_preds;
   LDB_UNROLL
   for (int p = 0; p < np; p++) {
      if (pass) pass = d_eval_pred(PV(m.preds[p], d->preds[p]), i);
   }
   return pass;
}

// the conjunction over U rows of one thread, predicate-major (ldb_device.h d_eval_conj_batch)
// (both scan kernels give every wave 64 consecutive rows per batch slot, so string conjuncts may
// stage through LDS: one LDS_STR_STAGE region per wave of the 256-thread block)
template <int U>
__device__ __forceinline__ void d_eval_conj_batch(const DScan& m, const DScan* __restrict__ d, const uint64_t (&rows)[U], bool (&pass)[U]) {
   __shared__ __attribute__((aligned(8))) uint8_t str_stage[SCAN_BLOCK / LDB_WAVE][LDS_STR_BYTES]; // strings, match bitmaps, pattern
   d_eval_conj_batch<U>(m.preds, d->preds, m.n_preds, rows, pass, d->n_rows, str_stage[threadIdx.x >> 6]);
}

// One block = 16384 consecutive rows; each wave handles 64 of the block's 256 bitmap words,
// 4 words (256 rows) per iteration: the four rows' column loads are independent and issue back to
// back before the first ballot (memory-level parallelism for an HBM-bound scan).
__device__ __forceinline__ void scan_bitmap_body(const DScan& m, const DScan* __restrict__ d, uint64_t* __restrict__ bitmap,
                                                 uint32_t* __restrict__ block_counts, uint32_t* s_cnt) {
   const uint64_t n = d->n_rows;
   const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   // (round 6: a short input is cut finer — gridDim.y = 2 … 16 workgroups share a zone's 256 words, so that 1 M rows are 976 workgroups instead of
   // 61 whose waves each walk 64 words one after the other: Q16's LIKE over the supplier comments took 1.1 ms that way)
   const uint32_t wpb = SCAN_WORDS_PER_BLOCK / gridDim.y;
   const uint64_t word0 = (uint64_t) blockIdx.x * SCAN_WORDS_PER_BLOCK + (uint64_t) blockIdx.y * wpb;
   const uint32_t block_slot = blockIdx.x * gridDim.y + blockIdx.y;
   // zone maps: this block's 16 384 rows are exactly one zone; if a conjunct's zone cannot match, the block is all zeros
   static_assert(SCAN_WORDS_PER_BLOCK * 64 == LDB_ZONE_ROWS, "one scan block per zone");
   {
      bool excluded = false;
      LDB_UNROLL
      for (int p = 0; p < LDB_MAX_PREDS; p++)
         if (p < m.n_preds && m.preds[p].zmin && d_pred_is_simple(m.preds[p]))
            excluded = excluded || !d_zone_may_pass(m.preds[p].op, gptr<int64_t>(d->preds[p].zmin)[blockIdx.x], gptr<int64_t>(d->preds[p].zmax)[blockIdx.x], (int64_t) m.preds[p].lo);
      if (excluded) { // (block-uniform)
         for (uint32_t w = threadIdx.x; w < wpb; w += SCAN_BLOCK)
            if ((word0 + w) * 64 < n) bitmap[word0 + w] = 0;
         if (threadIdx.x == 0) block_counts[block_slot] = 0;
         return;
      }
   }
   uint32_t cnt = 0;
   constexpr uint32_t WPW = SCAN_BLOCK / LDB_WAVE; // waves per block
   auto words = [&](const uint32_t bound) __attribute__((always_inline)) { // (bound is a multiple of 4 * WPW = 16)
      for (uint32_t w = wave; w < bound; w += 4 * WPW) {
         bool pass[4];
         uint64_t rows[4];
#pragma unroll
         for (int u = 0; u < 4; u++) {
            rows[u] = (word0 + w + u * WPW) * 64 + lane;
            pass[u] = rows[u] < n;
         }
         d_eval_conj_batch<4>(m, d, rows, pass);
#pragma unroll
         for (int u = 0; u < 4; u++) {
            uint64_t mask = __ballot(pass[u]);
            if (lane == 0 && (word0 + w + u * WPW) * 64 < n) bitmap[word0 + w + u * WPW] = mask;
            cnt += (uint32_t) __popcll(mask);
         }
      }
   };
   // the whole-zone form keeps its compile-time trip count (with a run-time bound the long scans lost a fifth: Q13's LIKE 3.3 → 4.1 ms).  A
   // specialised kernel is compiled for ONE of the two forms — the host says which in the metadata's n_rows (0 = whole zones, 1 = split; the row
   // count itself is read from the run-time descriptor) — so that the conjunction is not generated twice
#ifdef LDB_JIT_SPECIALIZED
   const bool whole = m.n_rows == 0;
#else
   const bool whole = gridDim.y == 1;
#endif
   if (whole) words(SCAN_WORDS_PER_BLOCK);
   else words(wpb);
   if (lane == 0) s_cnt[wave] = cnt;
   __syncthreads();
   if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int k = 0; k < SCAN_BLOCK / LDB_WAVE; k++) t += s_cnt[k];
      block_counts[block_slot] = t;
   }
}

// count-only variant: grid-stride, 4 rows in flight per thread, no bitmap
__device__ __forceinline__ void scan_count_body(const DScan& m, const DScan* __restrict__ d, unsigned long long* __restrict__ total) {
   const uint64_t n = d->n_rows;
   const uint64_t tid = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x, nth = (uint64_t) gridDim.x * blockDim.x;
   uint32_t cnt = 0;
   for (uint64_t i0 = tid; i0 < n; i0 += 4 * nth) {
      bool pass[4];
      uint64_t rows[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         rows[u] = i0 + (uint64_t) u * nth;
         pass[u] = rows[u] < n;
      }
      d_eval_conj_batch<4>(m, d, rows, pass);
#pragma unroll
      for (int u = 0; u < 4; u++) cnt += pass[u] ? 1u : 0u;
   }
   for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
   if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(total, (unsigned long long) cnt);
}

// ---- disjunctive normal form: OR of up to DNF_MAX_CLAUSES conjunctions over one pass (ldb_gpu_scan_filter_dnf)
#define DNF_MAX_CLAUSES 4
#define DNF_MAX_PREDS 24
struct DScanDnf {
   uint64_t n_rows; // run-time
   int32_t n_clauses;
   int32_t clause_end[DNF_MAX_CLAUSES]; // preds [clause_end[c-1], clause_end[c]) form clause c
   int32_t pad;
   DPred preds[DNF_MAX_PREDS];
};
// one row: clause after clause, a conjunct is evaluated only while its clause can still pass and no earlier clause has.
// Written as ONE loop over the conjuncts so that the specialised build unrolls it into straight-line code (p and the
// clause index are compile-time constants there).
__device__ __forceinline__ bool d_eval_dnf(const DScanDnf& m, const DScanDnf* __restrict__ d, uint64_t i) {
   bool pass = false, cp = true;
   int c = 0;
   const int total = m.clause_end[m.n_clauses - 1];
   LDB_UNROLL
   for (int p = 0; p < DNF_MAX_PREDS; p++) {
      if (p < total) {
         if (!pass && cp) cp = d_eval_pred(PV(m.preds[p], d->preds[p]), i);
         if (p + 1 == m.clause_end[c]) {
            pass = pass || cp;
            cp = true;
            c++;
         }
      }
   }
   return pass;
}
__device__ __forceinline__ void scan_bitmap_dnf_body(const DScanDnf& m, const DScanDnf* __restrict__ d, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ block_counts,
                                                     uint32_t* s_cnt) {
   const uint64_t n = d->n_rows;
   const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const uint32_t wpb = SCAN_WORDS_PER_BLOCK / gridDim.y; // (see scan_bitmap_body)
   const uint64_t word0 = (uint64_t) blockIdx.x * SCAN_WORDS_PER_BLOCK + (uint64_t) blockIdx.y * wpb;
   uint32_t cnt = 0;
   auto words = [&](const uint32_t bound) __attribute__((always_inline)) {
      for (uint32_t w = wave; w < bound; w += SCAN_BLOCK / LDB_WAVE) {
         const uint64_t i = (word0 + w) * 64 + lane;
         const bool pass = i < n && d_eval_dnf(m, d, i);
         const uint64_t mask = __ballot(pass);
         if (lane == 0 && (word0 + w) * 64 < n) bitmap[word0 + w] = mask;
         cnt += (uint32_t) __popcll(mask);
      }
   };
#ifdef LDB_JIT_SPECIALIZED
   const bool whole = m.n_rows == 0; // (as in scan_bitmap_body)
#else
   const bool whole = gridDim.y == 1;
#endif
   if (whole) words(SCAN_WORDS_PER_BLOCK);
   else words(wpb);
   if (lane == 0) s_cnt[wave] = cnt;
   __syncthreads();
   if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int k = 0; k < SCAN_BLOCK / LDB_WAVE; k++) t += s_cnt[k];
      block_counts[blockIdx.x * gridDim.y + blockIdx.y] = t;
   }
}
