// ldb_sort.hip — sort, top-k and hash-radix partitioning (all built on one stable counting sort).
// Replaces (reference): GrowingBuffer::sort / parallelSort (src/runtime/GrowingBuffer.cpp:54-78,
// src/runtime/Sorting.cpp:343-393) with the comparator of db.sort_compare
// (src/compiler/Conversion/DBToStd/LowerToStd.cpp:1046-1064) and Heap (src/runtime/Heap.cpp:8-72).
//
// MI355X design: the CPU sorts row POINTERS through an indirect comparator call.  Here every row
// gets an order-preserving fixed-width byte key (sign-flipped big-endian integers, zero-padded
// strings + length, bytes inverted for DESC) and a row permutation is LSD-radix-sorted on it,
// 4 bits per pass with a stable counting sort (per-thread register histograms → one device scan
// of the [digit][thread] matrix → ordered scatter).  Equal keys keep input order.
// TPC-H sorts are post-aggregation (≤ ~1 M rows); the pass structure favours simplicity.
#include "ldb_internal.h"
#include "ldb_keys.h"
#include <algorithm>
#include <memory>

#define CS_CHUNK 16 // rows per thread
#define CS_BLOCK 256

// ---------------------------------------------------------------- stable counting sort on 4-bit digits
// digit of perm_in[j] = (keys[perm_in[j] * words + word] >> shift) & 15
__global__ void k_cs_count(const uint64_t* __restrict__ keys, int words, int word, int shift, const uint32_t* __restrict__ perm_in, uint64_t n,
                           uint32_t* __restrict__ counts, uint64_t n_threads) {
   uint64_t t = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x;
   if (t >= n_threads) return;
   uint32_t c[16];
#pragma unroll
   for (int k = 0; k < 16; k++) c[k] = 0;
   uint64_t b = t * CS_CHUNK, e = b + CS_CHUNK < n ? b + CS_CHUNK : n;
   for (uint64_t j = b; j < e; j++) {
      uint32_t dg = (uint32_t) (keys[(uint64_t) perm_in[j] * words + word] >> shift) & 15u;
#pragma unroll
      for (int k = 0; k < 16; k++) c[k] += (dg == (uint32_t) k);
   }
#pragma unroll
   for (int k = 0; k < 16; k++) counts[(uint64_t) k * n_threads + t] = c[k];
}
__global__ void k_cs_scatter(const uint64_t* __restrict__ keys, int words, int word, int shift, const uint32_t* __restrict__ perm_in,
                             uint32_t* __restrict__ perm_out, uint64_t n, const uint32_t* __restrict__ offsets, uint64_t n_threads) {
   uint64_t t = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x;
   if (t >= n_threads) return;
   uint32_t o[16];
#pragma unroll
   for (int k = 0; k < 16; k++) o[k] = offsets[(uint64_t) k * n_threads + t];
   uint64_t b = t * CS_CHUNK, e = b + CS_CHUNK < n ? b + CS_CHUNK : n;
   for (uint64_t j = b; j < e; j++) {
      uint32_t p = perm_in[j];
      uint32_t dg = (uint32_t) (keys[(uint64_t) p * words + word] >> shift) & 15u;
      uint32_t dst = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
         if (dg == (uint32_t) k) {
            dst = o[k];
            o[k]++;
         }
      }
      perm_out[dst] = p;
   }
}

// ---------------------------------------------------------------- key normalisation
#define SORT_MAX_SPECS 8
struct DSortSpec {
   DCol col;
   int32_t descending;
   int32_t byte_off; // offset of this key inside the record
   int32_t nbytes; // bytes this key occupies
   int32_t str_pad; // utf8: padded string bytes (nbytes = str_pad + 4)
};
struct DSort {
   uint64_t n_rows;
   int32_t n_specs;
   int32_t words; // 64-bit words per record
   DSortSpec specs[SORT_MAX_SPECS];
};

// record bytes are big-endian inside big-endian words: byte b of the record is bits
// [63-8*(b%8) ..] of word b/8, so word-wise unsigned compare == bytewise compare.
__device__ __forceinline__ void d_put_byte(uint64_t* rec, int b, uint8_t v) { rec[b >> 3] |= (uint64_t) v << (56 - 8 * (b & 7)); }

__global__ void k_sort_keys(const DSort* __restrict__ d, uint64_t* __restrict__ keys) {
   const uint64_t n = d->n_rows;
   const int words = d->words;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint64_t* rec = keys + i * words;
      for (int w = 0; w < words; w++) rec[w] = 0;
      for (int s = 0; s < d->n_specs; s++) {
         const DSortSpec& sp = d->specs[s];
         const uint8_t inv = sp.descending ? 0xFF : 0x00;
         uint32_t row = d_phys_row(sp.col, i);
         int b = sp.byte_off;
         if (sp.col.type == LDB_T_UTF8) {
            uint32_t len;
            const uint8_t* p = d_load_str(sp.col, row, &len);
            for (int k = 0; k < sp.str_pad; k++) d_put_byte(rec, b + k, (uint8_t) (((uint32_t) k < len ? p[k] : 0) ^ inv));
            for (int k = 0; k < 4; k++) d_put_byte(rec, b + sp.str_pad + k, (uint8_t) ((len >> (24 - 8 * k)) ^ inv));
         } else if (sp.col.type == LDB_T_FLOAT64 || sp.col.type == LDB_T_FLOAT32) {
            uint64_t bits = (uint64_t) __double_as_longlong(d_load_f64(sp.col, row));
            bits = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
            for (int k = 0; k < 8; k++) d_put_byte(rec, b + k, (uint8_t) ((bits >> (56 - 8 * k)) ^ inv));
         } else if (sp.nbytes == 16) {
            u128 v = (u128) d_load_i128(sp.col, row) ^ ((u128) 1 << 127);
            for (int k = 0; k < 16; k++) d_put_byte(rec, b + k, (uint8_t) ((uint8_t) (v >> (120 - 8 * k)) ^ inv));
         } else {
            uint64_t v = (uint64_t) d_load_i64(sp.col, row) ^ 0x8000000000000000ull;
            for (int k = 0; k < 8; k++) d_put_byte(rec, b + k, (uint8_t) ((v >> (56 - 8 * k)) ^ inv));
         }
      }
   }
}

__global__ void k_str_maxlen(DCol col, uint64_t n, unsigned long long* out) {
   unsigned long long m = 0;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint32_t len;
      d_load_str(col, d_phys_row(col, i), &len);
      if (len > m) m = len;
   }
   if (m) atomicMax(out, m);
}
__global__ void k_iota(uint32_t* out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = (uint32_t) i;
}
__global__ void k_all_valid(DCol col, uint64_t n, unsigned int* flag) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      if (!d_valid(col, d_phys_row(col, i))) atomicOr(flag, 1u);
}

int32_t ldb_rel_select(ldb_ctx* ctx, ldb_rel* in, uint32_t* sel, int64_t n_sel, ldb_rel** out);
int32_t ldb_gather_column(ldb_ctx* ctx, const ldb_rel* r, ldb_colref ref, ldb_column* out);

// stable LSD sort of `perm` (n entries) by `words`-word records, digits [bit_lo, bit_hi) of each word
// `varmask` (host, one word per record word, may be NULL): bits that differ between some two
// records — a digit whose bits are all constant is skipped (an LSD pass over it is the identity).
static int32_t radix_sort_perm(ldb_ctx* ctx, const uint64_t* keys, int words, uint32_t** perm_io, uint64_t n, int first_word, int last_word, int bits_lo, int bits_hi,
                               const uint64_t* varmask) {
   if (n < 2) return LDB_OK;
   const uint64_t n_threads = (n + CS_CHUNK - 1) / CS_CHUNK;
   const int grid = (int) ((n_threads + CS_BLOCK - 1) / CS_BLOCK);
   uint32_t *counts, *offsets, *perm_b;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &counts, 4 * 16 * (size_t) n_threads));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &offsets, 4 * 16 * (size_t) n_threads));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &perm_b, 4 * (size_t) n));
   uint32_t* a = *perm_io;
   uint32_t* b = perm_b;
   for (int w = last_word; w >= first_word; w--) {
      for (int shift = bits_lo; shift < bits_hi; shift += 4) {
         if (varmask && ((varmask[w] >> shift) & 15ull) == 0) continue;
         hipLaunchKernelGGL(k_cs_count, dim3(grid), dim3(CS_BLOCK), 0, ctx->stream, keys, words, w, shift, (const uint32_t*) a, n, counts, n_threads);
         LDB_TRY(ldb_exclusive_scan_u32(ctx, counts, offsets, (int64_t) (16 * n_threads), nullptr));
         hipLaunchKernelGGL(k_cs_scatter, dim3(grid), dim3(CS_BLOCK), 0, ctx->stream, keys, words, w, shift, (const uint32_t*) a, b, n, (const uint32_t*) offsets, n_threads);
         std::swap(a, b);
      }
   }
   LDB_HIP(hipGetLastError());
   *perm_io = a;
   ldb_dev_free(ctx, b);
   ldb_dev_free(ctx, counts);
   ldb_dev_free(ctx, offsets);
   return LDB_OK;
}

// ---------------------------------------------------------------- small inputs: one workgroup, bitonic in LDS
// TPC-H's ORDER BYs are post-aggregation (4 .. a few thousand rows): a radix sort there is ~100
// launches of nothing.  One workgroup sorts up to SS_MAX row numbers by full record compare
// (ties: lower row number first = stable for an ascending input permutation).
#define SS_MAX 4096
#define SS_BLOCK 1024
__device__ __forceinline__ bool d_rec_greater(const uint64_t* __restrict__ keys, int words, uint32_t a, uint32_t b) {
   if (a == 0xFFFFFFFFu || b == 0xFFFFFFFFu) return a == 0xFFFFFFFFu && b != 0xFFFFFFFFu; // padding sorts last
   for (int w = 0; w < words; w++) {
      uint64_t x = keys[(uint64_t) a * words + w], y = keys[(uint64_t) b * words + w];
      if (x != y) return x > y;
   }
   return a > b;
}
__global__ __launch_bounds__(SS_BLOCK) void k_small_sort(const uint64_t* __restrict__ keys, int words, uint32_t* __restrict__ perm, uint32_t n) {
   __shared__ uint32_t idx[SS_MAX];
   uint32_t m = 1;
   while (m < n) m <<= 1;
   for (uint32_t i = threadIdx.x; i < m; i += SS_BLOCK) idx[i] = i < n ? perm[i] : 0xFFFFFFFFu;
   __syncthreads();
   for (uint32_t k = 2; k <= m; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
         for (uint32_t i = threadIdx.x; i < m; i += SS_BLOCK) {
            uint32_t l = i ^ j;
            if (l > i) {
               uint32_t a = idx[i], b = idx[l];
               bool asc = (i & k) == 0;
               if (d_rec_greater(keys, words, a, b) == asc) {
                  idx[i] = b;
                  idx[l] = a;
               }
            }
         }
         __syncthreads();
      }
   }
   for (uint32_t i = threadIdx.x; i < n; i += SS_BLOCK) perm[i] = idx[i];
}

// ---------------------------------------------------------------- which key bits vary at all
// OR and AND of every record word: bits where they agree are constant over the whole input and
// carry no order information (a decimal(33,4) revenue occupies 16 key bytes of which ~5 vary).
__global__ void k_key_bits(const uint64_t* __restrict__ keys, int words, uint64_t n, unsigned long long* __restrict__ or_and) {
   const int w = blockIdx.y;
   uint64_t o = 0, a = ~0ull;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint64_t v = keys[i * words + w];
      o |= v;
      a &= v;
   }
   for (int off = 32; off > 0; off >>= 1) {
      o |= (uint64_t) __shfl_xor((unsigned long long) o, off);
      a &= (uint64_t) __shfl_xor((unsigned long long) a, off);
   }
   if ((threadIdx.x & 63) == 0) {
      atomicOr(&or_and[w], (unsigned long long) o);
      atomicAnd(&or_and[words + w], (unsigned long long) a);
   }
}

// ---------------------------------------------------------------- top-k: radix select over the leading varying key bytes
// Select state (device): prefix value / mask per record word for the first SEL_WORDS words, and the
// rank still to find among the records that match the prefix so far.
#define SEL_WORDS 4
struct SelState {
   unsigned long long pval[SEL_WORDS];
   unsigned long long pmask[SEL_WORDS];
   unsigned long long rank;
   unsigned long long rank0; // k - 1: the rank asked for
   unsigned long long cand; // after a pass: rows whose selected bytes are <= the k-th row's = what the passes so far leave to sort
};
__device__ __forceinline__ bool d_sel_match(const uint64_t* __restrict__ rec, const SelState& st, int upto) {
   bool m = true;
   for (int w = 0; w <= upto; w++) m = m && ((rec[w] & st.pmask[w]) == st.pval[w]);
   return m;
}
// histogram of byte (w, shift) over the records matching the prefix chosen so far
__global__ void k_sel_hist(const uint64_t* __restrict__ keys, int words, uint64_t n, int w, int shift, const SelState* __restrict__ state, uint32_t* __restrict__ hist) {
   __shared__ uint32_t lh[256];
   for (int k = threadIdx.x; k < 256; k += blockDim.x) lh[k] = 0;
   __syncthreads();
   const SelState& st = *state; // wave-uniform: scalar loads
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint64_t* rec = keys + i * words;
      if (d_sel_match(rec, st, w)) atomicAdd(&lh[(rec[w] >> shift) & 255u], 1u);
   }
   __syncthreads();
   for (int k = threadIdx.x; k < 256; k += blockDim.x)
      if (lh[k]) atomicAdd(&hist[k], lh[k]);
}
__global__ void k_sel_pick(SelState* __restrict__ state, uint32_t* __restrict__ hist, int w, int shift) {
   if (threadIdx.x == 0) {
      unsigned long long rank = state->rank, cum = 0;
      int dsel = 255;
      for (int dg = 0; dg < 256; dg++) {
         if (cum + hist[dg] > rank) {
            dsel = dg;
            break;
         }
         cum += hist[dg];
      }
      state->pval[w] |= (unsigned long long) dsel << shift;
      state->pmask[w] |= 255ull << shift;
      state->rank = rank - cum;
      state->cand = (state->rank0 - (rank - cum)) + hist[dsel]; // rows below the chosen prefix + rows sharing it
   }
   __syncthreads();
   for (int k = threadIdx.x; k < 256; k += blockDim.x) hist[k] = 0;
}
// candidates = records whose selected bytes are lexicographically <= the k-th smallest's; one
// bitmap word + popcount per wave
__global__ void k_sel_flags(const uint64_t* __restrict__ keys, int words, uint64_t n, int sel_words, const SelState* __restrict__ state, uint64_t* __restrict__ bitmap,
                            uint32_t* __restrict__ pop) {
   const SelState& st = *state; // wave-uniform: scalar loads
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   for (uint64_t bw = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6; bw < n_words; bw += ((uint64_t) gridDim.x * blockDim.x) >> 6) {
      uint64_t i = bw * 64 + lane;
      bool keep = false;
      if (i < n) {
         keep = true; // equal on every selected byte → candidate
         for (int w = 0; w < sel_words; w++) {
            uint64_t x = keys[i * words + w] & st.pmask[w];
            if (x != st.pval[w]) {
               keep = x < st.pval[w];
               break;
            }
         }
      }
      uint64_t mask = __ballot(keep);
      if (lane == 0) {
         bitmap[bw] = mask;
         pop[bw] = (uint32_t) __popcll(mask);
      }
   }
}
__global__ void k_sel_expand(const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ off, uint32_t* __restrict__ out, uint64_t n_words, uint64_t cap) {
   const uint32_t lane = threadIdx.x & 63;
   for (uint64_t w = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6; w < n_words; w += ((uint64_t) gridDim.x * blockDim.x) >> 6) {
      uint64_t mask = bitmap[w];
      const uint64_t at = (uint64_t) off[w] + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
      if (((mask >> lane) & 1) && at < cap) out[at] = (uint32_t) (w * 64 + lane);
   }
}

// the k smallest records (all of them when k >= n), sorted: *perm_out holds >= min(k, n) row numbers
static int32_t sort_records(ldb_ctx* ctx, const uint64_t* keys, int words, uint64_t n, uint64_t k, uint32_t** perm_out) {
   uint32_t* perm;
   const int grid = ldb_grid_for(ctx, (int64_t) n, 256, 8);
   uint64_t m = n; // rows to sort
   std::vector<uint64_t> varmask((size_t) words, ~0ull);
   if (n > SS_MAX) {
      unsigned long long* d_bits;
      std::vector<unsigned long long> init((size_t) (2 * words), 0ull), bits((size_t) (2 * words));
      for (int w = 0; w < words; w++) init[(size_t) (words + w)] = ~0ull;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &d_bits, 16 * (size_t) words));
      LDB_TRY(ldb_h2d_small(ctx, d_bits, init.data(), 16 * (size_t) words)); // (pinned staging, never a pageable source inside a plan)
      // few blocks: every wave ends in two same-address atomics (≈10 ns each when contended)
      hipLaunchKernelGGL(k_key_bits, dim3(std::min(grid, 64), words), dim3(256), 0, ctx->stream, keys, words, n, d_bits);
      LDB_TRY(LDB_READBACK(ctx, bits.data(), d_bits, 16 * (size_t) words));
      ldb_dev_free(ctx, d_bits);
      for (int w = 0; w < words; w++) varmask[(size_t) w] = bits[(size_t) w] ^ bits[(size_t) (words + w)];
   }
   if (k < n && n > SS_MAX) {
      // radix select over the first (up to 8) varying bytes of the first SEL_WORDS words, most
      // significant first; decisions stay on the device (no host round trip per pass)
      struct Pass {
         int w, shift;
      } passes[8];
      int n_pass = 0, sel_words = 0;
      for (int w = 0; w < words && w < SEL_WORDS && n_pass < 8; w++)
         for (int shift = 56; shift >= 0 && n_pass < 8; shift -= 8)
            if ((varmask[(size_t) w] >> shift) & 255ull) {
               passes[n_pass++] = {w, shift};
               sel_words = w + 1;
            }
      SelState h_state;
      memset(&h_state, 0, sizeof(h_state));
      h_state.rank = h_state.rank0 = k ? k - 1 : 0;
      h_state.cand = n;
      LdbDesc<SelState> state_desc(ctx);
      uint32_t* hist;
      LDB_TRY(state_desc.upload(&h_state, sizeof(h_state), false)); // (k_sel_pick updates it)
      SelState* state = state_desc.p;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &hist, 4 * 256));
      LDB_HIP(hipMemsetAsync(hist, 0, 4 * 256, ctx->stream));
      // Passes stop as soon as what they leave — the rows below the k-th prefix + the rows sharing it — fits the one-workgroup sort (round 6: all
      // eight passes, sixteen launches, ran whatever the data; two or three bytes decide most of TPC-H's top-k inputs).  The count is an ordinary
      // read-back: a recording execution waits for it after every pass, a replayed plan launches exactly the recorded number of passes.  (The
      // varying-BIT count of a byte does not predict this: the bytes of a sign-flipped decimal column with both signs all vary, and all say the same.)
      const bool early = ldb_option("topk_short_select", 1) != 0;
      for (int p = 0; p < n_pass; p++) {
         hipLaunchKernelGGL(k_sel_hist, dim3(grid), dim3(256), 0, ctx->stream, keys, words, n, passes[p].w, passes[p].shift, (const SelState*) state, hist);
         hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(256), 0, ctx->stream, state, hist, passes[p].w, passes[p].shift);
         if (early && p + 1 < n_pass) {
            uint64_t cand = 0;
            LDB_TRY(ldb_read_u64(ctx, (const uint64_t*) &state->cand, &cand));
            if (cand <= SS_MAX) {
               sel_words = passes[p].w + 1;
               break;
            }
         }
      }
      const uint64_t n_words = (n + 63) / 64;
      uint64_t* bitmap;
      uint32_t *pop, *off;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &bitmap, 8 * (size_t) n_words));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &pop, 4 * (size_t) n_words));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &off, 4 * (size_t) n_words));
      hipLaunchKernelGGL(k_sel_flags, dim3(grid), dim3(256), 0, ctx->stream, keys, words, n, sel_words, (const SelState*) state, bitmap, pop);
      uint64_t* d_total;
      LDB_TRY(ldb_counters(ctx, 1, &d_total));
      LDB_TRY(ldb_exclusive_scan_u32(ctx, pop, off, (int64_t) n_words, d_total));
      LDB_TRY(ldb_read_u64(ctx, d_total, &m));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &perm, 4 * (size_t) (m ? m : 1)));
      hipLaunchKernelGGL(k_sel_expand, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t*) bitmap, (const uint32_t*) off, perm, n_words, (uint64_t) (m ? m : 1));
      LDB_HIP(hipGetLastError());
      state_desc.release();
      ldb_dev_free(ctx, hist);
      ldb_dev_free(ctx, bitmap);
      ldb_dev_free(ctx, pop);
      ldb_dev_free(ctx, off);
   } else {
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &perm, 4 * (size_t) (n ? n : 1)));
      if (n) hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, ctx->stream, perm, n);
   }
   if (m > 1 && m <= SS_MAX) hipLaunchKernelGGL(k_small_sort, dim3(1), dim3(SS_BLOCK), 0, ctx->stream, keys, words, perm, (uint32_t) m);
   else LDB_TRY(radix_sort_perm(ctx, keys, words, &perm, m, 0, words - 1, 0, 64, varmask.data()));
   LDB_HIP(hipGetLastError());
   *perm_out = perm;
   return LDB_OK;
}

static int32_t sort_perm(ldb_ctx* ctx, ldb_rel* in, const ldb_sort_spec* specs, int32_t n_specs, uint64_t k, uint32_t** perm_out) {
   if (n_specs < 1 || n_specs > SORT_MAX_SPECS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "sort: %d keys (1..%d supported)", n_specs, SORT_MAX_SPECS);
   const uint64_t n = (uint64_t) in->n_rows;
   auto hp = std::make_unique<DSort>();
   DSort* h = hp.get();
   memset(h, 0, sizeof(*h));
   h->n_rows = n;
   h->n_specs = n_specs;
   const int grid = ldb_grid_for(ctx, (int64_t) n, 256, 8);
   // pre-checks of all keys in one round trip: word 0 = "a key holds a NULL", word 1 + s = longest string of key s
   unsigned long long* d_chk = nullptr;
   bool any_chk = false;
   for (int s = 0; s < n_specs; s++) {
      DSortSpec& sp = h->specs[s];
      LDB_TRY(ldb_make_dcol_dict(in, specs[s].col, &sp.col)); // (an order-preserving dictionary: sorting the codes sorts the strings)
      sp.descending = specs[s].descending ? 1 : 0;
      const bool padded = sp.col.rowids && in->sides[(size_t) specs[s].col.side].may_null; // outer-join padding
      if (n && (sp.col.validity || padded || sp.col.type == LDB_T_UTF8)) {
         if (!any_chk) {
            LDB_TRY(ldb_counters(ctx, 1 + SORT_MAX_SPECS, (uint64_t**) &d_chk)); // zeroed arena words
            any_chk = true;
         }
         // db.sort_compare is only defined for non-nullable operands (LowerToStd.cpp:1050-1052)
         if (sp.col.validity || padded) hipLaunchKernelGGL(k_all_valid, dim3(grid), dim3(256), 0, ctx->stream, sp.col, n, (unsigned int*) d_chk);
         if (sp.col.type == LDB_T_UTF8) hipLaunchKernelGGL(k_str_maxlen, dim3(grid), dim3(256), 0, ctx->stream, sp.col, n, d_chk + 1 + s);
      }
   }
   unsigned long long chk[1 + SORT_MAX_SPECS] = {0};
   if (any_chk) {
      LDB_HIP(hipGetLastError());
      LDB_TRY(LDB_READBACK(ctx, chk, d_chk, sizeof(chk)));
      if (chk[0] & 1) LDB_FAIL(LDB_ERR_UNSUPPORTED, "sort: a key contains NULLs (nullable sort keys are not lowered to db.sort_compare)");
   }
   int off = 0;
   for (int s = 0; s < n_specs; s++) {
      DSortSpec& sp = h->specs[s];
      sp.byte_off = off;
      if (sp.col.type == LDB_T_UTF8) {
         const uint64_t m = chk[1 + s];
         if (m > 256) LDB_FAIL(LDB_ERR_UNSUPPORTED, "sort: string key longer than 256 bytes");
         sp.str_pad = (int32_t) m;
         sp.nbytes = sp.str_pad + 4;
      } else if (sp.col.type == LDB_T_DECIMAL128 && sp.col.precision >= 19) {
         sp.nbytes = 16;
      } else {
         sp.nbytes = 8;
      }
      off += sp.nbytes;
   }
   h->words = (off + 7) / 8;
   uint64_t* keys;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &keys, 8 * (size_t) h->words * (size_t) (n ? n : 1)));
   LdbDesc<DSort> d_desc(ctx);
   LDB_TRY(d_desc.upload(h, sizeof(*h)));
   DSort* d = d_desc.p;
   if (n) hipLaunchKernelGGL(k_sort_keys, dim3(grid), dim3(256), 0, ctx->stream, d, keys);
   LDB_HIP(hipGetLastError());
   LDB_TRY(sort_records(ctx, keys, h->words, n, k, perm_out));
   d_desc.release();
   ldb_dev_free(ctx, keys);
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_sort(ldb_ctx* ctx, ldb_rel* in, const ldb_sort_spec* specs, int32_t n_specs, ldb_rel** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "sort: NULL argument");
   LDB_TRY(ldb_rel_force(ctx, in));
   uint32_t* perm;
   LDB_TRY(sort_perm(ctx, in, specs, n_specs, (uint64_t) in->n_rows, &perm));
   return ldb_rel_select(ctx, in, perm, in->n_rows, out);
}
// Heap (reference src/runtime/Heap.cpp:8-72): the k smallest rows under the comparator, sorted.
// Radix select on the leading key word → the few candidate rows → sorted (see sort_records).
extern "C" int32_t ldb_gpu_topk(ldb_ctx* ctx, ldb_rel* in, const ldb_sort_spec* specs, int32_t n_specs, int64_t k, ldb_rel** out) {
   if (!ctx || !in || !out || k < 0) LDB_FAIL(LDB_ERR_INVALID, "topk: bad argument");
   LDB_TRY(ldb_rel_force(ctx, in));
   uint32_t* perm;
   LDB_TRY(sort_perm(ctx, in, specs, n_specs, (uint64_t) k, &perm));
   return ldb_rel_select(ctx, in, perm, std::min<int64_t>(k, in->n_rows), out);
}

// ---------------------------------------------------------------- hash-radix partition (multi-GPU shuffle)
__global__ void k_part_ids(const DKeys* __restrict__ d, uint64_t n, uint32_t nparts, uint64_t* __restrict__ ids) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      ids[i] = (d_hash_keys(*d, i) >> 16) % nparts;
}
__global__ void k_part_hist(const uint64_t* __restrict__ ids, uint64_t n, unsigned long long* __restrict__ hist) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) atomicAdd(&hist[ids[i]], 1ull);
}

extern "C" int32_t ldb_gpu_partition(ldb_ctx* ctx, ldb_rel* in, const ldb_colref* keys, int32_t n_keys, int32_t nparts, const ldb_colref* cols, int32_t n_cols,
                                     ldb_table** out, int64_t* counts) {
   if (!ctx || !in || !out || !counts || nparts < 1 || nparts > 256) LDB_FAIL(LDB_ERR_INVALID, "partition: bad argument (1..256 partitions)");
   LDB_TRY(ldb_rel_force(ctx, in));
   const uint64_t n = (uint64_t) in->n_rows;
   DKeys hk;
   LDB_TRY(ldb_make_dkeys(in, keys, n_keys, &hk));
   LdbDesc<DKeys> dk_desc(ctx);
   LDB_TRY(dk_desc.upload(&hk, sizeof(hk)));
   DKeys* dk = dk_desc.p;
   uint64_t* ids;
   uint32_t* perm;
   unsigned long long* hist;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &ids, 8 * (size_t) (n ? n : 1)));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &perm, 4 * (size_t) (n ? n : 1)));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &hist, 8 * 256));
   LDB_HIP(hipMemsetAsync(hist, 0, 8 * 256, ctx->stream));
   const int grid = ldb_grid_for(ctx, (int64_t) n, 256, 8);
   if (n) {
      hipLaunchKernelGGL(k_part_ids, dim3(grid), dim3(256), 0, ctx->stream, dk, n, (uint32_t) nparts, ids);
      hipLaunchKernelGGL(k_part_hist, dim3(grid), dim3(256), 0, ctx->stream, ids, n, hist);
      hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, ctx->stream, perm, n);
   }
   LDB_HIP(hipGetLastError());
   // stable counting sort on the partition id (<= 8 bits → two 4-bit passes)
   LDB_TRY(radix_sort_perm(ctx, ids, 1, &perm, n, 0, 0, 0, nparts > 16 ? 8 : 4, nullptr));
   std::vector<unsigned long long> hh(256);
   LDB_TRY(LDB_READBACK(ctx, hh.data(), hist, 8 * 256));
   for (int p = 0; p < nparts; p++) counts[p] = (int64_t) hh[(size_t) p];
   dk_desc.release();
   ldb_dev_free(ctx, ids);
   ldb_dev_free(ctx, hist);
   ldb_rel* permuted;
   LDB_TRY(ldb_rel_select(ctx, in, perm, (int64_t) n, &permuted));
   int32_t s = ldb_gpu_materialize(ctx, permuted, cols, n_cols, out);
   ldb_gpu_rel_release(ctx, permuted);
   return s;
}
