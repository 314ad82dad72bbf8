// ldb_wc.hip — write-combining radix partition of 32-bit keys (+ an optional 32-bit payload).
//
// North star: "radix-partitioned hash-join build/probe … LDS-staged".  A scatter with one LDS cursor per partition writes 4
// bytes at a time to P open streams per workgroup; with P in the thousands (what it takes to cut a 150 MB table into
// LDS-sized pieces) the L2s hold far more partially written lines than they have room for and evict them half-filled —
// round 3 measured 18.7 ms for 600 M keys into 2 300 partitions against 3.3 ms into 16.  Here a workgroup first SORTS a tile
// of 4 096 items by partition in LDS (counting sort: LDS histogram → scan → placement) and then writes the tile out in
// partition order: the items of one partition leave as one contiguous run — full cache lines when the fan-out of a pass is
// ≈ 64 (64-item runs on average) — and a large partition count is reached in TWO passes (√P each): pass 2 partitions every
// pass-1 partition by the low digit, in place of the one-pass scatter's P streams.
//   pass = k_wc_hist (items per (source partition, digit, chunk)) → exclusive scan → k_wc_scatter
// Bytes per item and pass: 4 read (histogram) + 4 | 8 read + 4 | 8 written (keys | keys + payload).
// The partition number of a key is ((key - bias) >> shift), 0 for keys outside [bias, bias + range] — the slot / word
// position of a direct-addressed join table (ldb_join.hip) or a direct group-by slot (ldb_gbhost.hip).
// The reference has no counterpart: its chained hash table takes the cache misses (LazyJoinHashtable.cpp:12-34) and its
// pre-aggregation merges per partition on the CPU (PreAggregationHashtable.cpp:76-158).
#include "ldb_internal.h"
#include <algorithm>

#define WC_BLOCK 256
#define WC_ITEMS 16
#define WC_TILE (WC_BLOCK * WC_ITEMS)
#define WC_MAX_DIGIT 256

struct DWc {
   uint64_t n;
   uint32_t bias, range, shift; // partition q = ((key - bias) > range ? 0 : (key - bias) >> shift)
   uint32_t lo_bits; // q = hi digit << lo_bits | lo digit
   uint32_t level; // 1: partition [0, n) by the hi digit; 2: partition every hi-digit range by the lo digit
   uint32_t n_src, digits, chunks; // sources (1 | number of hi digits), digits of this level, chunks per source
   uint32_t rep_bits; // LDS counters are replicated 2^rep_bits times (lane % R picks the replica): with <= 64 digits a wave's 64
                      // lanes would otherwise queue on a handful of addresses (the histogram ran at 2.1 TB/s instead of 4)
};
__device__ __forceinline__ uint32_t d_wc_digit(const DWc& d, uint32_t key) {
   const uint32_t r = key - d.bias;
   const uint32_t q = r > d.range ? 0u : r >> d.shift;
   return d.level == 1 ? q >> d.lo_bits : q & ((1u << d.lo_bits) - 1u);
}
// the rows a workgroup owns: chunk `c` of source `s` (level 1: the whole input; level 2: the s-th range of pass 1,
// whose bounds are the pass-1 offsets of its first chunk and of the next digit's first chunk)
__device__ __forceinline__ void d_wc_range(const DWc& d, const uint32_t* __restrict__ src_offs, uint32_t src_chunks, uint32_t s, uint32_t c, uint64_t* b, uint64_t* e) {
   uint64_t sb = 0, se = d.n;
   if (d.level == 2) {
      sb = src_offs[(uint64_t) s * src_chunks];
      se = s + 1 < d.n_src ? (uint64_t) src_offs[(uint64_t) (s + 1) * src_chunks] : d.n;
   }
   const uint64_t len = se - sb, per = (len + d.chunks - 1) / d.chunks;
   *b = sb + (uint64_t) c * per < se ? sb + (uint64_t) c * per : se;
   *e = *b + per < se ? *b + per : se;
}
// hist[((s * digits + p) * chunks) + c] = items of chunk (s, c) whose digit is p
__global__ __launch_bounds__(WC_BLOCK) void k_wc_hist(DWc d, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ src_offs, uint32_t src_chunks, uint32_t* __restrict__ hist) {
   __shared__ uint32_t h[WC_MAX_DIGIT * 8];
   const uint32_t s = blockIdx.x / d.chunks, c = blockIdx.x % d.chunks;
   const uint32_t R = 1u << d.rep_bits, rep = threadIdx.x & (R - 1);
   for (uint32_t k = threadIdx.x; k < d.digits * R; k += WC_BLOCK) h[k] = 0;
   __syncthreads();
   uint64_t b, e;
   d_wc_range(d, src_offs, src_chunks, s, c, &b, &e);
   // four keys per thread and step: the loads of a step are independent
   uint64_t i = b + threadIdx.x;
   for (; i + 3 * WC_BLOCK < e; i += 4 * WC_BLOCK) {
      const uint32_t k0 = keys[i], k1 = keys[i + WC_BLOCK], k2 = keys[i + 2 * WC_BLOCK], k3 = keys[i + 3 * WC_BLOCK];
      atomicAdd(&h[(d_wc_digit(d, k0) << d.rep_bits) | rep], 1u);
      atomicAdd(&h[(d_wc_digit(d, k1) << d.rep_bits) | rep], 1u);
      atomicAdd(&h[(d_wc_digit(d, k2) << d.rep_bits) | rep], 1u);
      atomicAdd(&h[(d_wc_digit(d, k3) << d.rep_bits) | rep], 1u);
   }
   for (; i < e; i += WC_BLOCK) atomicAdd(&h[(d_wc_digit(d, keys[i]) << d.rep_bits) | rep], 1u);
   __syncthreads();
   if (threadIdx.x < d.digits) {
      uint32_t sum = 0;
      for (uint32_t r = 0; r < R; r++) sum += h[(threadIdx.x << d.rep_bits) | r];
      hist[((uint64_t) s * d.digits + threadIdx.x) * d.chunks + c] = sum;
   }
}
// tile-sort in LDS, then write every partition's items of the tile as one run
__global__ __launch_bounds__(WC_BLOCK) void k_wc_scatter(DWc d, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pay, int identity_payload, const uint32_t* __restrict__ src_offs,
                                                         uint32_t src_chunks, const uint32_t* __restrict__ offs, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ pay_out) {
   __shared__ uint32_t s_key[WC_TILE];
   __shared__ uint32_t s_pay[WC_TILE];
   __shared__ uint32_t s_cnt[WC_MAX_DIGIT], s_base[WC_MAX_DIGIT], s_cur[WC_MAX_DIGIT], s_wave[4];
   const uint32_t s = blockIdx.x / d.chunks, c = blockIdx.x % d.chunks;
   const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
   // rank counters: (digit, replica) pairs, digit-major — the exclusive scan over them gives every pair its own contiguous
   // piece of the digit's run, so lanes that share a digit mostly bump different counters (digits x replicas <= 256)
   const uint32_t R = 1u << d.rep_bits, rep = t & (R - 1);
   s_cur[t] = t < d.digits ? offs[((uint64_t) s * d.digits + t) * d.chunks + c] : 0u;
   uint64_t b, e;
   d_wc_range(d, src_offs, src_chunks, s, c, &b, &e);
   const bool with_pay = pay_out != nullptr;
   for (uint64_t t0 = b; t0 < e; t0 += WC_TILE) {
      const uint32_t tile_n = (uint32_t) (e - t0 < WC_TILE ? e - t0 : WC_TILE);
      s_cnt[t] = 0;
      __syncthreads();
      uint32_t key[WC_ITEMS], pv[WC_ITEMS], dr[WC_ITEMS]; // dr = digit << 16 | rank inside the tile's partition
#pragma unroll
      for (int k = 0; k < WC_ITEMS; k++) {
         const uint32_t j = (uint32_t) k * WC_BLOCK + t;
         if (j < tile_n) {
            key[k] = keys[t0 + j];
            pv[k] = with_pay ? (identity_payload ? (uint32_t) (t0 + j) : pay[t0 + j]) : 0u;
         }
      }
#pragma unroll
      for (int k = 0; k < WC_ITEMS; k++) {
         const uint32_t j = (uint32_t) k * WC_BLOCK + t;
         if (j < tile_n) {
            const uint32_t cell = (d_wc_digit(d, key[k]) << d.rep_bits) | rep;
            dr[k] = (cell << 16) | atomicAdd(&s_cnt[cell], 1u); // (a tile holds 4 096 items: the rank fits 16 bits)
         }
      }
      __syncthreads();
      { // exclusive scan of the (<= 256) digit counts: one per thread
         const uint32_t own = s_cnt[t];
         uint32_t incl = own;
#pragma unroll
         for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if ((int) lane >= off) incl += up;
         }
         if (lane == 63) s_wave[wave] = incl;
         __syncthreads();
         uint32_t wave_off = 0;
         for (uint32_t w = 0; w < wave; w++) wave_off += s_wave[w];
         s_base[t] = wave_off + incl - own;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < WC_ITEMS; k++) {
         const uint32_t j = (uint32_t) k * WC_BLOCK + t;
         if (j < tile_n) {
            const uint32_t at = s_base[dr[k] >> 16] + (dr[k] & 0xFFFFu);
            s_key[at] = key[k];
            if (with_pay) s_pay[at] = pv[k];
         }
      }
      __syncthreads();
      for (uint32_t j = t; j < tile_n; j += WC_BLOCK) {
         const uint32_t kk = s_key[j];
         const uint32_t dg = d_wc_digit(d, kk);
         const uint64_t dest = (uint64_t) s_cur[dg] + (j - s_base[dg << d.rep_bits]); // s_base of the digit's first replica = start of its run in the tile
         keys_out[dest] = kk;
         if (with_pay) pay_out[dest] = s_pay[j];
      }
      __syncthreads();
      if (t < d.digits) {
         uint32_t sum = 0;
         for (uint32_t r = 0; r < R; r++) sum += s_cnt[(t << d.rep_bits) | r];
         s_cur[t] += sum;
      }
      __syncthreads();
   }
}

// Partitions n keys (and, if pay_out != NULL, a payload: pay_in, or the item number when pay_in == NULL) into `nparts`
// partitions q(key) = (key - bias) >> shift (0 outside [bias, bias + range]); keys_out / pay_out hold the partitions back to
// back in q order.  part_offs_out (device, may be NULL) receives an offsets table usable as offs[q * *chunks_out]: the
// start of partition q (the table has one entry per (partition, chunk); callers read entry 0 of each partition).
// Two passes when nparts > 64.  tmp buffers are the caller's (n items each; tmp_pay only with a payload and two passes).
int32_t ldb_wc_partition(ldb_ctx* ctx, const uint32_t* keys_in, const uint32_t* pay_in, uint64_t n, uint32_t bias, uint32_t range, uint32_t shift, uint32_t nparts, uint32_t* keys_out,
                         uint32_t* pay_out, uint32_t** part_offs_out, uint32_t* chunks_out, const char* prof_hist, const char* prof_scatter) {
   if (nparts < 1 || nparts > WC_MAX_DIGIT * WC_MAX_DIGIT) LDB_FAIL(LDB_ERR_INVALID, "wc_partition: %u partitions", nparts);
   if (n >= (uint64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "wc_partition: %llu items exceed 32-bit positions", (unsigned long long) n);
   uint32_t lo_bits = 0;
   if (nparts > 64) { // two passes of about sqrt(nparts) each
      uint32_t lg = 0;
      while ((1u << lg) < nparts) lg++;
      lo_bits = lg / 2;
   }
   const uint32_t lo_digits = 1u << lo_bits;
   const uint32_t hi_digits = (nparts + lo_digits - 1) / lo_digits;
   if (hi_digits > WC_MAX_DIGIT) LDB_FAIL(LDB_ERR_INVALID, "wc_partition: %u partitions need more than two passes", nparts);
   const bool two = lo_bits > 0;
   const char *ph = prof_hist, *ps = prof_scatter; // (string literals of the caller: LdbProf keeps the pointers)
   LdbBufs tmp(ctx);
   DWc d;
   memset(&d, 0, sizeof(d));
   d.n = n;
   d.bias = bias;
   d.range = range;
   d.shift = shift;
   d.lo_bits = lo_bits;
   // ---- pass 1: the whole input by the hi digit (the only pass when nparts <= 64)
   const uint32_t g1 = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) ctx->cus * 4, (n + 4 * WC_TILE - 1) / (4 * WC_TILE)));
   auto rep_bits_for = [](uint32_t digits) {
      uint32_t rb = 0;
      while (rb < 3 && (digits << (rb + 1)) <= WC_MAX_DIGIT) rb++;
      return rb;
   };
   d.level = 1;
   d.n_src = 1;
   d.digits = hi_digits;
   d.chunks = g1;
   d.rep_bits = rep_bits_for(hi_digits);
   uint32_t *hist1, *offs1;
   const size_t h1n = (size_t) hi_digits * g1;
   LDB_TRY(tmp.alloc(&hist1, 4 * h1n));
   LDB_TRY(tmp.alloc(&offs1, 4 * (h1n + 1)));
   uint32_t *k1 = keys_out, *p1 = pay_out;
   if (two) {
      LDB_TRY(tmp.alloc(&k1, 4 * (size_t) (n ? n : 1)));
      if (pay_out) LDB_TRY(tmp.alloc(&p1, 4 * (size_t) (n ? n : 1)));
   }
   {
      LdbProf prof_(ctx, ph);
      hipLaunchKernelGGL(k_wc_hist, dim3(g1), dim3(WC_BLOCK), 0, ctx->stream, d, keys_in, (const uint32_t*) nullptr, 0u, hist1);
   }
   LDB_TRY(ldb_exclusive_scan_u32(ctx, hist1, offs1, (int64_t) h1n, nullptr));
   {
      LdbProf prof_(ctx, ps);
      hipLaunchKernelGGL(k_wc_scatter, dim3(g1), dim3(WC_BLOCK), 0, ctx->stream, d, keys_in, pay_in, pay_in ? 0 : 1, (const uint32_t*) nullptr, 0u, (const uint32_t*) offs1, k1, p1);
   }
   LDB_HIP(hipGetLastError());
   if (!two) {
      if (part_offs_out) {
         tmp.keep(offs1);
         *part_offs_out = offs1;
      }
      if (chunks_out) *chunks_out = g1;
      return LDB_OK;
   }
   // ---- pass 2: every hi-digit range by the lo digit
   const uint32_t c2 = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(64, (n / hi_digits + 2 * WC_TILE - 1) / (2 * WC_TILE)));
   d.level = 2;
   d.n_src = hi_digits;
   d.digits = lo_digits;
   d.chunks = c2;
   d.rep_bits = rep_bits_for(lo_digits);
   uint32_t *hist2, *offs2;
   const size_t h2n = (size_t) hi_digits * lo_digits * c2;
   LDB_TRY(tmp.alloc(&hist2, 4 * h2n));
   LDB_TRY(tmp.alloc(&offs2, 4 * (h2n + 1)));
   {
      LdbProf prof_(ctx, ph);
      hipLaunchKernelGGL(k_wc_hist, dim3(hi_digits * c2), dim3(WC_BLOCK), 0, ctx->stream, d, (const uint32_t*) k1, (const uint32_t*) offs1, g1, hist2);
   }
   LDB_TRY(ldb_exclusive_scan_u32(ctx, hist2, offs2, (int64_t) h2n, nullptr));
   {
      LdbProf prof_(ctx, ps);
      hipLaunchKernelGGL(k_wc_scatter, dim3(hi_digits * c2), dim3(WC_BLOCK), 0, ctx->stream, d, (const uint32_t*) k1, (const uint32_t*) p1, 0, (const uint32_t*) offs1, g1, (const uint32_t*) offs2, keys_out,
                         pay_out);
   }
   LDB_HIP(hipGetLastError());
   if (part_offs_out) {
      tmp.keep(offs2);
      *part_offs_out = offs2;
   }
   if (chunks_out) *chunks_out = c2;
   return LDB_OK;
}
