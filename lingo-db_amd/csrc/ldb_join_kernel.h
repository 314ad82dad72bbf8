// ldb_join_kernel.h — device code of the hash-join build / probe kernels.
// Compiled ahead of time (generic) and, for large inputs, at run time with the key metadata and
// join kind as compile-time constants (ldb_jit.hip) — same source (see ldb_gb_kernel.h).
// Replaces (reference): HashIndexedView::build (src/runtime/LazyJoinHashtable.cpp:12-34) and the
// generated probe (LookupHashIndexedViewLowering / ScanListLowering,
// src/compiler/Conversion/SubOpToControlFlow/SubOpToControlFlow.cpp:2558-2586, 2254-2313).
#pragma once
#include "ldb_keys.h"

struct DJoin {
   // ---- run-time part
   uint64_t n_rows; // rows of the relation the kernel iterates (build or probe)
   uint64_t cap; // table capacity (pow2)
   uint64_t slots; // uint64_t*
   uint64_t out_probe; // uint32_t*
   uint64_t out_build; // uint32_t*
   uint64_t out_cap;
   uint64_t counter; // unsigned long long*: [0] = rows produced (may exceed out_cap), [1] = matches
   uint64_t bitmap; // uint64_t*: SEMI / ANTI / unique-build INNER
   uint64_t mark; // uint8_t*: MARK
   uint64_t match; // uint32_t*: unique-build path: build row (or LDB_NULL_ROW) per probe row; pairs path: per-chunk counts / offsets
   uint64_t flags; // uint32_t*: build: [0] |= 1 when two build rows carry the same key (or tag), |= 2 on a long probe run
   int64_t kmin, kmax; // KEY32 + ordered slots: range of the build keys
   uint64_t kmult; // slot = ((key - kmin) * kmult) >> 32
   uint64_t key_bits; // uint32_t*: has_key_bits: bit (key - kmin) set ⇔ key is in the table
   uint64_t next; // uint32_t*: chained: next[row] = following row of the same key + 1 (0 = end)
   // ---- metadata
   int32_t key32;
   int32_t kind;
   int32_t has_bitmap, has_mark, has_flags;
   // KEY32 tables place a key at a slot proportional to its position in [kmin, kmax] instead of at
   // its hash: fact tables are clustered on their foreign keys, so consecutive probe rows then
   // touch neighbouring slots (cache lines are reused, the table is walked almost sequentially)
   // instead of one random line each.  The build measures its probe runs and falls back to hashed
   // slots (ordered_slots = 0) when the key distribution makes them long.
   int32_t ordered_slots;
   DKeys bkeys;
   DKeys pkeys;
   // conjuncts of a lazy (not materialised) probe relation, evaluated per probe row before the
   // lookup: scan → filter → probe in one kernel, like the reference's fused pipelines
   int32_t n_ppreds;
   // ordered KEY32 tables over a small key range also keep one BIT per key value (key_bits): the
   // probe tests it first.  The bit array is 64x smaller than the slot array (Q9's 1.08 M green
   // part keys over a 20 M range: 2.5 MB, L2-resident, against a 32 MB slot array), so a selective
   // probe of unclustered keys (600 M random l_partkey values, 5.4 % hits) mostly never leaves L2.
   int32_t has_key_bits;
   // Duplicate-heavy build keys: the default layout spends one slot per build ROW, so a key that
   // repeats d times costs O(d^2) slot visits to insert and O(d) extra per probe miss in its run.
   // When a build pass reports such runs the table is rebuilt CHAINED: one slot per distinct key,
   // the key's rows linked through next[] — the reference's layout (chained HashIndexedView).
   int32_t chained;
   int32_t pad3;
   DPred ppreds[LDB_MAX_PREDS];
};

__device__ __forceinline__ bool d_probe_pass(const DJoin& m, const DJoin* __restrict__ d, uint64_t i) {
   bool pass = true;
   const int np = m.n_ppreds;
   LDB_UNROLL
   for (int p = 0; p < np; p++)
      if (pass) pass = d_eval_pred(PV(m.ppreds[p], d->ppreds[p]), i);
   return pass;
}

// first slot of a key: hashed, or (KEY32 tables with ordered slots) proportional to the key's
// position in the build key range
#define JOIN_LONG_RUN 512
__device__ __forceinline__ uint64_t d_join_slot(const DJoin& m, const DJoin* __restrict__ d, uint64_t h, int64_t key, uint64_t mask) {
   if (m.key32 && m.ordered_slots) return (((uint64_t) (key - d->kmin) * d->kmult) >> 32) & mask;
   // The reference indexes its chained table with `hash & mask`; under OPEN ADDRESSING the low bits
   // of db.hash are not good enough — structured keys such as (ps_partkey, ps_suppkey) produced
   // probe runs of 500+ slots in a table filled to 26 % (Q9's two-column join: 17 ms → 1 ms).  A
   // finalising mix (positions are internal, results do not depend on them) spreads them.
   uint64_t x = h;
   x ^= x >> 33;
   x *= 0xFF51AFD7ED558CCDull;
   x ^= x >> 33;
   return x & mask;
}

__device__ __forceinline__ void join_build_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows, mask = d->cap - 1;
   unsigned long long* slots = gptr_mut<unsigned long long>(d->slots);
   const KV bkeys(m.bkeys, d->bkeys);
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      bool nul;
      uint64_t h = d_hash_keys(bkeys, i, &nul);
      if (nul) continue; // a NULL key can never be matched (eq on NULL is false)
      uint64_t word;
      int64_t key = 0;
      if (m.key32) {
         const CV c = bkeys.col(0);
         key = d_load_i64(c, d_phys_row(c, i));
         word = ((uint64_t) (uint32_t) key << 32) | (uint64_t) ((uint32_t) i + 1u);
      } else {
         word = (h & 0xFFFFFFFF00000000ull) | (uint64_t) ((uint32_t) i + 1u);
      }
      uint64_t pos = d_join_slot(m, d, h, key, mask); // (the key bits are set by join_key_bits_body, a pass of its own)
      uint32_t steps = 0;
      if (m.chained) {
         // one slot per DISTINCT key; the rows of a key hang off it through next[] (push-front with a
         // CAS on the slot word, like the reference's HashIndexedView::build, LazyJoinHashtable.cpp:12-34).
         // Insertion cost no longer depends on how often a key repeats.
         uint32_t* next = gptr_mut<uint32_t>(d->next);
         for (;;) {
            unsigned long long w = __hip_atomic_load(&slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == 0) {
               next[i] = 0;
               w = atomicCAS(&slots[pos], 0ull, (unsigned long long) word);
               if (w == 0) break;
            }
            if ((w >> 32) == (word >> 32) && (m.key32 || d_keys_equal(bkeys, (uint64_t) ((uint32_t) w - 1u), bkeys, i, false))) {
               for (;;) { // same key: become the new head
                  next[i] = (uint32_t) w;
                  const unsigned long long old = atomicCAS(&slots[pos], w, (unsigned long long) word);
                  if (old == w) break;
                  w = old; // another row of this key got in first (the key part of the word is unchanged)
               }
               break;
            }
            pos = (pos + 1) & mask;
         }
         continue;
      }
      for (;;) {
         unsigned long long old = atomicCAS(&slots[pos], 0ull, (unsigned long long) word);
         if (old == 0) break;
         // same key (KEY32) / same hash tag AND equal key columns (TAG mode: a 32-bit tag collision of
         // different keys is not a duplicate): the build side is not unique
         if (m.has_flags && (old >> 32) == (word >> 32) && (m.key32 || d_keys_equal(bkeys, (uint64_t) ((uint32_t) old - 1u), bkeys, i, false))) {
            uint32_t* f = gptr_mut<uint32_t>(d->flags);
            // test before the atomic: a build with many duplicate keys would otherwise queue millions
            // of atomics on this one address (~10 ns each: 15 ms on Q9's 32 M-row build)
            if ((__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) == 0) atomicOr(f, 1u);
         }
         pos = (pos + 1) & mask;
         // a long run: skewed keys under ordered slots, or a key that repeats very often (one slot
         // per ROW makes runs cost O(length^2)).  Give up at once; the host rebuilds — hashed first,
         // then chained.
         if (++steps == JOIN_LONG_RUN && m.has_flags) {
            atomicOr(gptr_mut<uint32_t>(d->flags), 2u);
            break;
         }
      }
   }
}

// min / max of the (single, integer) build key — the range the ordered slots are spread over
__device__ __forceinline__ void join_key_range_body(const DJoin& m, const DJoin* __restrict__ d, long long* __restrict__ out) {
   const uint64_t n = d->n_rows;
   const KV bkeys(m.bkeys, d->bkeys);
   const CV c = bkeys.col(0);
   long long lo = 0x7FFFFFFFFFFFFFFFll, hi = -0x7FFFFFFFFFFFFFFFll - 1;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      uint32_t row = d_phys_row(c, i);
      if (!d_valid(c, row)) continue;
      long long k = d_load_i64(c, row);
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

// has_key_bits: set bit (key - kmin) for every build key.  Build sides are usually key-ordered
// (a primary key column, or a filtered subset of one), so the lanes of a wave hit the same 32-bit
// word: adjacent lanes with the same word OR their bits together first and only the first lane of
// each run issues the atomic (32x fewer atomics on one address for a dense key column; any lane
// order stays correct — a run split in two just issues two atomics).
__device__ __forceinline__ void join_key_bits_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   const KV bkeys(m.bkeys, d->bkeys);
   const CV c = bkeys.col(0);
   uint32_t* bits = gptr_mut<uint32_t>(d->key_bits);
   const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
   for (uint64_t base = blockIdx.x * (uint64_t) blockDim.x; base < n; base += stride) { // uniform trip count per block
      const uint64_t i = base + threadIdx.x;
      unsigned long long w = ~0ull;
      uint32_t v = 0;
      if (i < n) {
         const uint32_t row = d_phys_row(c, i);
         if (d_valid(c, row)) {
            const uint64_t r = (uint64_t) (d_load_i64(c, row) - d->kmin);
            w = r >> 5;
            v = 1u << (r & 31);
         }
      }
      for (int off = 1; off < 64; off <<= 1) { // a run may span the whole wave (duplicate keys)
         const unsigned long long w2 = __shfl_down(w, off);
         const uint32_t v2 = __shfl_down(v, off);
         if (w2 == w) v |= v2; // (past the end of the wave __shfl_down returns the lane's own value)
      }
      const unsigned long long wprev = __shfl_up(w, 1);
      const bool leader = (threadIdx.x & 63) == 0 || wprev != w;
      if (leader && w != ~0ull) atomicOr(bits + w, v);
   }
}

// chained tables: the rows of one key, head first (next[row] = following row + 1, 0 = end)
template <typename EMIT>
__device__ __forceinline__ uint32_t d_probe_chain(const DJoin* __restrict__ d, uint32_t head_plus1, EMIT emit) {
   const uint32_t* next = gptr<uint32_t>(d->next);
   uint32_t matches = 0;
   for (uint32_t r = head_plus1; r != 0; r = next[r - 1u]) {
      matches++;
      if (!emit(r - 1u)) break;
   }
   return matches;
}

// One probe row → visits its slot run.  EMIT is called for every match with the build row and
// returns whether to keep scanning.
template <typename EMIT>
__device__ __forceinline__ uint32_t d_probe_row(const DJoin& m, const DJoin* __restrict__ d, uint64_t i, EMIT emit) {
   const KV pkeys(m.pkeys, d->pkeys);
   bool nul;
   uint64_t h = d_hash_keys(pkeys, i, &nul);
   if (nul) return 0;
   const uint64_t mask = d->cap - 1;
   const uint64_t* slots = gptr<uint64_t>(d->slots);
   uint64_t pos = d_join_slot(m, d, h, 0, mask); // hashed start slot (ordered KEY32 tables: recomputed from the key below)
   uint32_t matches = 0;
   if (m.key32) {
      const CV c = pkeys.col(0);
      int64_t kv = d_load_i64(c, d_phys_row(c, i));
      if (kv != (int64_t) (int32_t) kv) return 0; // wider probe value can equal no 32-bit build key
      if (m.ordered_slots) {
         if (kv < d->kmin || kv > d->kmax) return 0; // outside the build key range
         if (m.has_key_bits) {
            const uint64_t r = (uint64_t) (kv - d->kmin);
            if (((gptr<uint32_t>(d->key_bits)[r >> 5] >> (r & 31)) & 1u) == 0) return 0; // not a build key
         }
         pos = d_join_slot(m, d, h, kv, mask);
      }
      const uint32_t key = (uint32_t) kv;
      for (;;) {
         uint64_t w = slots[pos];
         if (w == 0) break;
         if ((uint32_t) (w >> 32) == key) {
            if (m.chained) return d_probe_chain(d, (uint32_t) w, emit); // the key's only slot: its rows are the chain
            matches++;
            if (!emit((uint32_t) w - 1u)) break;
         }
         pos = (pos + 1) & mask;
      }
   } else {
      const KV bkeys(m.bkeys, d->bkeys);
      for (;;) {
         uint64_t w = slots[pos];
         if (w == 0) break;
         if ((w >> 32) == (h >> 32) && d_keys_equal(bkeys, (uint64_t) ((uint32_t) w - 1u), pkeys, i, false)) {
            if (m.chained) return d_probe_chain(d, (uint32_t) w, emit);
            matches++;
            if (!emit((uint32_t) w - 1u)) break;
         }
         pos = (pos + 1) & mask;
      }
   }
   return matches;
}

// INNER / LEFT_OUTER / SINGLE with possibly duplicated build keys, two passes and no atomics on
// an output cursor (one contended cursor address serialises at ~10 ns per wave-append in the
// L2: Q12's 150 M-row probe spent 22 ms there).  Pass 1 counts the rows every 64-row chunk
// produces, a device scan turns the counts into chunk offsets, pass 2 re-walks (the slot runs are
// now cache-resident) and every lane writes its pairs at chunk offset + in-wave prefix: output in
// probe order, sized exactly.
// rows one probe row contributes: its matches, or one NULL-padded row when an outer join finds none
__device__ __forceinline__ uint32_t d_pairs_of_row(const DJoin& m, const DJoin* __restrict__ d, uint64_t i) {
   if (!d_probe_pass(m, d, i)) return 0; // (the host only fuses filters for INNER here)
   uint32_t c = d_probe_row(m, d, i, [&](uint32_t) { return m.kind != LDB_JOIN_SINGLE; });
   return (m.kind != LDB_JOIN_INNER && c == 0) ? 1u : c;
}
__device__ __forceinline__ void join_probe_pairs_count_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   const uint64_t n_chunks = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   uint32_t* chunk_cnt = gptr_mut<uint32_t>(d->match);
   unsigned long long total = 0; // exact 64-bit row count (the 32-bit offsets cannot detect > 4 G rows)
   for (uint64_t w = wave; w < n_chunks; w += n_waves) {
      const uint64_t i = w * 64 + lane;
      uint32_t c = i < n ? d_pairs_of_row(m, d, i) : 0u;
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
      if (lane == 0) chunk_cnt[w] = c;
      total += c;
   }
   if (lane == 0 && total) atomicAdd(gptr_mut<unsigned long long>(d->counter), total);
}
__device__ __forceinline__ void join_probe_pairs_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   const uint64_t n_chunks = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   const uint32_t* chunk_off = gptr<uint32_t>(d->match);
   uint32_t* out_probe = gptr_mut<uint32_t>(d->out_probe);
   uint32_t* out_build = gptr_mut<uint32_t>(d->out_build);
   for (uint64_t w = wave; w < n_chunks; w += n_waves) {
      const uint64_t i = w * 64 + lane;
      const uint32_t c = i < n ? d_pairs_of_row(m, d, i) : 0u;
      uint32_t incl = c;
      for (int off = 1; off < 64; off <<= 1) {
         uint32_t up = __shfl_up(incl, off);
         if (lane >= (uint32_t) off) incl += up;
      }
      if (c != 0) {
         const uint64_t at = (uint64_t) chunk_off[w] + (incl - c);
         uint32_t emitted = 0;
         d_probe_row(m, d, i, [&](uint32_t brow) {
            out_probe[at + emitted] = (uint32_t) i;
            out_build[at + emitted] = brow;
            emitted++;
            return m.kind != LDB_JOIN_SINGLE;
         });
         if (emitted == 0) { // unmatched (or NULL-key) probe row of an outer join
            out_probe[at] = (uint32_t) i;
            out_build[at] = LDB_NULL_ROW;
         }
      }
   }
}

// count matches only — the probe micro-benchmark kernel (Grows/s)
__device__ __forceinline__ void join_probe_count_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   unsigned long long local = 0;
   const uint64_t tid = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x, nth = (uint64_t) gridDim.x * blockDim.x;
   for (uint64_t i0 = tid; i0 < n; i0 += 4 * nth) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const uint64_t i = i0 + (uint64_t) u * nth;
         if (i < n && d_probe_pass(m, d, i)) local += d_probe_row(m, d, i, [](uint32_t) { return true; });
      }
   }
   for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
   if ((threadIdx.x & 63) == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter) + 1, local);
}

// ---------------------------------------------------------------- fused filter: LDS-staged compaction
// With a fused filter a lane whose row fails would idle while its neighbours walk the table, and a
// probe is latency-bound: throughput is proportional to the lanes actually probing (54 % for
// Q3's lineitem side).  So a workgroup first filters a TILE of JT_ROWS consecutive rows
// (predicate-major, JT_U rows per thread, coalesced), compacts the survivors' tile-relative row
// numbers into an LDS queue (ballot + popcount rank, one LDS atomic per wave and batch row), and
// then probes from the queue with every lane busy.  Results that must stay in row order go through
// a per-tile bitmap in LDS (JT_ROWS / 64 words) that is written out once per tile.
#define JT_BLOCK 256
#define JT_U 8
#define JT_ROWS (JT_BLOCK * JT_U)
struct JoinTile {
   unsigned short q[JT_ROWS];
   unsigned long long bm[JT_ROWS / 64];
   uint32_t qn;
};
// filter rows [base, base + JT_ROWS) → st.q[0 .. st.qn); ends with a barrier
__device__ __forceinline__ void d_tile_filter(const DJoin& m, const DJoin* __restrict__ d, uint64_t base, uint64_t n, JoinTile& st) {
   const uint32_t t = threadIdx.x, lane = t & 63;
   if (t < JT_ROWS / 64) st.bm[t] = 0;
   if (t == 0) st.qn = 0;
   __syncthreads();
   uint64_t rows[JT_U];
   bool pass[JT_U];
#pragma unroll
   for (int u = 0; u < JT_U; u++) {
      rows[u] = base + (uint64_t) u * JT_BLOCK + t;
      pass[u] = rows[u] < n;
   }
   d_eval_conj_batch<JT_U>(m.ppreds, d->ppreds, m.n_ppreds, rows, pass);
#pragma unroll
   for (int u = 0; u < JT_U; u++) {
      const uint64_t mask = __ballot(pass[u]);
      uint32_t b = 0;
      if (mask != 0 && lane == (uint32_t) __builtin_ctzll(mask)) b = atomicAdd(&st.qn, (uint32_t) __popcll(mask));
      b = __shfl(b, mask ? __builtin_ctzll(mask) : 0);
      if (pass[u]) st.q[b + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short) (u * JT_BLOCK + t);
   }
   __syncthreads();
}

__device__ __forceinline__ void join_probe_unique_filtered_body(const DJoin& m, const DJoin* __restrict__ d) {
   __shared__ JoinTile st;
   const uint64_t n = d->n_rows;
   const uint64_t n_words = (n + 63) / 64, n_tiles = (n + JT_ROWS - 1) / JT_ROWS;
   uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
   uint32_t* match = gptr_mut<uint32_t>(d->match);
   unsigned long long local = 0;
   for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      const uint64_t base = tile * JT_ROWS;
      d_tile_filter(m, d, base, n, st);
      const uint32_t qn = st.qn;
      for (uint32_t j = threadIdx.x; j < qn; j += JT_BLOCK) {
         const uint32_t r = st.q[j];
         uint32_t b = LDB_NULL_ROW;
         d_probe_row(m, d, base + r, [&](uint32_t x) {
            b = x;
            return false;
         });
         if (b != LDB_NULL_ROW) {
            match[base + r] = b;
            atomicOr(&st.bm[r >> 6], 1ull << (r & 63));
         }
      }
      __syncthreads();
      if (threadIdx.x < JT_ROWS / 64) {
         const uint64_t w = tile * (JT_ROWS / 64) + threadIdx.x;
         if (w < n_words) {
            bitmap[w] = st.bm[threadIdx.x];
            local += (unsigned long long) __popcll(st.bm[threadIdx.x]);
         }
      }
      __syncthreads();
   }
   if (threadIdx.x < 64) {
      for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
      if (threadIdx.x == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
   }
}

// SEMI / ANTI / MARK: existence per probe row → bitmap word per wave (rows in ascending order)
__device__ __forceinline__ void join_probe_exists_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   if (m.n_ppreds > 0) { // fused filter: tile compaction (never MARK: the host forces those)
      __shared__ JoinTile st;
      const uint64_t n_words = (n + 63) / 64, n_tiles = (n + JT_ROWS - 1) / JT_ROWS;
      uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
      unsigned long long local = 0;
      for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
         const uint64_t base = tile * JT_ROWS;
         d_tile_filter(m, d, base, n, st);
         const uint32_t qn = st.qn;
         for (uint32_t j = threadIdx.x; j < qn; j += JT_BLOCK) {
            const uint32_t r = st.q[j];
            const bool hit = d_probe_row(m, d, base + r, [](uint32_t) { return false; }) != 0;
            if (m.kind == LDB_JOIN_ANTI ? !hit : hit) atomicOr(&st.bm[r >> 6], 1ull << (r & 63));
         }
         __syncthreads();
         if (threadIdx.x < JT_ROWS / 64) {
            const uint64_t w = tile * (JT_ROWS / 64) + threadIdx.x;
            if (w < n_words) {
               bitmap[w] = st.bm[threadIdx.x];
               local += (unsigned long long) __popcll(st.bm[threadIdx.x]);
            }
         }
         __syncthreads();
      }
      if (threadIdx.x < 64) {
         for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
         if (threadIdx.x == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
      }
      return;
   }
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
   uint8_t* mark = gptr_mut<uint8_t>(d->mark);
   unsigned long long local = 0;
   for (uint64_t w = wave; w < n_words; w += n_waves) {
      uint64_t i = w * 64 + lane;
      bool hit = false;
      const bool pass = i < n && d_probe_pass(m, d, i); // rows a fused filter rejects do not exist
      if (pass) hit = d_probe_row(m, d, i, [](uint32_t) { return false; }) != 0;
      bool keep = pass && (m.kind == LDB_JOIN_ANTI ? !hit : hit);
      if (m.has_mark && i < n) mark[i] = hit ? 1 : 0;
      uint64_t mm = __ballot(keep);
      if (lane == 0) {
         bitmap[w] = mm;
         local += (unsigned long long) __popcll(mm);
      }
   }
   if (lane == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
}

// Build-side semi / anti join (the reference's `reverseSides` scheme: the build rows carry a
// boolean flag that matching probe tuples set, then the build buffer is re-scanned on the flag —
// translateHJWithMarker, src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp:1248-1287).
// The flag store is idempotent, so no atomic is needed (the CPU path uses an atomic OR).
__device__ __forceinline__ void join_probe_markbuild_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   uint8_t* flags = gptr_mut<uint8_t>(d->mark);
   if (m.n_ppreds > 0) { // fused filter: tile compaction, then every lane probes
      __shared__ JoinTile st;
      const uint64_t n_tiles = (n + JT_ROWS - 1) / JT_ROWS;
      for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
         const uint64_t base = tile * JT_ROWS;
         d_tile_filter(m, d, base, n, st);
         const uint32_t qn = st.qn;
         for (uint32_t j = threadIdx.x; j < qn; j += JT_BLOCK)
            d_probe_row(m, d, base + st.q[j], [&](uint32_t b) {
               flags[b] = 1;
               return true;
            });
         __syncthreads();
      }
      return;
   }
   const uint64_t tid = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x, nth = (uint64_t) gridDim.x * blockDim.x;
   for (uint64_t i0 = tid; i0 < n; i0 += 2 * nth) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
         const uint64_t i = i0 + (uint64_t) u * nth;
         if (i < n && d_probe_pass(m, d, i))
            d_probe_row(m, d, i, [&](uint32_t b) {
               flags[b] = 1;
               return true;
            });
      }
   }
}
// flags (one byte per build row) → ballot bitmap of the rows to keep (+ their count)
__device__ __forceinline__ void join_flags_bitmap_body(const uint8_t* __restrict__ flags, uint64_t n, int anti, uint64_t* __restrict__ bitmap,
                                                       unsigned long long* __restrict__ counter) {
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   unsigned long long local = 0;
   for (uint64_t w = wave; w < n_words; w += n_waves) {
      const uint64_t i = w * 64 + lane;
      bool keep = i < n && ((flags[i] != 0) != (anti != 0));
      uint64_t mm = __ballot(keep);
      if (lane == 0) {
         bitmap[w] = mm;
         local += (unsigned long long) __popcll(mm);
      }
   }
   if (lane == 0 && local) atomicAdd(counter, local);
}

// Unique build side (primary-key joins: every TPC-H join): a probe row has at most one match, so
// the kernel writes match[i] densely (coalesced) plus a ballot bitmap, and the pairs are produced
// by the ordered bitmap expansion — no atomics on an output cursor, deterministic ascending
// order.  Two words (128 rows) per wave iteration keep two independent probes in flight.
__device__ __forceinline__ void join_probe_unique_body(const DJoin& m, const DJoin* __restrict__ d) {
   if (m.n_ppreds > 0 && m.has_bitmap) { // INNER with a fused filter
      join_probe_unique_filtered_body(m, d);
      return;
   }
   const uint64_t n = d->n_rows;
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6;
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
   uint32_t* match = gptr_mut<uint32_t>(d->match);
   unsigned long long local = 0;
   for (uint64_t w0 = wave; w0 < n_words; w0 += 2 * n_waves) {
      uint32_t brow[2];
      uint64_t rows[2];
      bool pass[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
         const uint64_t w = w0 + (uint64_t) u * n_waves;
         rows[u] = w * 64 + lane;
         pass[u] = w < n_words && rows[u] < n;
      }
      d_eval_conj_batch<2>(m.ppreds, d->ppreds, m.n_ppreds, rows, pass); // fused filter of a lazy probe side
#pragma unroll
      for (int u = 0; u < 2; u++) {
         const uint64_t w = w0 + (uint64_t) u * n_waves;
         const uint64_t i = w * 64 + lane;
         brow[u] = LDB_NULL_ROW;
         if (pass[u]) {
            uint32_t b = LDB_NULL_ROW;
            d_probe_row(m, d, i, [&](uint32_t x) {
               b = x;
               return false;
            });
            brow[u] = b;
         }
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
         const uint64_t w = w0 + (uint64_t) u * n_waves;
         const uint64_t i = w * 64 + lane;
         // INNER reads match[] only at the bitmap's set bits: unmatched rows are not written (a
         // selective join would otherwise stream 4 B per probe row for nothing)
         if (w < n_words && i < n && (brow[u] != LDB_NULL_ROW || !m.has_bitmap)) match[i] = brow[u];
         uint64_t mm = __ballot(brow[u] != LDB_NULL_ROW);
         if (lane == 0 && w < n_words) {
            if (m.has_bitmap) bitmap[w] = mm;
            local += (unsigned long long) __popcll(mm);
         }
      }
   }
   if (lane == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
}
