// ldb_join_kernel.h — device code of the hash-join build / probe kernels.
// Compiled ahead of time (generic) and, for large inputs, at run time with the key metadata and
// join kind as compile-time constants (ldb_jit.hip) — same source (see ldb_gb_kernel.h).
// Replaces (reference): HashIndexedView::build (src/runtime/LazyJoinHashtable.cpp:12-34) and the
// generated probe (LookupHashIndexedViewLowering / ScanListLowering,
// src/compiler/Conversion/SubOpToControlFlow/SubOpToControlFlow.cpp:2558-2586, 2254-2313).
#pragma once
#include "ldb_keys.h"

#define LDB_MAX_RESID 2
#define LDB_MAX_M2PREDS 2
struct DJoinResid {
   DCol pcol; // read at the probe relation's logical row
   DCol bcol; // read at the build relation's logical row
   int32_t op; // ldb_filter_op comparison: pcol OP bcol
   int32_t pad;
};
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
   uint64_t mark2; // uint8_t*: n_m2preds > 0 (build-side semi + anti in one pass): second flag per build row
   uint64_t match; // uint32_t*: unique-build path: build row (or LDB_NULL_ROW) per probe row; pairs path: per-chunk counts / offsets
   uint64_t flags; // uint32_t*: build: [0] |= 1 when two build rows carry the same key (or tag), |= 2 on a long probe run
   int64_t kmin, kmax; // KEY32 + ordered slots: range of the build keys
   uint64_t kmult; // slot = ((key - kmin) * kmult) >> 32
   uint32_t kmult32, ksh; // slot32: slot = mulhi32((key - kmin) << ksh, kmult32) — one 32-bit multiply
   uint64_t key_bits; // uint32_t*: has_key_bits: bit (key - kmin) set ⇔ key is in the table
   uint64_t next; // uint32_t*: chained: next[row] = following row of the same key + 1 (0 = end)
   uint64_t coarse; // const uint32_t*: has_coarse: one bit per 64 key values (any build key in that block?)
   uint32_t coarse_words, pad_c;
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
   // the build keys are known to be unique (verified by the build): a probe stops at its first key match
   int32_t build_unique;
   // residual conjuncts of the join predicate beyond the key equality, each comparing a probe-side
   // column with a build-side column of the candidate pair (Q21's l2.l_suppkey <> l1.l_suppkey): a
   // key match only counts when all of them hold (NULL operands fail).  The reference evaluates them
   // as the filter behind the lookup (SpecializeSubOpPass.cpp:152-205 injects map + filter).
   int32_t n_resid;
   // ordered slots computed in 32-bit arithmetic (capacity <= 2^31 and a key range below 2^32: every table
   // of 32-bit keys up to a billion rows); 64-bit integer multiplies are four quarter-rate instructions
   int32_t slot32;
   // DIRECT addressing (KEY32, key range at most a few times the build rows — every primary key of TPC-H):
   // `slots` is a uint32_t array indexed by (key - kmin) holding build row + 1 (0 = no such key).  No
   // hash, no tag, no collision walk, 4 bytes per key VALUE instead of 8 bytes per slot of a half-empty
   // table; clustered probes (lineitem → orders) sweep it sequentially.  Duplicate keys chain through
   // next[] (`chained`: push-front with one atomic exchange per build row).
   // direct == 2, the RANK-BITMAP table (unique keys): `slots` is a uint64_t array with one word per 32 key VALUES,
   //   low half  = presence bits of the keys (key - kmin) in [32w, 32w + 32),
   //   high half = number of build keys below 32w (exclusive prefix of the popcounts),
   // so a probe is ONE 8-byte load: hit ⇔ its bit is set, and the key's rank = prefix + popcount(bits below it).  The
   // build row of rank r is r itself when the build keys are strictly ascending without NULLs (`rank_sorted`: a
   // primary-key table or a filtered subset of one), else perm[r] (`next` carries perm).  range / 4 bytes in all —
   // 150 MB for the 600 M key values of o_orderkey at SF100, whatever fraction of the orders the build side keeps —
   // against 4 bytes per key VALUE for direct == 1 and 16-32 bytes per build ROW for open addressing.
   int32_t direct;
   int32_t rank_sorted;
   // A selective build side over a small key range (Q17's 20 k of 20 M part keys) leaves almost every probe a miss, yet
   // each miss costs its lane one L2 request for the table word: 600 M random 8-byte requests run at the L2's request
   // rate (~200 G/s, 0.1 of the HBM roofline) although the 5 MB table sits in L2.  `has_coarse`: a bitmap with one bit per
   // 64 key values (20 M keys: 39 KB) is copied into LDS by every workgroup (512 threads: four of them share a CU's 160 KB)
   // and tested first — a clear bit proves the miss without leaving the CU.  Only for probe launches without a fused
   // filter (the tile kernels are written for 256-thread workgroups).
   int32_t has_coarse;
   // TWO 4-byte integer keys (ps_partkey, ps_suppkey; l_suppkey, c_nationkey …) in a hashed table: a slot is a PAIR of words, the usual
   // {hash tag : build row + 1} and behind it the two key values themselves, so a probe verifies its candidate from the 16 bytes it has just
   // loaded instead of gathering two key columns at the build row (Q9: 32 M probes, two random lines less each).  1 = verify from the pair,
   // 2 = the pair layout is there but the probe's key columns are not both 4 bytes wide: verify through the rows as before.
   int32_t pair32;
   DJoinResid resid[LDB_MAX_RESID];
   DPred ppreds[LDB_MAX_PREDS];
   // Build-side semi AND anti join against the same table in ONE pass (TPC-H Q21: EXISTS (l2 …) AND NOT EXISTS (l3 … AND l3.l_receiptdate >
   // l3.l_commitdate) probe the same l_orderkey table with the same residual): a matching pair sets the build row's first flag, and its
   // second flag too when the PROBE row also satisfies these conjuncts — evaluated only for rows that found a partner, so their columns
   // are read for a few per cent of the probe side.  The reference runs two marker joins (translateHJWithMarker, RelAlgToSubOp.cpp:1248-1287).
   int32_t n_m2preds;
   int32_t pad_m2;
   DPred m2preds[LDB_MAX_M2PREDS];
};

extern __shared__ uint32_t ldb_join_lds[]; // dynamic LDS of the probe kernels: the coarse key bitmap (has_coarse), else empty
__device__ __forceinline__ void d_stage_coarse(const DJoin& m, const DJoin* __restrict__ d) {
   if (!m.has_coarse) return;
   const uint32_t* src = gptr<uint32_t>(d->coarse);
   for (uint32_t i = threadIdx.x; i < d->coarse_words; i += blockDim.x) ldb_join_lds[i] = src[i];
   __syncthreads();
}
// (has_coarse holds the filter's granularity: 6 = one bit per 64 key values, the <= 40 KB form; 5 / 4 = one per 32 / 16 — round 6: up to 156 KB, one
// 1024-thread workgroup per CU, for builds too dense for 64-key blocks to be empty — Q9's green parts, 5.4 % of the key range)
__device__ __forceinline__ bool d_coarse_hit(const DJoin& m, uint32_t r) {
   const uint32_t s = (uint32_t) m.has_coarse & 31u; // (has_coarse = 32 | log2 of the key values per bit: non-zero also for the exact filter)
   return (ldb_join_lds[r >> (s + 5u)] >> ((r >> s) & 31u)) & 1u;
}
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
   if (m.key32 && m.ordered_slots) {
      if (m.slot32) return (uint64_t) __umulhi(((uint32_t) key - (uint32_t) d->kmin) << d->ksh, d->kmult32);
      return (((uint64_t) (key - d->kmin) * d->kmult) >> 32) & mask;
   }
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

// the two 4-byte key values of logical row i as one word (pair32; the caller has checked that neither is NULL)
__device__ __forceinline__ unsigned long long d_key_pair32(const KV& keys, uint64_t i) {
   const CV a = keys.col(0), b = keys.col(1);
   const uint32_t ka = (uint32_t) d_load_i64(a, d_phys_row(a, i)), kb = (uint32_t) d_load_i64(b, d_phys_row(b, i));
   return (unsigned long long) ka | ((unsigned long long) kb << 32);
}

__device__ __forceinline__ void join_build_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows, mask = d->cap - 1;
   unsigned long long* slots = gptr_mut<unsigned long long>(d->slots);
   const KV bkeys(m.bkeys, d->bkeys);
   if (m.direct) {
      uint32_t* tab = gptr_mut<uint32_t>(d->slots);
      uint32_t* next = gptr_mut<uint32_t>(d->next);
      const CV c = bkeys.col(0);
      for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
         const uint32_t row = d_phys_row(c, i);
         if (!d_valid(c, row)) continue; // a NULL key can never be matched
         const uint64_t r = (uint64_t) (d_load_i64(c, row) - d->kmin);
         if (m.chained) {
            next[i] = atomicExch(&tab[r], (uint32_t) i + 1u); // push-front: the previous head (or 0) follows this row
         } else {
            const uint32_t old = atomicCAS(&tab[r], 0u, (uint32_t) i + 1u);
            if (old != 0 && m.has_flags) { // a second row with this key: the host rebuilds chained
               uint32_t* f = gptr_mut<uint32_t>(d->flags);
               if ((__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u) == 0) atomicOr(f, 1u);
            }
         }
      }
      return;
   }
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
      const int ss = m.pair32 ? 1 : 0; // pair32: slot p = words 2p (tag : row) and 2p + 1 (the two key values)
      for (;;) {
         unsigned long long old = atomicCAS(&slots[pos << ss], 0ull, (unsigned long long) word);
         if (old == 0) {
            if (m.pair32) slots[(pos << 1) + 1] = d_key_pair32(bkeys, i); // (read by the probe kernels only: a later launch)
            break;
         }
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

// rank-bitmap word → rank + 1 of key offset r (0 = the key is not in the table)
__device__ __forceinline__ uint32_t d_rank_word(uint64_t w, uint32_t r) {
   const uint32_t bits = (uint32_t) w, b = r & 31u;
   if (!((bits >> b) & 1u)) return 0u;
   return (uint32_t) (w >> 32) + (uint32_t) __popc(bits & ((1u << b) - 1u)) + 1u;
}
// rank-bitmap table (DJoin::direct == 2), build pass 1: presence bits into the low halves of the 64-bit words, the
// number of non-NULL build keys (counter[0]; equals the number of set bits iff the keys are unique) and whether the
// keys are strictly ascending without NULLs (flags |= 4 otherwise: then a key's rank is not its row and the
// build adds the rank → row permutation).  Lanes whose keys fall into the same word OR their bits together first.
__device__ __forceinline__ void join_rank_bits_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   const KV bkeys(m.bkeys, d->bkeys);
   const CV c = bkeys.col(0);
   uint32_t* lo = gptr_mut<uint32_t>(d->slots);
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
   unsigned long long present = 0;
   bool unordered = false;
   for (uint64_t base = blockIdx.x * (uint64_t) blockDim.x; base < n; base += stride) { // uniform trip count per block
      const uint64_t i = base + threadIdx.x;
      unsigned long long w = ~0ull;
      uint32_t v = 0;
      bool valid = false;
      if (i < n) {
         const uint32_t row = d_phys_row(c, i);
         if (d_valid(c, row)) {
            const int64_t key = d_load_i64(c, row);
            const uint64_t r = (uint64_t) (key - d->kmin);
            w = r >> 5;
            v = 1u << (r & 31);
            valid = true;
            if (i > 0) {
               const uint32_t prow = d_phys_row(c, i - 1);
               if (!d_valid(c, prow) || !(d_load_i64(c, prow) < key)) unordered = true;
            }
         } else {
            unordered = true;
         }
      }
      present += (unsigned long long) __popcll(__ballot(valid));
      for (int off = 1; off < 64; off <<= 1) {
         const unsigned long long w2 = __shfl_down(w, off);
         const uint32_t v2 = __shfl_down(v, off);
         if (w2 == w) v |= v2;
      }
      const unsigned long long wprev = __shfl_up(w, 1);
      const bool leader = lane == 0 || wprev != w;
      if (leader && w != ~0ull) atomicOr(lo + 2 * w, v);
   }
   if (__ballot(unordered) != 0 && lane == 0) {
      uint32_t* f = gptr_mut<uint32_t>(d->flags);
      if ((__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u) == 0) atomicOr(f, 4u);
   }
   if (lane == 0 && present) atomicAdd(gptr_mut<unsigned long long>(d->counter), present);
}
// pass 3 (only when the keys are not ascending): perm[rank(key of row i)] = i
__device__ __forceinline__ void join_rank_perm_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   const KV bkeys(m.bkeys, d->bkeys);
   const CV c = bkeys.col(0);
   const uint64_t* tab = gptr<uint64_t>(d->slots);
   uint32_t* perm = gptr_mut<uint32_t>(d->next);
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t row = d_phys_row(c, i);
      if (!d_valid(c, row)) continue;
      const uint32_t r = (uint32_t) (d_load_i64(c, row) - d->kmin);
      const uint32_t rk = d_rank_word(tab[r >> 5], r);
      if (rk) perm[rk - 1u] = (uint32_t) i;
   }
}

// residual conjuncts on the candidate pair (probe logical row i, build logical row brow)
__device__ __forceinline__ bool d_resid_ok(const DJoin& m, const DJoin* __restrict__ d, uint64_t i, uint32_t brow) {
   bool ok = true;
   const int nr = m.n_resid;
   LDB_UNROLL
   for (int k = 0; k < nr; k++) {
      if (ok) {
         const CV pc(m.resid[k].pcol, d->resid[k].pcol), bc(m.resid[k].bcol, d->resid[k].bcol);
         const uint32_t pr = d_phys_row(pc, i), br = d_phys_row(bc, brow);
         if (!d_valid(pc, pr) || !d_valid(bc, br)) {
            ok = false;
         } else if (d_is_wide(pc) || d_is_wide(bc)) {
            ok = d_cmp_vals<i128>(m.resid[k].op, d_load_i128(pc, pr), d_load_i128(bc, br));
         } else {
            ok = d_cmp_vals<int64_t>(m.resid[k].op, d_load_i64(pc, pr), d_load_i64(bc, br));
         }
      }
   }
   return ok;
}

// existence kinds over a direct table with key bits: the bit IS the answer (round 6: Q9's 600 M l_partkey probes of the green parts' bits ran one
// dependent chain deep — key, then bit — with the waves waiting 89 % of their cycles; the bit words now travel through the same two-deep pipeline as
// the rank words, DProbePipe)
__device__ __forceinline__ bool d_exists_by_bits(const DJoin& m) {
   return m.direct == 1 && m.has_key_bits && m.n_resid == 0 && (m.kind == LDB_JOIN_SEMI || m.kind == LDB_JOIN_ANTI || m.kind == LDB_JOIN_MARK);
}
// direct-addressed table: resolve the batch from its table words (0 = no such key)
template <int U, typename EMIT>
__device__ __forceinline__ void d_direct_resolve(const DJoin& m, const DJoin* __restrict__ d, const uint64_t (&rows)[U], const bool (&live)[U], const uint32_t (&hw)[U], bool bits_only,
                                                 uint32_t (&matches)[U], EMIT emit) {
#pragma unroll
   for (int u = 0; u < U; u++) {
      if (!live[u] || hw[u] == 0) continue;
      if (bits_only) {
         matches[u]++;
         (void) emit(u, 0u);
      } else if (m.chained) {
         const uint32_t* next = gptr<uint32_t>(d->next);
         for (uint32_t r = hw[u]; r != 0; r = next[r - 1u]) {
            if (m.n_resid == 0 || d_resid_ok(m, d, rows[u], r - 1u)) {
               matches[u]++;
               if (!emit(u, r - 1u)) break;
            }
         }
      } else if (m.n_resid == 0 || d_resid_ok(m, d, rows[u], hw[u] - 1u)) {
         matches[u]++;
         (void) emit(u, hw[u] - 1u);
      }
   }
}

// U probe rows of one lane, phase-separated for memory-level parallelism.  A probe is a chain of
// dependent loads (key → [key bit] → slot → next slot …): one row at a time a wave has ONE such
// chain in flight per lane and the kernel runs at memory LATENCY (600 M FK probes: 112 Grows/s,
// 0.17 of the HBM roofline although it moves no more than the algorithmic bytes).  Here the U key
// loads issue back to back, then the U key-bit loads, then the first TWO slots of every row (the
// second is in the same 64 B line seven times out of eight; slot loads are predicated on the row
// still being alive — their values are not needed before the resolve phase, so no wait separates
// them — key / key-bit loads are branch-free, a dead row reads element 0); only then the rows are resolved, and
// for unique ordered KEY32 tables almost every row resolves from its prefetched slots without
// entering the walk loop.  EMIT(u, build_row) is called per match in the order row-major / slot
// order and returns whether to keep scanning that row.
// `pre` (optional): the batch's keys, already loaded by the caller one iteration ahead (d_prefetch_keys32).
template <int U, typename EMIT>
__device__ __forceinline__ void d_probe_batch(const DJoin& m, const DJoin* __restrict__ d, const uint64_t (&rows)[U], const bool (&act)[U], uint32_t (&matches)[U],
                                              EMIT emit, const uint32_t* pre = nullptr, const uint32_t* prew = nullptr) {
   const KV pkeys(m.pkeys, d->pkeys);
   const uint64_t mask = d->cap - 1;
   const uint64_t* slots = gptr<uint64_t>(d->slots);
   uint64_t pos[U], tag[U];
   bool live[U];
#pragma unroll
   for (int u = 0; u < U; u++) {
      matches[u] = 0;
      live[u] = act[u];
      pos[u] = 0;
      tag[u] = 0;
   }
   if (m.direct && prew) { // the batch's table words came through the pipeline (already 0 for keys outside the table's range)
      uint32_t hw[U];
#pragma unroll
      for (int u = 0; u < U; u++) hw[u] = prew[u];
      d_direct_resolve<U>(m, d, rows, live, hw, d_exists_by_bits(m), matches, emit);
      return;
   }
   if (m.key32) {
      const CV c = pkeys.col(0);
      // a 4-byte integer probe column (every TPC-H foreign key) stays in 32-bit registers
      const bool narrow = c.m.width == 4 && c.m.type != LDB_T_FLOAT32;
      uint32_t k32[U];
      if (!c.m.rowids && !c.m.validity) {
         // branch-free (a dead row reads row 0; callers guarantee n >= 1): a predicated load would
         // need its value at the join point — one wait per load instead of one per batch
#pragma unroll
         for (int u = 0; u < U; u++) {
            const uint32_t row = act[u] ? (uint32_t) rows[u] : 0u;
            if (narrow && pre) {
               k32[u] = pre[u];
            } else if (narrow) {
               k32[u] = (uint32_t) gptr<int32_t>(c.p.values)[row];
            } else {
               const int64_t kv = d_load_i64(c, row);
               if (kv != (int64_t) (int32_t) kv) live[u] = false; // a wider probe value can equal no 32-bit build key
               k32[u] = (uint32_t) kv;
            }
         }
      } else {
         uint32_t pr[U];
#pragma unroll
         for (int u = 0; u < U; u++) pr[u] = act[u] ? d_phys_row(c, rows[u]) : LDB_NULL_ROW;
#pragma unroll
         for (int u = 0; u < U; u++) {
            k32[u] = 0;
            if (act[u]) {
               if (d_valid(c, pr[u])) {
                  const int64_t kv = d_load_i64(c, pr[u]);
                  if (kv != (int64_t) (int32_t) kv) live[u] = false;
                  k32[u] = (uint32_t) kv;
               } else {
                  live[u] = false; // a NULL key matches nothing
               }
            }
         }
      }
#pragma unroll
      for (int u = 0; u < U; u++) tag[u] = (uint64_t) k32[u];
      if (m.ordered_slots || m.direct) {
         // (key - kmin) as an unsigned 32-bit offset: keys below kmin wrap to values above the span
         const uint32_t kmin32 = (uint32_t) d->kmin, span = (uint32_t) (d->kmax - d->kmin);
         uint32_t r[U];
#pragma unroll
         for (int u = 0; u < U; u++) {
            r[u] = k32[u] - kmin32;
            live[u] = live[u] && r[u] <= span; // outside the build key range
            if (m.has_coarse) live[u] = live[u] && d_coarse_hit(m, live[u] ? r[u] : 0u); // no build key in its block of key values (LDS)
         }
         if (m.has_key_bits) {
            const uint32_t* bits = gptr<uint32_t>(d->key_bits);
            uint32_t bw[U];
#pragma unroll
            for (int u = 0; u < U; u++) bw[u] = bits[live[u] ? r[u] >> 5 : 0u];
#pragma unroll
            for (int u = 0; u < U; u++) live[u] = live[u] && ((bw[u] >> (r[u] & 31u)) & 1u); // not a build key
         }
         if (m.direct) {
#pragma unroll
            for (int u = 0; u < U; u++) pos[u] = live[u] ? (uint64_t) r[u] : 0;
         } else if (m.slot32) {
            const uint32_t kmult32 = d->kmult32, ksh = d->ksh;
#pragma unroll
            for (int u = 0; u < U; u++) pos[u] = live[u] ? (uint64_t) __umulhi(r[u] << ksh, kmult32) : 0;
         } else {
            const uint64_t kmult = d->kmult;
#pragma unroll
            for (int u = 0; u < U; u++) pos[u] = live[u] ? ((((uint64_t) r[u] * kmult) >> 32) & mask) : 0;
         }
      } else {
#pragma unroll
         for (int u = 0; u < U; u++)
            if (live[u]) pos[u] = d_join_slot(m, d, d_hash_keys(pkeys, rows[u]), 0, mask);
      }
   } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
         if (act[u]) {
            bool nul;
            const uint64_t h = d_hash_keys(pkeys, rows[u], &nul);
            live[u] = !nul;
            tag[u] = h >> 32;
            pos[u] = nul ? 0 : d_join_slot(m, d, h, 0, mask);
         }
      }
   }
   if (m.direct == 2) {
      const uint64_t* tab = gptr<uint64_t>(d->slots);
      const uint32_t* perm = gptr<uint32_t>(d->next);
      const bool bits_only = m.n_resid == 0 && (m.kind == LDB_JOIN_SEMI || m.kind == LDB_JOIN_ANTI || m.kind == LDB_JOIN_MARK);
      uint64_t ww[U];
      uint32_t hw[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
         ww[u] = 0;
         if (live[u]) ww[u] = tab[pos[u] >> 5];
      }
#pragma unroll
      for (int u = 0; u < U; u++) hw[u] = live[u] ? d_rank_word(ww[u], (uint32_t) pos[u]) : 0u;
      if (!m.rank_sorted && !bits_only) {
#pragma unroll
         for (int u = 0; u < U; u++)
            if (hw[u]) hw[u] = perm[hw[u] - 1u] + 1u;
      }
      d_direct_resolve<U>(m, d, rows, live, hw, bits_only, matches, emit);
      return;
   }
   if (m.direct) {
      // one 4-byte word per key value: build row + 1 (the head of the key's chain when `chained`)
      const uint32_t* tab = gptr<uint32_t>(d->slots);
      uint32_t hw[U];
      // existence kinds over a table with key bits: the bit already answered, the build row is never used
      const bool bits_only = m.has_key_bits && m.n_resid == 0 && (m.kind == LDB_JOIN_SEMI || m.kind == LDB_JOIN_ANTI || m.kind == LDB_JOIN_MARK);
#pragma unroll
      for (int u = 0; u < U; u++) {
         hw[u] = 0;
         if (!live[u]) continue;
         hw[u] = bits_only ? 1u : tab[pos[u]];
      }
      d_direct_resolve<U>(m, d, rows, live, hw, bits_only, matches, emit);
      return;
   }
   // pair32 (two 4-byte keys): slot p = words 2p (tag : build row + 1) and 2p + 1 (the two key values) — a candidate is verified against
   // the second word of the 16 bytes its first word came with, instead of two key columns gathered at the build row
   const int ss = (m.pair32 && !m.chained) ? 1 : 0;
   const bool pair_cmp = m.pair32 == 1 && !m.chained;
   unsigned long long mine[U];
#pragma unroll
   for (int u = 0; u < U; u++) mine[u] = 0;
   // the first two slots of every live row
   uint64_t w0[U], w1[U];
#pragma unroll
   for (int u = 0; u < U; u++) {
#ifndef JOIN_LOAD_DEAD
      w0[u] = w1[u] = 0;
      if (!live[u]) continue; // (predicated by EXEC: the live lanes' loads still issue back to back)
#endif
      w0[u] = slots[pos[u] << ss];
#ifndef JOIN_NO_PREFETCH2
      // the second slot only for tables with duplicate keys (their rows sit in consecutive slots): for a
      // unique table it is needed after a collision only, and fetching it always costs an unclustered
      // probe 40 % (600 M random probes: 18.3 → 13.0 ms) for 5 % on a clustered one
      if (!m.build_unique) w1[u] = slots[((pos[u] + 1) & mask) << ss];
#endif
      if (pair_cmp) mine[u] = d_key_pair32(pkeys, rows[u]);
   }
   const KV bkeys(m.bkeys, d->bkeys);
#pragma unroll
   for (int u = 0; u < U; u++) {
      if (!live[u]) continue;
      uint64_t p = pos[u], w = w0[u];
      uint32_t step = 0;
      for (;;) {
         if (w == 0) break;
         if ((w >> 32) == tag[u] && (m.key32 || (pair_cmp ? slots[(p << 1) + 1] == mine[u] : d_keys_equal(bkeys, (uint64_t) ((uint32_t) w - 1u), pkeys, rows[u], false)))) {
            if (m.chained) { // the key's only slot: its rows are the chain
               const uint32_t* next = gptr<uint32_t>(d->next);
               for (uint32_t r = (uint32_t) w; r != 0; r = next[r - 1u]) {
                  if (m.n_resid == 0 || d_resid_ok(m, d, rows[u], r - 1u)) {
                     matches[u]++;
                     if (!emit(u, r - 1u)) break;
                  }
               }
               break;
            }
            bool stop = false;
            if (m.n_resid == 0 || d_resid_ok(m, d, rows[u], (uint32_t) w - 1u)) {
               matches[u]++;
               stop = !emit(u, (uint32_t) w - 1u);
            }
#ifndef JOIN_NO_UNIQUE_BREAK
            if (stop || m.build_unique) break;
#else
            if (stop) break;
#endif
         }
         p = (p + 1) & mask;
         step++;
#ifndef JOIN_NO_PREFETCH2
         w = (step == 1 && !m.build_unique) ? w1[u] : slots[p << ss];
#else
         w = slots[p << ss];
#endif
      }
   }
}

// one probe row (the kinds that have no batch to offer)
template <typename EMIT>
__device__ __forceinline__ uint32_t d_probe_row(const DJoin& m, const DJoin* __restrict__ d, uint64_t i, EMIT emit) {
   const uint64_t rows[1] = {i};
   const bool act[1] = {true};
   uint32_t mt[1];
   d_probe_batch<1>(m, d, rows, act, mt, [&](int, uint32_t b) { return emit(b); });
   return mt[0];
}

// first matching build row of each batch row (LDB_NULL_ROW = none): what the unique-build, SEMI /
// ANTI / MARK kinds need — no side effects inside the walk, the callers store afterwards
template <int U>
__device__ __forceinline__ void d_probe_first(const DJoin& m, const DJoin* __restrict__ d, const uint64_t (&rows)[U], const bool (&act)[U], uint32_t (&brow)[U],
                                              const uint32_t* pre = nullptr, const uint32_t* prew = nullptr) {
   uint32_t mt[U];
#pragma unroll
   for (int u = 0; u < U; u++) brow[u] = LDB_NULL_ROW;
   d_probe_batch<U>(
      m, d, rows, act, mt,
      [&](int u, uint32_t b) {
         brow[u] = b;
         return false;
      },
      pre, prew);
}

// Software pipelining of the probe's first dependent load.  A wave's iterations are serial chains
// key → [key bit] → slot → resolve, and the PMC passes over the FK-probe micro-benchmark
// (profiles/r02_pmc_probe.txt) show the waves parked on s_waitcnt for 83 % of their cycles while the
// memory side moves only 2 TB/s: latency, not bandwidth.  The keys of the wave's NEXT batch are
// therefore loaded before the current batch is probed, so they travel while its slots do — for the
// common shape, a dense 4-byte key column without a fused filter (rows past the end read row 0).
__device__ __forceinline__ bool d_keys_prefetchable(const DJoin& m) {
   return m.key32 && m.n_ppreds == 0 && m.pkeys.cols[0].width == 4 && m.pkeys.cols[0].type != LDB_T_FLOAT32 && !m.pkeys.cols[0].rowids && !m.pkeys.cols[0].validity;
}
template <int U>
__device__ __forceinline__ void d_prefetch_keys32(const DJoin* __restrict__ d, uint64_t row0, uint64_t n, uint32_t (&k)[U]) {
   // row0 is wave-uniform: the tile's base and the rows left from there are scalar, a lane only compares its
   // 32-bit offset (rows past the end read the tile's first row — or row 0 when the whole tile lies past the end)
   const uint64_t base = row0 < n ? row0 : 0;
   const uint64_t left = n - base;
   const uint32_t left32 = left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) left;
   const int32_t* kt = gptr<int32_t>(d->pkeys.cols[0].values) + base;
   const uint32_t lane = threadIdx.x & 63;
#pragma unroll
   for (int u = 0; u < U; u++) {
      const uint32_t o = (uint32_t) u * 64u + lane;
      // (a probe key is read once: a non-temporal load keeps the stream from evicting the table words / key bits the L2 is there for)
      k[u] = (uint32_t) __builtin_nontemporal_load(&kt[o < left32 ? o : 0u]);
   }
}
// The per-wave software pipeline of the probe loops.  A probe is the dependent chain key → table word →
// resolve; a wave that walks it tile by tile spends one memory round trip per link.  With a dense 4-byte key
// column the keys of the NEXT tile are loaded while the current one is probed (`pf`), and for a direct
// table without key bits the chain is pipelined two deep (`pw`): in the step of tile s the table words of
// tile s+1 and the keys of tile s+2 are issued BEFORE tile s is resolved, so every load has a whole
// iteration to arrive and no wait drains the queue (vmcnt counts in order; what a step waits for is older
// than the 2U loads it just issued).  Issue order over the steps: … keys(s+1), words(s) | keys(s+2), words(s+1) | …
//   LDB_PIN(x): x must be in its register HERE — the compiler puts the s_waitcnt of the producing load at
//   this point, not earlier, and moves no memory access across.  Without the pins the scheduler rotates the
//   loop back into issue-then-consume.
//   Two sets of word registers alternate (template parameter P of step): copying a register whose load is
//   still in flight would wait for it, so the loops are unrolled by two instead of rotating registers.
#define LDB_PIN(x) asm volatile("" : "+v"(x))
#define LDB_PIN64(x) asm volatile("" : "+v"(x)) // (a 64-bit value: a VGPR pair)
template <int P>
struct DPar {
   static constexpr int v = P;
};
template <int U>
struct DProbePipe {
   // (pf / pw are recomputed from the — compile-time — metadata at every use: as members they would keep the
   // struct in memory and the specialised kernel could not fold them)
   static __device__ __forceinline__ bool pf(const DJoin& m) { return d_keys_prefetchable(m); }
   static __device__ __forceinline__ bool bits(const DJoin& m) { return d_exists_by_bits(m); }
   static __device__ __forceinline__ bool pw(const DJoin& m) { return d_keys_prefetchable(m) && ((m.direct == 1 && !m.has_key_bits) || (m.direct == 2 && m.rank_sorted) || bits(m)); }
   uint64_t stride; // rows between a wave's consecutive tiles
   uint32_t ck[U], cw[U]; // pf: keys of the tile being resolved;  pw: its table words
   uint32_t k[U]; // keys in flight (pf: of the coming tile; pw: of the tile after the coming one once its step ran)
   uint64_t w[2][U]; // pw: table words in flight, alternating sets (direct == 1 uses the low half)
   uint32_t off[2][U]; // pw: the keys' table offsets (key - kmin), ~0 = outside the table's range
   // the table offsets of the keys in k[]; the words are loaded by words_at
   __device__ __forceinline__ void offsets_of(const DJoin& m, const DJoin* __restrict__ d, int set) {
      const bool has_coarse = m.has_coarse != 0;
      const uint32_t kmin32 = (uint32_t) d->kmin, span = (uint32_t) (d->kmax - d->kmin);
#pragma unroll
      for (int u = 0; u < U; u++) {
         const uint32_t r = k[u] - kmin32;
         off[set][u] = r <= span ? r : 0xFFFFFFFFu;
         if (has_coarse && r <= span && !d_coarse_hit(m, r)) off[set][u] = 0xFFFFFFFFu; // a proven miss never asks the L2 for its word
      }
   }
   __device__ __forceinline__ void words_at(const DJoin& m, const DJoin* __restrict__ d, int set) {
#pragma unroll
      for (int u = 0; u < U; u++) {
         const uint32_t r = off[set][u] == 0xFFFFFFFFu ? 0u : off[set][u];
         if (m.direct == 2) w[set][u] = gptr<uint64_t>(d->slots)[r >> 5];
         else if (bits(m)) w[set][u] = (uint64_t) gptr<uint32_t>(d->key_bits)[r >> 5];
         else w[set][u] = (uint64_t) gptr<uint32_t>(d->slots)[r];
      }
   }
   __device__ __forceinline__ void start(const DJoin& m, const DJoin* __restrict__ d, uint64_t row0, uint64_t stride_rows, uint64_t n) {
      stride = stride_rows;
#pragma unroll
      for (int u = 0; u < U; u++) {
         ck[u] = cw[u] = k[u] = 0;
         w[0][u] = w[1][u] = 0;
         off[0][u] = off[1][u] = 0xFFFFFFFFu;
      }
      if (pf(m)) d_prefetch_keys32<U>(d, row0, n, k);
      if (pw(m)) {
         offsets_of(m, d, 0);
         d_prefetch_keys32<U>(d, row0 + stride, n, k);
         words_at(m, d, 0);
      }
   }
   // at the top of the iteration over the tile starting at row0: afterwards ck / cw belong to that tile.
   // Steps alternate P = 0, 1, 0, … starting with 0.
   template <int P>
   __device__ __forceinline__ void step(const DJoin& m, const DJoin* __restrict__ d, uint64_t row0, uint64_t n, DPar<P>) {
      if (pw(m)) {
#pragma unroll
         for (int u = 0; u < U; u++) LDB_PIN(k[u]); // keys(s+1) are here (words(s) may still be in flight)
         offsets_of(m, d, 1 - P);
         d_prefetch_keys32<U>(d, row0 + 2 * stride, n, k); // keys(s+2), into the registers just consumed
         asm volatile("" : : : "memory"); // (keeps the key loads ahead of the word loads in issue order; no wait)
         words_at(m, d, 1 - P); // words(s+1)
#pragma unroll
         for (int u = 0; u < U; u++) LDB_PIN64(w[P][u]); // words(s) are here
#pragma unroll
         for (int u = 0; u < U; u++) { // → build row + 1 of every row of the tile (0 = no partner)
            if (off[P][u] == 0xFFFFFFFFu) cw[u] = 0u;
            else if (bits(m)) cw[u] = ((uint32_t) w[P][u] >> (off[P][u] & 31u)) & 1u; // 1 = "some build row" (existence kinds never ask which)
            else cw[u] = m.direct == 2 ? d_rank_word(w[P][u], off[P][u]) : (uint32_t) w[P][u];
         }
      } else if (pf(m)) {
#pragma unroll
         for (int u = 0; u < U; u++) ck[u] = k[u];
         d_prefetch_keys32<U>(d, row0 + stride, n, k);
      }
   }
   __device__ __forceinline__ const uint32_t* keys(const DJoin& m) const { return pf(m) && !pw(m) ? ck : nullptr; }
   __device__ __forceinline__ const uint32_t* words(const DJoin& m) const { return pw(m) ? cw : nullptr; }
};

// a wave's tiles first, first + stride, … < bound.  With the two-deep pipeline (pw) two tiles per trip, so that its
// word registers alternate without copies; otherwise one tile per trip (three inlined copies of every tile body
// cost the other shapes registers — 49 → 65 VGPRs on the pairs kernel, one wave per SIMD less — for nothing)
template <int U, typename TILE>
__device__ __forceinline__ void d_tile_loop(const DJoin& m, uint64_t first, uint64_t bound, uint64_t stride, TILE tile) {
   if (DProbePipe<U>::pw(m)) {
      uint64_t t0 = first;
      for (; t0 + stride < bound; t0 += 2 * stride) {
         tile(t0, DPar<0>{});
         tile(t0 + stride, DPar<1>{});
      }
      if (t0 < bound) tile(t0, DPar<0>{});
   } else {
      for (uint64_t t0 = first; t0 < bound; t0 += stride) tile(t0, DPar<0>{});
   }
}
// global wave number as a wave-uniform (scalar) value: the tile loops then run on scalar control flow
__device__ __forceinline__ uint64_t d_wave_id() {
   return (uint64_t) blockIdx.x * (blockDim.x >> 6) + (uint64_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
}
#ifndef JOIN_BATCH
#define JOIN_BATCH 4 // probe rows in flight per lane (8 costs more in occupancy than it gains: 3.2 vs 4.0 ms on 600 M FK probes)
#endif

// INNER / LEFT_OUTER / SINGLE with possibly duplicated build keys, two passes and no atomics on
// an output cursor (one contended cursor address serialises at ~10 ns per wave-append in the
// L2: Q12's 150 M-row probe spent 22 ms there).  Pass 1 counts the rows every 64-row chunk
// produces, a device scan turns the counts into chunk offsets, pass 2 re-walks (the slot runs are
// now cache-resident) and every lane writes its pairs at chunk offset + in-wave prefix: output in
// probe order, sized exactly.  A wave handles JP_U consecutive chunks per iteration (batched probe).
#define JP_U 4
// rows each probe row of the batch contributes: its matches, or one NULL-padded row when an outer join finds none
template <int U>
__device__ __forceinline__ void d_pairs_of_rows(const DJoin& m, const DJoin* __restrict__ d, const uint64_t (&rows)[U], bool (&act)[U], uint32_t (&cnt)[U],
                                                const uint32_t* pre = nullptr, const uint32_t* prew = nullptr) {
   d_eval_conj_batch<U>(m.ppreds, d->ppreds, m.n_ppreds, rows, act); // (the host only fuses filters for INNER here)
   d_probe_batch<U>(m, d, rows, act, cnt, [&](int, uint32_t) { return m.kind != LDB_JOIN_SINGLE; }, pre, prew);
#pragma unroll
   for (int u = 0; u < U; u++)
      if (act[u] && m.kind != LDB_JOIN_INNER && cnt[u] == 0) cnt[u] = 1u;
}
__device__ __forceinline__ void join_probe_pairs_count_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   d_stage_coarse(m, d); // (a no-op unless the launch brought a coarse key bitmap)
   const uint64_t n_chunks = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   uint32_t* chunk_cnt = gptr_mut<uint32_t>(d->match);
   unsigned long long total = 0; // exact 64-bit row count (the 32-bit offsets cannot detect > 4 G rows)
   DProbePipe<JP_U> pipe;
   pipe.start(m, d, wave * JP_U * 64, n_waves * JP_U * 64, n);
   auto tile = [&](uint64_t w0, auto par) __attribute__((always_inline)) {
      uint64_t rows[JP_U];
      bool act[JP_U];
      uint32_t c[JP_U];
#pragma unroll
      for (int u = 0; u < JP_U; u++) {
         rows[u] = (w0 + u) * 64 + lane;
         act[u] = rows[u] < n;
      }
      pipe.step(m, d, w0 * 64, n, par);
      d_pairs_of_rows<JP_U>(m, d, rows, act, c, pipe.keys(m), pipe.words(m));
#pragma unroll
      for (int u = 0; u < JP_U; u++) {
         uint32_t s = c[u];
         for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
         if (lane == 0 && w0 + u < n_chunks) chunk_cnt[w0 + u] = s;
         total += s;
      }
   };
   d_tile_loop<JP_U>(m, wave * JP_U, n_chunks, n_waves * JP_U, tile);
   if (lane == 0 && total) atomicAdd(gptr_mut<unsigned long long>(d->counter), total);
}
__device__ __forceinline__ void join_probe_pairs_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   d_stage_coarse(m, d); // (a no-op unless the launch brought a coarse key bitmap)
   const uint64_t n_chunks = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   const uint32_t* chunk_off = gptr<uint32_t>(d->match);
   uint32_t* out_probe = gptr_mut<uint32_t>(d->out_probe);
   uint32_t* out_build = gptr_mut<uint32_t>(d->out_build);
   const uint64_t out_cap = d->out_cap; // (= the count pass's total; smaller only when that total was a replayed one that no longer holds: never write past the buffers)
   for (uint64_t w0 = wave * JP_U; w0 < n_chunks; w0 += n_waves * JP_U) {
      uint64_t rows[JP_U], at[JP_U];
      bool act[JP_U];
      uint32_t c[JP_U], emitted[JP_U];
#pragma unroll
      for (int u = 0; u < JP_U; u++) {
         rows[u] = (w0 + u) * 64 + lane;
         act[u] = rows[u] < n;
      }
      d_pairs_of_rows<JP_U>(m, d, rows, act, c);
#pragma unroll
      for (int u = 0; u < JP_U; u++) {
         uint32_t incl = c[u];
         for (int off = 1; off < 64; off <<= 1) {
            uint32_t up = __shfl_up(incl, off);
            if (lane >= (uint32_t) off) incl += up;
         }
         at[u] = (uint64_t) (w0 + u < n_chunks ? chunk_off[w0 + u] : 0u) + (incl - c[u]);
         emitted[u] = 0;
         act[u] = act[u] && c[u] != 0;
      }
      uint32_t again[JP_U];
      d_probe_batch<JP_U>(m, d, rows, act, again, [&](int u, uint32_t brow) {
         if (at[u] + emitted[u] < out_cap) {
            out_probe[at[u] + emitted[u]] = (uint32_t) rows[u];
            out_build[at[u] + emitted[u]] = brow;
         }
         emitted[u]++;
         return m.kind != LDB_JOIN_SINGLE;
      });
#pragma unroll
      for (int u = 0; u < JP_U; u++) {
         if (act[u] && emitted[u] == 0 && at[u] < out_cap) { // unmatched (or NULL-key) probe row of an outer join
            out_probe[at[u]] = (uint32_t) rows[u];
            out_build[at[u]] = LDB_NULL_ROW;
         }
      }
   }
}

// count matches only — the probe micro-benchmark kernel (Grows/s).  A wave owns tiles of
// 64 x JOIN_BATCH consecutive rows: the batch's key loads are JOIN_BATCH coalesced 256-byte
// segments next to each other and, for clustered keys, its slots share cache lines.
__device__ __forceinline__ void join_probe_count_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   d_stage_coarse(m, d); // (a no-op unless the launch brought a coarse key bitmap)
   unsigned long long local = 0;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   const uint64_t n_tiles = (n + 64 * JOIN_BATCH - 1) / (64 * JOIN_BATCH);
   DProbePipe<JOIN_BATCH> pipe;
   pipe.start(m, d, wave * JOIN_BATCH * 64, n_waves * JOIN_BATCH * 64, n);
   auto tile = [&](uint64_t t, auto par) __attribute__((always_inline)) {
      uint64_t rows[JOIN_BATCH];
      bool act[JOIN_BATCH];
      uint32_t mt[JOIN_BATCH];
#pragma unroll
      for (int u = 0; u < JOIN_BATCH; u++) {
         rows[u] = (t * JOIN_BATCH + u) * 64 + lane;
         act[u] = rows[u] < n;
      }
      pipe.step(m, d, t * JOIN_BATCH * 64, n, par);
      d_eval_conj_batch<JOIN_BATCH>(m.ppreds, d->ppreds, m.n_ppreds, rows, act);
      d_probe_batch<JOIN_BATCH>(m, d, rows, act, mt, [](int, uint32_t) { return true; }, pipe.keys(m), pipe.words(m));
#pragma unroll
      for (int u = 0; u < JOIN_BATCH; u++) local += mt[u];
   };
   d_tile_loop<JOIN_BATCH>(m, wave, n_tiles, n_waves, tile);
   for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
   if (lane == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter) + 1, local);
}

// ---------------------------------------------------------------- fused filter: LDS-staged compaction
// With a fused filter a lane whose row fails would idle while its neighbours walk the table, and a
// probe is latency-bound: throughput is proportional to the lanes actually probing (54 % for
// Q3's lineitem side).  So a workgroup first filters a TILE of JT_ROWS consecutive rows
// (predicate-major, JT_U rows per thread, coalesced), compacts the survivors' tile-relative row
// numbers into an LDS queue (ballot + popcount rank, one LDS atomic per wave and batch row), and
// then probes from the queue with every lane busy, JT_PB queue entries per lane at a time (batched
// probe).  Results that must stay in row order go through a per-tile bitmap in LDS (JT_ROWS / 64
// words) that is written out once per tile.
#define JT_BLOCK 256
#define JT_U 8
#define JT_ROWS (JT_BLOCK * JT_U)
#ifndef JT_PB
#define JT_PB 4
#endif
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
// probe the tile's queue, JT_PB entries per lane at a time; ON_MATCH(tile-relative row, build row) → keep scanning?
// AFTER(tile-relative row, #matches) runs once per queued row
template <typename ON_MATCH, typename AFTER>
__device__ __forceinline__ void d_tile_probe(const DJoin& m, const DJoin* __restrict__ d, uint64_t base, JoinTile& st, ON_MATCH on_match, AFTER after) {
#ifdef JT_EXTRA_SYNC
   __syncthreads();
#endif
   const uint32_t qn = st.qn;
   for (uint32_t j0 = 0; j0 < qn; j0 += JT_PB * JT_BLOCK) {
      uint64_t rows[JT_PB];
      uint32_t rel[JT_PB], mt[JT_PB];
      bool act[JT_PB];
#pragma unroll
      for (int u = 0; u < JT_PB; u++) {
         const uint32_t j = j0 + (uint32_t) u * JT_BLOCK + threadIdx.x;
         act[u] = j < qn;
         rel[u] = act[u] ? st.q[j] : 0u;
         rows[u] = base + rel[u];
      }
      d_probe_batch<JT_PB>(m, d, rows, act, mt, [&](int u, uint32_t b) { return on_match(rel[u], b); });
#pragma unroll
      for (int u = 0; u < JT_PB; u++)
         if (act[u]) after(rel[u], mt[u]);
   }
}
// the same for the kinds that only need each queued row's FIRST match: AFTER(tile-relative row, build row | LDB_NULL_ROW)
template <typename AFTER>
__device__ __forceinline__ void d_tile_probe_first(const DJoin& m, const DJoin* __restrict__ d, uint64_t base, JoinTile& st, AFTER after) {
   const uint32_t qn = st.qn;
   for (uint32_t j0 = 0; j0 < qn; j0 += JT_PB * JT_BLOCK) {
      uint64_t rows[JT_PB];
      uint32_t rel[JT_PB], brow[JT_PB];
      bool act[JT_PB];
#pragma unroll
      for (int u = 0; u < JT_PB; u++) {
         const uint32_t j = j0 + (uint32_t) u * JT_BLOCK + threadIdx.x;
         act[u] = j < qn;
         rel[u] = act[u] ? st.q[j] : 0u;
         rows[u] = base + rel[u];
      }
      d_probe_first<JT_PB>(m, d, rows, act, brow);
#ifdef JOIN_DEBUG_COUNTS
      {
         unsigned long long* dc = gptr_mut<unsigned long long>(d->counter);
         unsigned long long q = 0, hit = 0, inr = 0;
#pragma unroll
         for (int u = 0; u < JT_PB; u++) {
            q += act[u] ? 1 : 0;
            hit += (act[u] && brow[u] != LDB_NULL_ROW) ? 1 : 0;
            if (act[u]) {
               const CV c = KV(m.pkeys, d->pkeys).col(0);
               const int64_t kv = d_load_i64(c, (uint32_t) rows[u]);
               if (kv >= d->kmin && kv <= d->kmax) {
                  const uint64_t r = (uint64_t) (kv - d->kmin);
                  inr += (gptr<uint32_t>(d->key_bits)[r >> 5] >> (r & 31)) & 1u;
               }
            }
         }
         atomicAdd(dc + 2, q);
         atomicAdd(dc + 3, inr);
         atomicAdd(dc + 4, hit);
      }
#endif
#pragma unroll
      for (int u = 0; u < JT_PB; u++)
         if (act[u]) after(rel[u], brow[u]);
   }
}
// write the tile's LDS bitmap out and count its bits; ends with a barrier
__device__ __forceinline__ void d_tile_flush(const DJoin* __restrict__ d, uint64_t tile, uint64_t n_words, JoinTile& st, unsigned long long& local) {
   uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
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
__device__ __forceinline__ void d_block_count_add(const DJoin* __restrict__ d, unsigned long long local) {
   if (threadIdx.x < 64) {
      for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
      if (threadIdx.x == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
   }
}

__device__ __forceinline__ void join_probe_unique_filtered_body(const DJoin& m, const DJoin* __restrict__ d) {
   __shared__ JoinTile st;
   const uint64_t n = d->n_rows;
   const uint64_t n_words = (n + 63) / 64, n_tiles = (n + JT_ROWS - 1) / JT_ROWS;
   uint32_t* match = gptr_mut<uint32_t>(d->match);
   unsigned long long local = 0;
   for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      const uint64_t base = tile * JT_ROWS;
      d_tile_filter(m, d, base, n, st);
      d_tile_probe_first(m, d, base, st, [&](uint32_t r, uint32_t b) {
         if (b != LDB_NULL_ROW) {
            match[base + r] = b;
            atomicOr(&st.bm[r >> 6], 1ull << (r & 63));
         }
      });
      d_tile_flush(d, tile, n_words, st, local);
   }
   d_block_count_add(d, local);
}

// SEMI / ANTI / MARK: existence per probe row → bitmap word per wave (rows in ascending order)
#define JE_U 4
__device__ __forceinline__ void join_probe_exists_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   d_stage_coarse(m, d); // (a no-op unless the launch brought a coarse key bitmap)
   if (m.n_ppreds > 0) { // fused filter: tile compaction (never MARK: the host forces those)
      __shared__ JoinTile st;
      const uint64_t n_words = (n + 63) / 64, n_tiles = (n + JT_ROWS - 1) / JT_ROWS;
      unsigned long long local = 0;
      for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
         const uint64_t base = tile * JT_ROWS;
         d_tile_filter(m, d, base, n, st);
         d_tile_probe_first(m, d, base, st, [&](uint32_t r, uint32_t b) {
            if (m.kind == LDB_JOIN_ANTI ? b == LDB_NULL_ROW : b != LDB_NULL_ROW) atomicOr(&st.bm[r >> 6], 1ull << (r & 63));
         });
         d_tile_flush(d, tile, n_words, st, local);
      }
      d_block_count_add(d, local);
      return;
   }
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
   uint8_t* mark = gptr_mut<uint8_t>(d->mark);
   unsigned long long local = 0;
   DProbePipe<JE_U> pipe;
   pipe.start(m, d, wave * JE_U * 64, n_waves * JE_U * 64, n);
   auto tile = [&](uint64_t w0, auto par) __attribute__((always_inline)) {
      uint64_t rows[JE_U];
      bool act[JE_U];
      uint32_t first[JE_U];
#pragma unroll
      for (int u = 0; u < JE_U; u++) {
         rows[u] = (w0 + u) * 64 + lane;
         act[u] = rows[u] < n;
      }
      pipe.step(m, d, w0 * 64, n, par);
      d_probe_first<JE_U>(m, d, rows, act, first, pipe.keys(m), pipe.words(m));
#pragma unroll
      for (int u = 0; u < JE_U; u++) {
         const bool hit = first[u] != LDB_NULL_ROW;
         const bool keep = act[u] && (m.kind == LDB_JOIN_ANTI ? !hit : hit);
         if (m.has_mark && act[u]) mark[rows[u]] = hit ? 1 : 0;
         const uint64_t mm = __ballot(keep);
         if (lane == 0 && w0 + u < n_words) {
            bitmap[w0 + u] = mm;
            local += (unsigned long long) __popcll(mm);
         }
      }
   };
   d_tile_loop<JE_U>(m, wave * JE_U, n_words, n_waves * JE_U, tile);
   if (lane == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
}

// Build-side semi / anti join (the reference's `reverseSides` scheme: the build rows carry a
// boolean flag that matching probe tuples set, then the build buffer is re-scanned on the flag —
// translateHJWithMarker, src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp:1248-1287).
// The flag store is idempotent, so no atomic is needed (the CPU path uses an atomic OR).
__device__ __forceinline__ void join_probe_markbuild_body(const DJoin& m, const DJoin* __restrict__ d) {
   const uint64_t n = d->n_rows;
   d_stage_coarse(m, d); // (a no-op unless the launch brought a coarse key bitmap)
   uint8_t* flags = gptr_mut<uint8_t>(d->mark);
   if (m.n_ppreds > 0) { // fused filter: tile compaction, then every lane probes
      __shared__ JoinTile st;
      const uint64_t n_tiles = (n + JT_ROWS - 1) / JT_ROWS;
      for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
         const uint64_t base = tile * JT_ROWS;
         d_tile_filter(m, d, base, n, st);
         d_tile_probe(
            m, d, base, st,
            [&](uint32_t, uint32_t b) {
               flags[b] = 1;
               return true;
            },
            [](uint32_t, uint32_t) {});
         __syncthreads();
      }
      return;
   }
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   const uint64_t n_tiles = (n + 64 * JE_U - 1) / (64 * JE_U);
   DProbePipe<JE_U> pipe;
   pipe.start(m, d, wave * JE_U * 64, n_waves * JE_U * 64, n);
   auto tile = [&](uint64_t t, auto par) __attribute__((always_inline)) {
      uint64_t rows[JE_U];
      bool act[JE_U];
      uint32_t mt[JE_U];
#pragma unroll
      for (int u = 0; u < JE_U; u++) {
         rows[u] = (t * JE_U + u) * 64 + lane;
         act[u] = rows[u] < n;
      }
      pipe.step(m, d, t * JE_U * 64, n, par);
      d_probe_batch<JE_U>(
         m, d, rows, act, mt,
         [&](int u, uint32_t b) {
            flags[b] = 1;
            if (m.n_m2preds > 0) { // the second marker: this probe row also satisfies the extra conjuncts
               bool also = true;
               LDB_UNROLL
               for (int p = 0; p < m.n_m2preds; p++)
                  if (also) also = d_eval_pred(PV(m.m2preds[p], d->m2preds[p]), rows[u]);
               if (also) gptr_mut<uint8_t>(d->mark2)[b] = 1;
            }
            return true;
         },
         pipe.keys(m), pipe.words(m));
   };
   d_tile_loop<JE_U>(m, wave, n_tiles, n_waves, tile);
}
// flags (one byte per build row) → ballot bitmap of the rows to keep (+ their count)
__device__ __forceinline__ void join_flags_bitmap_body(const uint8_t* __restrict__ flags, uint64_t n, int anti, uint64_t* __restrict__ bitmap,
                                                       unsigned long long* __restrict__ counter, const uint8_t* __restrict__ not_flags = nullptr) {
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   unsigned long long local = 0;
   for (uint64_t w = wave; w < n_words; w += n_waves) {
      const uint64_t i = w * 64 + lane;
      bool keep = i < n && ((flags[i] != 0) != (anti != 0));
      if (not_flags && keep) keep = not_flags[i] == 0; // semi + anti in one pass: the first marker set, the second not
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
// order.  JE_U consecutive words (256 rows) per wave iteration, probed as one batch.
__device__ __forceinline__ void join_probe_unique_body(const DJoin& m, const DJoin* __restrict__ d) {
   d_stage_coarse(m, d); // (round 6: also in front of the tile kernels — a small filter, <= 16 KB, beside the tile's queue; the host decides)
   if (m.n_ppreds > 0 && m.has_bitmap) { // INNER with a fused filter
      join_probe_unique_filtered_body(m, d);
      return;
   }
   const uint64_t n = d->n_rows;
   const uint64_t n_words = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const uint64_t wave = d_wave_id();
   const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   uint64_t* bitmap = gptr_mut<uint64_t>(d->bitmap);
   uint32_t* match = gptr_mut<uint32_t>(d->match);
   unsigned long long local = 0;
   DProbePipe<JE_U> pipe;
   pipe.start(m, d, wave * JE_U * 64, n_waves * JE_U * 64, n);
   auto tile = [&](uint64_t w0, auto par) __attribute__((always_inline)) {
      uint32_t brow[JE_U];
      uint64_t rows[JE_U];
      bool act[JE_U], pass[JE_U];
#pragma unroll
      for (int u = 0; u < JE_U; u++) {
         rows[u] = (w0 + u) * 64 + lane;
         act[u] = pass[u] = rows[u] < n;
      }
      pipe.step(m, d, w0 * 64, n, par);
      d_eval_conj_batch<JE_U>(m.ppreds, d->ppreds, m.n_ppreds, rows, pass); // fused filter of a lazy probe side
      d_probe_first<JE_U>(m, d, rows, pass, brow, pipe.keys(m), pipe.words(m));
#pragma unroll
      for (int u = 0; u < JE_U; u++) {
         // INNER reads match[] only at the bitmap's set bits: unmatched rows are not written (a
         // selective join would otherwise stream 4 B per probe row for nothing)
         if (act[u] && (brow[u] != LDB_NULL_ROW || !m.has_bitmap)) match[rows[u]] = brow[u];
         const uint64_t mm = __ballot(brow[u] != LDB_NULL_ROW);
         if (lane == 0 && w0 + u < n_words) {
            if (m.has_bitmap) bitmap[w0 + u] = mm;
            local += (unsigned long long) __popcll(mm);
         }
      }
   };
   d_tile_loop<JE_U>(m, wave * JE_U, n_words, n_waves * JE_U, tile);
   if (lane == 0 && local) atomicAdd(gptr_mut<unsigned long long>(d->counter), local);
}
