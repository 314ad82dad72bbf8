// ldb_gb_kernel.h — device code of the fused scan + predicate + hash group-by kernel.
// Compiled twice: ahead of time by hipcc (generic: the descriptor is read from memory) and, for
// large inputs, at run time by hiprtc with the descriptor's metadata baked in as a constexpr
// (ldb_jit.hip) — the same hand-written source, the compiler folds every type/op switch and
// unrolls every descriptor loop.  This mirrors what the reference does with LLVM: its per-tuple
// pipeline code is generated per query (SubOpToControlFlow.cpp) and JIT-compiled
// (src/execution/LLVMBackends.cpp:219-406).
//
// Replaces (reference): LookupPreAggrHtFragment + ReduceOpLowering
// (src/compiler/Conversion/SubOpToControlFlow/SubOpToControlFlow.cpp:3065-3157, 3719-3768),
// PreAggregationHashtableFragment::insert / PreAggregationHashtable::merge
// (src/runtime/PreAggregationHashtable.cpp:46-60, 76-158), SimpleState (src/runtime/SimpleState.cpp:8-30).
#pragma once
#include "ldb_keys.h"

#define GB_BLOCK 256
// Rows per thread per loop iteration = gb_body's template argument.  The specialised kernel keeps
// only the referenced columns in registers, so it can afford several rows in flight: the host
// picks DGroupBy::batch_rows (8 for a multi-column filter chain — every conjunct column costs a
// dependent memory round trip per batch, so deeper batches amortise it: Q6 2.1 → 1.8 ms — else 4);
// the generic kernel indexes its value cache dynamically and stays at 1.
// 1: evaluate each conjunct for the whole row batch (loads first, then compares);
// 0: evaluate the conjunction row by row (a dependent load → compare chain per row)
#ifndef GB_PRED_BATCH
#define GB_PRED_BATCH 1
#endif
#define GB_MAX_COLS 12
#define GB_MAX_ACCS 20
#define GB_MAX_CPREDS 6
#define GB_MAX_OUT 16
#define GB_MAX_WORDS 24

enum { ACC_SUM64 = 0,
       ACC_SUM128 = 1,
       ACC_COUNT = 2,
       ACC_MIN64 = 3,
       ACC_MAX64 = 4,
       ACC_SUMF64 = 5,
       ACC_MINF64 = 6,
       ACC_MAXF64 = 7,
       ACC_MIN128 = 8, // three words: low, high (signed), lock
       ACC_MAX128 = 9 };

struct DFactorG {
   int32_t has_col;
   int32_t col_idx; // into DGroupBy::cols
   int64_t a, b;
};
struct DTermG {
   // fits64 (host: GbBuilder::conv_expr): every factor and the whole product provably fit 63 bits — from the widths the columns are STORED at (a column
   // narrowed to w bytes holds |v| < 2^(8w-1)) and their decimal precisions — so the product is formed with 64-bit multiplies and only the
   // accumulation is 128 bits wide.  Same value as the wrapping 128-bit product whenever the proof holds (round 6: the compressed resident format
   // makes Q1 ALU-bound, and its time is the two 128 x 128-bit products of sum_disc_price / sum_charge)
   int32_t n_factors, negate, div_pow10, fits64;
   DFactorG f[LDB_MAX_FACTORS];
};
struct DExprG {
   int32_t n_terms, is_float;
   DTermG t[LDB_MAX_TERMS];
};
struct DAcc {
   int32_t kind;
   int32_t word; // first accumulator word
   int32_t n_cpreds;
   int32_t cpred[LDB_MAX_AGG_PREDS]; // indexes into DGroupBy::cpreds
   int32_t count_rows; // ACC_COUNT: 1 = count rows (COUNT(*)), 0 = count non-NULL expr
   int32_t pad;
   DExprG e;
};
struct DOut { // one output aggregate column
   int32_t fn; // ldb_agg_fn
   int32_t acc; // value accumulator
   int32_t cnt_acc; // AVG divisor / validity counter (-1: always valid)
   int32_t wide;
   int32_t avg_pow10;
   int32_t out_width; // bytes per output value
   int32_t is_float;
   // conditional SUM (case when p then x else 0 end): a row failing p contributes a non-NULL 0, so
   // the result is NULL only if every row passes p with a NULL x: valid ⇔ rows − pass + nonnull > 0
   int32_t cnt_rows_acc; // all rows of the group (-1: not needed)
   int32_t cnt_pass_acc; // rows passing the aggregate's predicates (-1: argument not nullable)
   int32_t pad;
   uint64_t out_values; // device address
   uint64_t out_valid; // one byte per group (packed later) or 0
   DExprG e; // ANY: evaluated on the representative row
};
struct DGroupBy {
   // ---- run-time part (never specialised on)
   uint64_t n_rows;
   uint64_t g_cap; // global capacity (pow2)
   uint64_t g_keys; // uint64_t*
   uint64_t g_acc; // uint64_t*: word w of slot p at g_acc[w * g_cap + p]
   uint64_t g_flags; // uint32_t*: [0] = overflow
   uint32_t lds_slots, lds_reps; // S (pow2), R (pow2)
   int64_t kmin; // ordered_slots: smallest key value
   uint64_t kmult; // ordered_slots: global slot = ((key - kmin) * kmult) >> 32
   // ---- metadata (+ the addresses inside the DCols, which are run-time too)
   int32_t n_preds, n_cols, n_accs, n_words, n_cpreds, n_outs;
   int32_t keyless, use_lds;
   int32_t batch_rows; // rows per thread per iteration of the specialised kernel (1, 2, 4 or 8)
   // single integer key, high-cardinality mode: a group's global slot follows the key's position in
   // the column's value range instead of its hash — input clustered on the key (lineitem on
   // l_orderkey) then walks the table almost sequentially, and the groups come out in key order.
   // Long probe runs (skewed keys) raise flag bit 1 and the host retries hashed.
   int32_t ordered_slots;
   // single key over a column that is SORTED (a cached column statistic), no filter: equal keys
   // are adjacent, so group g is simply the g-th key change.  Pass 1 counts the key changes per
   // 64-row chunk, a scan numbers the groups, pass 2 (this kernel) reduces each run in the wave and
   // writes it to its dense slot — a run that starts and ends inside the wave with a PLAIN store,
   // only runs crossing a wave boundary with atomics.  No hash table, no probing: Q18's 150 M-group
   // GROUP BY l_orderkey needs ~2 atomics per 64 rows instead of 2 per group.
   int32_t dense_sorted;
   // single NOT NULL integer key whose value range fits the table: slot = key - kmin, no slot word,
   // no probing, no key verification — a row costs its accumulator atomics only (the hashed / ordered
   // paths add a random slot read and a random read of the representative row's key: Q13's 148 M-row
   // count per customer 11.4 → see DESIGN.md).  Occupancy = the unconditional row counter
   // `direct_word`; the output key column is written from the slot number by k_gb_finalize.
   int32_t direct;
   uint64_t chunk_off; // uint32_t*: dense_sorted: number of groups that start before each 64-row chunk
   // dense_sorted with in-kernel finalisation (dense_out): a group that begins and ends inside one wave never touches a table — its head
   // lane reduces the run and writes the group's FINAL output values and its representative row; only groups that cross a 64-row chunk
   // boundary are accumulated with atomics, in a table with ONE slot per chunk (slot = the chunk that holds the group's first row; g_acc,
   // g_cap = number of chunks), and finished by k_gb_finalize_cross from the per-chunk flags.  Q18 (600 M rows → 150 M groups): no slot
   // word, no 16-byte accumulator per group written and read back twice, no occupancy scan (DESIGN §2 Group-by).
   int32_t dense_out;
   // dense_out: 1 / 2 = the group's first row also writes its KEY into the output key column (direct_keys_out, direct_key_width) — no gather of
   // representative rows afterwards; 2 = and no representative rows at all (no ANY aggregate needs them)
   int32_t dense_keys;
   uint64_t rep_rows_out; // uint32_t*: dense_out: representative row of every group (written by the group's first row)
   uint64_t cross_flags; // uint8_t*, one per chunk, pre-zeroed: 1 = the group that begins in this chunk continues into the next one
   uint64_t dense_groups; // dense_sorted: number of output slots (a group number beyond it — possible only under a mis-speculated replay — is dropped)
   uint64_t direct_keys_out; // device address of the output key column (run-time)
   int32_t direct_word; // accumulator word of the row counter
   int32_t direct_key_width; // 4, 8 or 16 bytes per output key
   DKeys keys;
   DPred preds[LDB_MAX_PREDS];
   DPred cpreds[GB_MAX_CPREDS];
   DCol cols[GB_MAX_COLS];
   DAcc accs[GB_MAX_ACCS];
   uint64_t word_init[GB_MAX_WORDS];
   DOut outs[GB_MAX_OUT];
};

// ---------------------------------------------------------------- expression evaluation
// value cache: the low 64 bits of every referenced narrow column, loaded once per row into a
// register array (a bare array: only array allocas are promoted to registers; generic kernels
// index it with s_set_gpr_idx by the wave-uniform col_idx, specialised ones statically).
typedef long long RowVals[GB_MAX_COLS];

__device__ __forceinline__ void d_load_vals(const DGroupBy& m, const DGroupBy* __restrict__ d, uint64_t i, RowVals& rv, uint32_t& rvalid) {
   rvalid = 0; // bit c = column c non-NULL
   const int nc = m.n_cols;
#pragma unroll
   for (int c = 0; c < GB_MAX_COLS; c++) {
      long long x = 0;
      if (c < nc) {
         const CV col(m.cols[c], d->cols[c]);
         uint32_t row = d_phys_row(col, i);
         if (d_valid(col, row)) {
            rvalid |= 1u << c;
            if (!d_is_flt(col)) x = d_load_i64(col, row);
            else x = __double_as_longlong(d_load_f64(col, row));
         }
      }
      rv[c] = x;
   }
}

// Σ_t ± Π_f (a + b*col) / 10^k in wrapping 128-bit arithmetic (DecimalMulOpLowering /
// DecimalBinOpLowering, reference LowerToStd.cpp:653-699).  Returns false when a referenced
// column is NULL.
__device__ __forceinline__ bool d_eval_int(const DGroupBy& m, const DGroupBy* __restrict__ d, const DExprG& e, const RowVals& rv, uint32_t rvalid,
                                           uint64_t i, i128* out) {
   u128 total = 0;
   const int nt = e.n_terms;
   LDB_UNROLL
   for (int t = 0; t < nt; t++) {
      const DTermG& tm = e.t[t];
      u128 prod = 1;
      const int nf = tm.n_factors;
      if (tm.fits64) { // (no wide column among the factors: the host checked)
         long long p64 = 1;
         LDB_UNROLL
         for (int f = 0; f < nf; f++) {
            const DFactorG& fa = tm.f[f];
            long long v = fa.a;
            if (fa.has_col) {
               const int ci = fa.col_idx;
               if (!((rvalid >> ci) & 1)) return false;
               v += fa.b * rv[ci];
            }
            p64 = f == 0 ? v : p64 * v;
         }
         i128 wide = (i128) p64;
         if (tm.div_pow10 > 0) wide = d_sdiv128(wide, d_pow10(tm.div_pow10));
         total = tm.negate ? total - (u128) wide : total + (u128) wide;
         continue;
      }
      LDB_UNROLL
      for (int f = 0; f < nf; f++) {
         const DFactorG& fa = tm.f[f];
         i128 v = (i128) fa.a;
         if (fa.has_col) {
            const int ci = fa.col_idx;
            if (!((rvalid >> ci) & 1)) return false;
            const CV col(m.cols[ci], d->cols[ci]);
            if (d_is_wide(col)) v = (i128) ((u128) v + (u128) (i128) fa.b * (u128) d_load_i128(col, d_phys_row(col, i)));
            else v += (i128) fa.b * (i128) rv[ci]; // 64x64 → 128, exact
         }
         prod = f == 0 ? (u128) v : prod * (u128) v;
      }
      if (tm.div_pow10 > 0) prod = (u128) d_sdiv128((i128) prod, d_pow10(tm.div_pow10));
      total = tm.negate ? total - prod : total + prod;
   }
   *out = (i128) total;
   return true;
}
__device__ __forceinline__ bool d_eval_flt(const DGroupBy& m, const DExprG& e, const RowVals& rv, uint32_t rvalid, double* out) {
   double total = 0;
   const int nt = e.n_terms;
   LDB_UNROLL
   for (int t = 0; t < nt; t++) {
      const DTermG& tm = e.t[t];
      double prod = 1;
      const int nf = tm.n_factors;
      LDB_UNROLL
      for (int f = 0; f < nf; f++) {
         const DFactorG& fa = tm.f[f];
         double v = (double) fa.a;
         if (fa.has_col) {
            const int ci = fa.col_idx;
            if (!((rvalid >> ci) & 1)) return false;
            const bool colf = m.cols[ci].type == LDB_T_FLOAT64 || m.cols[ci].type == LDB_T_FLOAT32;
            double x = colf ? __longlong_as_double(rv[ci]) : (double) rv[ci];
            v += (double) fa.b * x;
         }
         prod *= v;
      }
      total = tm.negate ? total - prod : total + prod;
   }
   *out = total;
   return true;
}

// ---------------------------------------------------------------- accumulator sinks
// double min/max via CAS on the bit pattern
__device__ __forceinline__ void d_atomic_minmax_f64(unsigned long long* p, double v, bool is_min) {
   unsigned long long old = *p;
   for (;;) {
      double cur = __longlong_as_double((long long) old);
      bool better = is_min ? v < cur : v > cur;
      if (!better) return;
      unsigned long long prev = atomicCAS(p, old, (unsigned long long) __double_as_longlong(v));
      if (prev == old) return;
      old = prev;
   }
}

struct Sink {
   unsigned long long* base; // word 0 of this slot
   uint64_t stride; // distance between consecutive words of one slot
   bool plain = false; // this lane is the only contributor of its slot: store instead of atomics (run combining only)
   __device__ __forceinline__ unsigned long long* w(int k) const { return base + (uint64_t) k * stride; }
};
// `v` folded into an identity-initialised word by its only contributor, or atomically
__device__ __forceinline__ void d_sink_add64(const Sink& s, int word, unsigned long long v) {
   if (s.plain) *s.w(word) = v;
   else atomicAdd(s.w(word), v);
}

// 128-bit SUM as two 64-bit words: the carry out of the low word is the only cross-word traffic
__device__ __forceinline__ void d_sink_add128(const Sink& s, int word, u128 v) {
   unsigned long long lo = (unsigned long long) v, hi = (unsigned long long) (v >> 64);
   if (s.plain) { // only contributor of a zero-initialised slot
      *s.w(word) = lo;
      *s.w(word + 1) = hi;
      return;
   }
   unsigned long long old = atomicAdd(s.w(word), lo);
   hi += (unsigned long long) (old + lo < old);
   if (hi) atomicAdd(s.w(word + 1), hi);
}

// 128-bit MIN / MAX: gfx950 has no 128-bit atomic, and the two halves cannot be lowered independently
// (the low word only means something next to its high word), so the pair is updated under a
// per-slot lock word.  A lane that spins on a lock while another lane OF THE SAME WAVE holds it would
// never let the holder run (the wave executes one side of a branch at a time), so the lanes of a
// wave that need a lock take turns: one leader at a time spins, updates and releases while the others
// wait at the reconvergence point holding nothing.  A value whose high word is already worse than the
// slot's (which only ever improves) is rejected without the lock.
// (the reference's reduce function is a signed compare-and-select on i128 values inside the
// single-writer fragment of a thread; MergePreAggrHashMap combines fragments, SubOpToControlFlow.cpp:1861-1938)
__device__ __forceinline__ void d_sink_minmax128(const Sink& s, int word, i128 v, bool is_min) {
   const unsigned long long lo = (unsigned long long) (u128) v;
   const long long hi = (long long) (v >> 64);
   if (s.plain) { // only contributor of an identity-initialised slot
      *s.w(word) = lo;
      *s.w(word + 1) = (unsigned long long) hi;
      return;
   }
   const long long seen = (long long) __hip_atomic_load(s.w(word + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   if (is_min ? hi > seen : hi < seen) return;
   unsigned long long* lock = s.w(word + 2);
   const unsigned int lane = __lane_id();
   for (unsigned long long turn = __ballot(1); turn; turn &= turn - 1) {
      if (lane != (unsigned int) (__ffsll((long long) turn) - 1)) continue;
      while (atomicCAS(lock, 0ull, 1ull) != 0ull) __builtin_amdgcn_s_sleep(1);
      __threadfence();
      const unsigned long long clo = __hip_atomic_load(s.w(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long chi = (long long) __hip_atomic_load(s.w(word + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const i128 cur = (i128) (((u128) (unsigned long long) chi << 64) | clo);
      if (is_min ? v < cur : v > cur) {
         __hip_atomic_store(s.w(word), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store(s.w(word + 1), (unsigned long long) hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __threadfence();
      atomicExch(lock, 0ull);
   }
}
#define GB_I128_MAX ((i128) ((((u128) 0x7FFFFFFFFFFFFFFFull) << 64) | (u128) 0xFFFFFFFFFFFFFFFFull))
#define GB_I128_MIN ((i128) (((u128) 0x8000000000000000ull) << 64))

// fold one input row into the accumulators of its group
__device__ __forceinline__ void d_accumulate(const DGroupBy& m, const DGroupBy* __restrict__ d, const RowVals& rv, uint32_t rvalid, uint64_t i,
                                             const Sink& s) {
   const int na = m.n_accs;
   LDB_UNROLL
   for (int a = 0; a < na; a++) {
      const DAcc& acc = m.accs[a];
      bool pass = true;
      LDB_UNROLL
      for (int p = 0; p < acc.n_cpreds; p++)
         if (pass) pass = d_eval_pred(PV(m.cpreds[acc.cpred[p]], d->cpreds[acc.cpred[p]]), i);
      if (!pass) continue; // sum(case when p then x else 0 end) adds 0
      if (acc.kind == ACC_COUNT && acc.count_rows) {
         atomicAdd(s.w(acc.word), 1ull);
         continue;
      }
      if (acc.e.is_float) {
         double fv;
         if (!d_eval_flt(m, acc.e, rv, rvalid, &fv)) continue;
         switch (acc.kind) {
            case ACC_COUNT: atomicAdd(s.w(acc.word), 1ull); break;
            case ACC_SUMF64: atomicAdd((double*) s.w(acc.word), fv); break;
            case ACC_MINF64: d_atomic_minmax_f64(s.w(acc.word), fv, true); break;
            default: d_atomic_minmax_f64(s.w(acc.word), fv, false); break;
         }
         continue;
      }
      i128 v;
      if (!d_eval_int(m, d, acc.e, rv, rvalid, i, &v)) continue;
      switch (acc.kind) {
         case ACC_COUNT: atomicAdd(s.w(acc.word), 1ull); break;
         case ACC_SUM64: atomicAdd(s.w(acc.word), (unsigned long long) v); break; // i64 wrap = SUM in the argument type
         case ACC_SUM128: d_sink_add128(s, acc.word, (u128) v); break;
         case ACC_MIN64: atomicMin((long long*) s.w(acc.word), (long long) v); break;
         case ACC_MIN128: d_sink_minmax128(s, acc.word, v, true); break;
         case ACC_MAX128: d_sink_minmax128(s, acc.word, v, false); break;
         default: atomicMax((long long*) s.w(acc.word), (long long) v); break;
      }
   }
}

// ---------------------------------------------------------------- run combining (high-cardinality path)
// When the LDS table is off (more groups than it can hold) every row would pay a global atomic
// per accumulator — ~15 G atomics/s on random addresses, 40 ms for Q18's 600 M rows.  Fact tables
// are clustered on their key (lineitem by l_orderkey), so neighbouring lanes very often carry the
// SAME group: lanes whose key equals the previous lane's form a run, the run is reduced inside the
// wave with shuffles and only its head lane touches the global table (one lookup + one atomic per
// accumulator per run).  Exact: integer adds commute; float sums are reassociated (as atomics
// already do).  Every lane of the wave executes the shuffles; lanes without a row carry the
// identity.
__device__ __forceinline__ unsigned long long d_shfl_down(unsigned long long v, int off) { return __shfl_down(v, off); }
__device__ __forceinline__ long long d_shfl_down(long long v, int off) { return __shfl_down(v, off); }
__device__ __forceinline__ double d_shfl_down(double v, int off) { return __shfl_down(v, off); }
__device__ __forceinline__ u128 d_shfl_down(u128 v, int off) {
   unsigned long long lo = __shfl_down((unsigned long long) v, off), hi = __shfl_down((unsigned long long) (v >> 64), off);
   return ((u128) hi << 64) | lo;
}
// reduction of v over lanes [lane, run_end] (run_end = last lane of this lane's run)
template <typename T, typename OP>
__device__ __forceinline__ T d_seg_reduce(T v, uint32_t lane, uint32_t run_end, OP op) {
#pragma unroll
   for (int off = 1; off < 64; off <<= 1) {
      T o = d_shfl_down(v, off);
      if (lane + (uint32_t) off <= run_end) v = op(v, o);
   }
   return v;
}

#define GB_I64_MAX 0x7FFFFFFFFFFFFFFFll
#define GB_I64_MIN (-0x7FFFFFFFFFFFFFFFll - 1)
// d_accumulate for a whole run: `pass` = this lane has a row of the run, `apply` = this lane is the
// run's head and owns a valid slot `s`.  Must be called by all lanes of the wave.
__device__ __forceinline__ void d_accumulate_runs(const DGroupBy& m, const DGroupBy* __restrict__ d, const RowVals& rv, uint32_t rvalid, uint64_t i, bool pass,
                                                  bool apply, uint32_t lane, uint32_t run_end, const Sink& s) {
   const int na = m.n_accs;
   auto add_u64 = [](unsigned long long a, unsigned long long b) { return a + b; };
   LDB_UNROLL
   for (int a = 0; a < na; a++) {
      const DAcc& acc = m.accs[a];
      bool ok = pass;
      LDB_UNROLL
      for (int p = 0; p < acc.n_cpreds; p++)
         if (ok) ok = d_eval_pred(PV(m.cpreds[acc.cpred[p]], d->cpreds[acc.cpred[p]]), i);
      if (acc.kind == ACC_COUNT && acc.count_rows) {
         unsigned long long c = d_seg_reduce<unsigned long long>(ok ? 1ull : 0ull, lane, run_end, add_u64);
         if (apply && c) d_sink_add64(s, acc.word, c);
         continue;
      }
      if (acc.e.is_float) {
         double fv = 0;
         if (ok) ok = d_eval_flt(m, acc.e, rv, rvalid, &fv);
         switch (acc.kind) {
            case ACC_COUNT: {
               unsigned long long c = d_seg_reduce<unsigned long long>(ok ? 1ull : 0ull, lane, run_end, add_u64);
               if (apply && c) d_sink_add64(s, acc.word, c);
               break;
            }
            case ACC_SUMF64: {
               unsigned long long c = d_seg_reduce<unsigned long long>(ok ? 1ull : 0ull, lane, run_end, add_u64);
               double r = d_seg_reduce<double>(ok ? fv : 0.0, lane, run_end, [](double x, double y) { return x + y; });
               if (apply && c) {
                  if (s.plain) *(double*) s.w(acc.word) = r;
                  else atomicAdd((double*) s.w(acc.word), r);
               }
               break;
            }
            case ACC_MINF64: {
               double r = d_seg_reduce<double>(ok ? fv : __builtin_inf(), lane, run_end, [](double x, double y) { return y < x ? y : x; });
               if (apply && r != __builtin_inf()) {
                  if (s.plain) *(double*) s.w(acc.word) = r;
                  else d_atomic_minmax_f64(s.w(acc.word), r, true);
               }
               break;
            }
            default: {
               double r = d_seg_reduce<double>(ok ? fv : -__builtin_inf(), lane, run_end, [](double x, double y) { return y > x ? y : x; });
               if (apply && r != -__builtin_inf()) {
                  if (s.plain) *(double*) s.w(acc.word) = r;
                  else d_atomic_minmax_f64(s.w(acc.word), r, false);
               }
               break;
            }
         }
         continue;
      }
      i128 v = 0;
      if (ok) ok = d_eval_int(m, d, acc.e, rv, rvalid, i, &v);
      switch (acc.kind) {
         case ACC_COUNT: {
            unsigned long long c = d_seg_reduce<unsigned long long>(ok ? 1ull : 0ull, lane, run_end, add_u64);
            if (apply && c) d_sink_add64(s, acc.word, c);
            break;
         }
         case ACC_SUM64: {
            unsigned long long r = d_seg_reduce<unsigned long long>(ok ? (unsigned long long) v : 0ull, lane, run_end, add_u64);
            if (apply && r) d_sink_add64(s, acc.word, r);
            break;
         }
         case ACC_SUM128: {
            u128 r = d_seg_reduce<u128>(ok ? (u128) v : (u128) 0, lane, run_end, [](u128 x, u128 y) { return x + y; });
            if (apply && r) d_sink_add128(s, acc.word, r);
            break;
         }
         case ACC_MIN64: {
            long long r = d_seg_reduce<long long>(ok ? (long long) v : GB_I64_MAX, lane, run_end, [](long long x, long long y) { return y < x ? y : x; });
            if (apply && r != GB_I64_MAX) {
               if (s.plain) *(long long*) s.w(acc.word) = r;
               else atomicMin((long long*) s.w(acc.word), r);
            }
            break;
         }
         case ACC_MIN128: {
            u128 r = d_seg_reduce<u128>(ok ? (u128) v : (u128) GB_I128_MAX, lane, run_end, [](u128 x, u128 y) { return (i128) y < (i128) x ? y : x; });
            if (apply && (i128) r != GB_I128_MAX) d_sink_minmax128(s, acc.word, (i128) r, true);
            break;
         }
         case ACC_MAX128: {
            u128 r = d_seg_reduce<u128>(ok ? (u128) v : (u128) GB_I128_MIN, lane, run_end, [](u128 x, u128 y) { return (i128) y > (i128) x ? y : x; });
            if (apply && (i128) r != GB_I128_MIN) d_sink_minmax128(s, acc.word, (i128) r, false);
            break;
         }
         default: {
            long long r = d_seg_reduce<long long>(ok ? (long long) v : GB_I64_MIN, lane, run_end, [](long long x, long long y) { return y > x ? y : x; });
            if (apply && r != GB_I64_MIN) {
               if (s.plain) *(long long*) s.w(acc.word) = r;
               else atomicMax((long long*) s.w(acc.word), r);
            }
            break;
         }
      }
   }
}

// merge accumulator words of an LDS slot into the global slot (combine step of
// MergePreAggrHashMap, reference SubOpToControlFlow.cpp:1861-1938)
__device__ __forceinline__ void d_combine(const DGroupBy& m, const Sink& src, const Sink& dst) {
   const int na = m.n_accs;
   LDB_UNROLL
   for (int a = 0; a < na; a++) {
      const DAcc& acc = m.accs[a];
      unsigned long long x = *src.w(acc.word);
      switch (acc.kind) {
         case ACC_COUNT:
         case ACC_SUM64:
            if (x) atomicAdd(dst.w(acc.word), x);
            break;
         case ACC_SUM128: {
            unsigned long long hi = *src.w(acc.word + 1);
            d_sink_add128(dst, acc.word, ((u128) hi << 64) | x);
            break;
         }
         case ACC_MIN64: atomicMin((long long*) dst.w(acc.word), (long long) x); break;
         case ACC_MAX64: atomicMax((long long*) dst.w(acc.word), (long long) x); break;
         case ACC_MIN128:
         case ACC_MAX128: {
            const i128 v = (i128) (((u128) *src.w(acc.word + 1) << 64) | x);
            if (v != (acc.kind == ACC_MIN128 ? GB_I128_MAX : GB_I128_MIN)) d_sink_minmax128(dst, acc.word, v, acc.kind == ACC_MIN128);
            break;
         }
         case ACC_SUMF64:
            if (__longlong_as_double((long long) x) != 0.0) atomicAdd((double*) dst.w(acc.word), __longlong_as_double((long long) x));
            break;
         case ACC_MINF64: d_atomic_minmax_f64(dst.w(acc.word), __longlong_as_double((long long) x), true); break;
         default: d_atomic_minmax_f64(dst.w(acc.word), __longlong_as_double((long long) x), false); break;
      }
   }
}

// ---------------------------------------------------------------- a group's output row
// The output aggregates of ONE group from its accumulator words (word w at acc[w * stride]): the last step of the reference's
// aggregation (the scan over the merged hash table that feeds the result columns, SubOpToControlFlow.cpp:1861-1938 + the AVG / validity
// arithmetic of LowerToStd.cpp:631-651).  Called by k_gb_finalize for table slots, and by the dense_sorted kernel itself for groups
// that live inside one wave (acc = the head lane's private words, stride 1).  `g` = output position, `rep` = representative input row.
__device__ __forceinline__ void d_finalize_group(const DGroupBy& m, const DGroupBy* __restrict__ d, const unsigned long long* acc, uint64_t stride, uint64_t g, uint32_t rep) {
   const int no = m.n_outs;
   LDB_UNROLL
   for (int o = 0; o < no; o++) {
      const DOut& out = m.outs[o];
      const uint64_t out_values = d->outs[o].out_values, out_valid = d->outs[o].out_valid;
      bool ok = true;
      unsigned long long cnt = 0;
      if (out.cnt_acc >= 0) {
         cnt = acc[(uint64_t) m.accs[out.cnt_acc].word * stride];
         ok = cnt != 0;
      }
      if (out.cnt_rows_acc >= 0) { // conditional SUM: rows failing the predicates contribute a non-NULL 0
         unsigned long long rows = acc[(uint64_t) m.accs[out.cnt_rows_acc].word * stride];
         unsigned long long passing = out.cnt_pass_acc >= 0 ? acc[(uint64_t) m.accs[out.cnt_pass_acc].word * stride] : 0;
         unsigned long long nonnull = out.cnt_pass_acc >= 0 ? cnt : 0; // not nullable: no passing row is NULL
         ok = rows - passing + nonnull != 0;
      }
      if (out.is_float) {
         double v = 0;
         if (out.fn == LDB_AGG_COUNT || out.fn == LDB_AGG_COUNT_STAR) {
            // counts are integers; handled below
         } else if (out.fn == LDB_AGG_ANY) {
            RowVals rv;
            uint32_t rvalid = 0;
            if (d->n_rows) d_load_vals(m, d, rep, rv, rvalid);
            ok = d->n_rows != 0 && d_eval_flt(m, out.e, rv, rvalid, &v);
         } else {
            v = __longlong_as_double((long long) acc[(uint64_t) m.accs[out.acc].word * stride]);
            if (out.fn == LDB_AGG_AVG && ok) v = v / (double) cnt;
         }
         if (out.fn != LDB_AGG_COUNT && out.fn != LDB_AGG_COUNT_STAR) {
            ((double*) out_values)[g] = ok ? v : 0.0;
            if (out_valid) ((uint8_t*) out_valid)[g] = ok ? 1 : 0;
            continue;
         }
      }
      i128 v = 0;
      switch (out.fn) {
         case LDB_AGG_COUNT:
         case LDB_AGG_COUNT_STAR:
            v = (i128) acc[(uint64_t) m.accs[out.acc].word * stride];
            ok = true;
            break;
         case LDB_AGG_ANY: {
            if (d->n_rows == 0) { // key-less aggregation over no rows: the pre-seeded group has no representative row
               ok = false;
               break;
            }
            RowVals rv;
            uint32_t rvalid;
            d_load_vals(m, d, rep, rv, rvalid);
            ok = d_eval_int(m, d, out.e, rv, rvalid, rep, &v);
            break;
         }
         default: {
            const DAcc& a = m.accs[out.acc];
            uint64_t lo = acc[(uint64_t) a.word * stride];
            if (a.kind == ACC_SUM128 || a.kind == ACC_MIN128 || a.kind == ACC_MAX128) v = (i128) (((u128) acc[(uint64_t) (a.word + 1) * stride] << 64) | lo);
            else v = (i128) (int64_t) lo;
            if (out.fn == LDB_AGG_AVG && ok) {
               // (sum * 10^k) sdiv count in i128 (DecimalOpScaledLowering, LowerToStd.cpp:631-651)
               v = d_sdiv128((i128) ((u128) v * (u128) d_pow10(out.avg_pow10)), (i128) cnt);
            }
            break;
         }
      }
      if (!ok) v = 0;
      switch (out.out_width) {
         case 4: ((int32_t*) out_values)[g] = (int32_t) v; break;
         case 8: ((int64_t*) out_values)[g] = (int64_t) v; break;
         default: {
            if (!out.wide && out.fn != LDB_AGG_AVG) v = (i128) (int64_t) v;
            ((uint64_t*) out_values)[2 * g] = (uint64_t) v;
            ((uint64_t*) out_values)[2 * g + 1] = (uint64_t) (v >> 64);
         }
      }
      if (out_valid) ((uint8_t*) out_valid)[g] = ok ? 1 : 0;
   }
}

// ---------------------------------------------------------------- tables
// global find-or-insert; returns slot or ~0 on overflow
__device__ __forceinline__ uint64_t d_global_slot(const DGroupBy& m, const DGroupBy* __restrict__ d, uint64_t h, uint64_t i) {
   if (m.keyless) return 0;
   const uint64_t mask = d->g_cap - 1;
   unsigned long long* gk = gptr_mut<unsigned long long>(d->g_keys);
   const uint64_t mine = (h & 0xFFFFFFFF00000000ull) | (uint64_t) ((uint32_t) i + 1u);
   uint64_t pos = (h ^ (h >> 29)) & mask;
   const KV keys(m.keys, d->keys);
   if (m.ordered_slots) {
      const CV c = keys.col(0);
      const uint32_t row = d_phys_row(c, i);
      if (d_valid(c, row)) pos = (((uint64_t) (d_load_i64(c, row) - d->kmin) * d->kmult) >> 32) & mask; // a NULL key keeps its hashed slot
   }
   for (uint64_t step = 0; step <= mask; step++) {
      if (m.ordered_slots && step == 512) { // skewed keys: give up at once (runs cost O(length^2)), the host retries hashed
         atomicOr(gptr_mut<uint32_t>(d->g_flags), 2u);
         return ~0ull;
      }
      // a long run means the table is (nearly) full — the estimate was too low, or there was none: without this exit every remaining
      // row walks the whole table before it reports the overflow (6 M rows x 4 M slots).  The first lane to see it raises the flag, the
      // others notice within 32 steps and stop probing; the host retries with 8 x the capacity.
      if (step >= 32 && (step & 31) == 0 && (step >= 4096 || (__hip_atomic_load(gptr_mut<uint32_t>(d->g_flags), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u))) {
         atomicOr(gptr_mut<uint32_t>(d->g_flags), 1u);
         return ~0ull;
      }
      unsigned long long w = __hip_atomic_load(&gk[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w == 0) {
         unsigned long long old = atomicCAS(&gk[pos], 0ull, (unsigned long long) mine);
         if (old == 0) return pos;
         w = old;
      }
      if ((w >> 32) == (h >> 32) && d_keys_equal(keys, (uint64_t) ((uint32_t) w - 1u), keys, i, true)) return pos;
      pos = (pos + 1) & mask;
   }
   atomicOr(gptr_mut<uint32_t>(d->g_flags), 1u);
   return ~0ull;
}

// direct-address mode (DGroupBy::direct): the slot IS the key
__device__ __forceinline__ uint64_t d_direct_slot(const DGroupBy& m, const DGroupBy* __restrict__ d, uint64_t i) {
   const KV keys(m.keys, d->keys);
   const CV c = keys.col(0);
   return (uint64_t) (d_load_i64(c, d_phys_row(c, i)) - d->kmin);
}

// find-or-insert the group of logical row i in replica `rep` of the workgroup's LDS table;
// returns the slot or -1 (table region full / long probe run → caller uses the global table)
__device__ __forceinline__ int32_t d_lds_slot(const DGroupBy& m, KV keys, unsigned long long* l_keys, uint32_t S, uint32_t R, uint32_t rep, uint64_t h, uint64_t i) {
   if (m.keyless) return 0;
   const unsigned long long mine = (h & 0xFFFFFFFF00000000ull) | (unsigned long long) ((uint32_t) i + 1u);
   uint32_t pos = (uint32_t) (h >> 6) & (S - 1);
   for (uint32_t step = 0; step < S && step < 16; step++) {
      unsigned long long w = l_keys[pos * R + rep];
      if (w == 0) {
         unsigned long long old = atomicCAS(&l_keys[pos * R + rep], 0ull, mine);
         if (old == 0) return (int32_t) pos;
         w = old;
      }
      if ((w >> 32) == (h >> 32) && d_keys_equal(keys, (uint64_t) ((uint32_t) w - 1u), keys, i, true)) return (int32_t) pos;
      pos = (pos + 1) & (S - 1);
   }
   return -1;
}

// the kernel body: `m` = metadata source (== *d in the generic kernel, a constexpr in a
// specialised one), `d` = this launch's descriptor in device memory (addresses, sizes)
template <int ROWS>
__device__ __forceinline__ void gb_body(const DGroupBy& m, const DGroupBy* __restrict__ d, unsigned long long* gb_lds) {
   const uint32_t S = d->lds_slots, R = d->lds_reps;
   const uint32_t SR = S * R;
   const int nw = m.n_words;
   const bool use_lds = m.use_lds != 0;
   unsigned long long* l_keys = gb_lds;
   unsigned long long* l_acc = gb_lds + SR; // word w of index idx at l_acc[w*SR + idx], idx = slot*R + replica
   if (use_lds) {
      for (uint32_t k = threadIdx.x; k < SR; k += GB_BLOCK) l_keys[k] = m.keyless ? 1ull : 0ull;
      LDB_UNROLL
      for (int w = 0; w < nw; w++) {
         unsigned long long init = m.word_init[w];
         for (uint32_t k = threadIdx.x; k < SR; k += GB_BLOCK) l_acc[(uint32_t) w * SR + k] = init;
      }
      __syncthreads();
   }
   const uint64_t n = d->n_rows;
   const uint32_t rep = threadIdx.x & (R - 1);
   const int np = m.n_preds;
   const KV keys(m.keys, d->keys);
   unsigned long long* g_acc = gptr_mut<unsigned long long>(d->g_acc);
   const uint64_t g_cap = d->g_cap;
   // ROWS rows per thread per iteration: phase A issues the predicate / key / value loads of all
   // rows (independent → memory-level parallelism, the dependent predicate chain of a selective
   // scan overlaps across rows), phase B folds each surviving row into its group.
   const uint64_t tid = blockIdx.x * (uint64_t) GB_BLOCK + threadIdx.x;
   const uint64_t nthreads = (uint64_t) gridDim.x * GB_BLOCK;
   const uint32_t lane = threadIdx.x & 63;
   // wave-uniform trip count (the wave's first row decides): the run-combining path shuffles
   for (uint64_t i0 = tid; i0 - lane < n; i0 += nthreads * ROWS) {
      bool passv[ROWS];
      uint64_t hv[ROWS];
      long long rvv[ROWS][GB_MAX_COLS];
      uint32_t rvalidv[ROWS];
      uint64_t rowsv[ROWS];
#pragma unroll
      for (int u = 0; u < ROWS; u++) {
         rowsv[u] = i0 + (uint64_t) u * nthreads;
         passv[u] = rowsv[u] < n;
      }
      // predicate-major: each conjunct is evaluated for all rows of the batch (loads first, then
      // compares), so the dependent filter chain costs one memory round trip per conjunct column
      // for the whole batch rather than one per row
#if GB_PRED_BATCH
      d_eval_conj_batch<ROWS>(m.preds, d->preds, np, rowsv, passv);
#else
#pragma unroll
      for (int u = 0; u < ROWS; u++) {
         bool pass = passv[u];
         LDB_UNROLL
         for (int p = 0; p < np; p++)
            if (pass) pass = d_eval_pred(PV(m.preds[p], d->preds[p]), rowsv[u]);
         passv[u] = pass;
      }
#endif
#pragma unroll
      for (int u = 0; u < ROWS; u++) {
         hv[u] = 0;
         rvalidv[u] = 0;
         if (passv[u]) {
            if (!m.keyless && !m.dense_sorted && !m.direct) hv[u] = d_hash_keys(keys, rowsv[u]); // (dense_sorted / direct: no hash)
            d_load_vals(m, d, rowsv[u], rvv[u], rvalidv[u]);
         }
      }
      if (!use_lds) {
         // high-cardinality path: runs of neighbouring lanes with the same key are combined in the
         // wave, the run's head lane does the one lookup and the atomics (see d_accumulate_runs)
         // every memory access the combining step depends on is issued here for the whole batch
         // (key of the previous row, the chunk's group base, the next chunk's first key), so the
         // loop below only shuffles, stores and issues atomics: without this each batch row paid
         // three dependent round trips of its own and the kernel ran at latency, not bandwidth
         bool eqprev[ROWS], eqnext[ROWS];
         uint32_t gbase[ROWS], gprev[ROWS];
         long long kcur[ROWS]; // dense_sorted: the row's key (also the value of the output key column, dense_keys)
         if (m.dense_sorted) {
            // one dense NOT NULL integer key (what the sorted statistic guarantees): branch-free
            // clamped loads of the previous / own / next key for the whole batch, compares after
            const CV kc = keys.col(0);
            long long kprev[ROWS], knext[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; u++) {
               const uint64_t ii = rowsv[u] < n ? rowsv[u] : n - 1;
               kcur[u] = d_load_i64(kc, (uint32_t) ii);
               // (the neighbours' keys are LOADED, not shuffled: taking the lane below's key by __shfl_up and loading only the chunk's outer
               // neighbours in lanes 0 / 63 made this kernel 28 % slower at SF100, 3.93 → 5.04 ms — the extra loads hit lines already in flight,
               // the 64-bit cross-lane moves and the two divergent single-lane loads sit on the critical path of every batch)
               kprev[u] = d_load_i64(kc, (uint32_t) (ii > 0 ? ii - 1 : 0));
               knext[u] = d_load_i64(kc, (uint32_t) (ii + 1 < n ? ii + 1 : ii));
               gbase[u] = gptr<uint32_t>(d->chunk_off)[ii >> 6];
               if (m.dense_out) gprev[u] = gptr<uint32_t>(d->chunk_off)[(ii >> 6) ? (ii >> 6) - 1 : 0];
            }
#pragma unroll
            for (int u = 0; u < ROWS; u++) {
               const uint64_t i = rowsv[u];
               eqprev[u] = passv[u] & (i > 0) & (kprev[u] == kcur[u]);
               eqnext[u] = (lane == 63) & passv[u] & (i + 1 < n) & (knext[u] == kcur[u]);
            }
         } else {
#pragma unroll
            for (int u = 0; u < ROWS; u++) {
               const uint64_t i = rowsv[u];
               eqprev[u] = passv[u] && i > 0 && (m.keyless || d_keys_equal(keys, i - 1, keys, i, true));
               eqnext[u] = false;
               gbase[u] = 0;
               kcur[u] = 0;
            }
         }
#pragma unroll
         for (int u = 0; u < ROWS; u++) {
            const uint64_t i = rowsv[u];
            const bool pass = passv[u];
            const uint64_t h = hv[u];
            const bool pprev = __shfl_up(pass ? 1 : 0, 1) != 0;
            const bool same = lane > 0 && pass && pprev && eqprev[u]; // lane - 1 holds row i - 1
            const bool head = pass && !same;
            const uint64_t headmask = __ballot(head), passmask = __ballot(pass);
            if (passmask == 0) continue; // wave-uniform
            const uint64_t above = lane >= 63 ? 0ull : (~0ull << (lane + 1));
            const uint64_t brk = (headmask | ~passmask) & above; // first lane after this one that is not a member of its run
            const uint32_t run_end = !pass ? lane : (brk ? (uint32_t) __builtin_ctzll(brk) - 1u : 63u);
            uint64_t g = ~0ull;
            bool plain = false;
            if (m.dense_sorted) {
               // sorted key column: group number = key changes up to this row (see DGroupBy::dense_sorted).
               // The wave's rows are one aligned 64-row chunk (i - lane is a multiple of 64).
               const bool true_head = pass && !eqprev[u]; // the key changes at this row (lane 0 included)
               const bool cont = __shfl(eqnext[u] ? 1 : 0, 63) != 0; // does the chunk's last run continue into the next chunk?
               const uint64_t thmask = __ballot(true_head);
               if (head) {
                  const uint64_t upto = lane >= 63 ? ~0ull : ((2ull << lane) - 1ull);
                  g = (uint64_t) gbase[u] + (uint64_t) __popcll(thmask & upto) - 1ull;
                  plain = true_head && !(run_end == 63 && cont); // the whole group lives inside this run
                  if (g >= d->dense_groups) g = ~0ull; // (only under a mis-speculated replay: the output arrays were sized from the recorded count)
                  if (m.dense_out) {
                     if (true_head && g != ~0ull) {
                        if (m.dense_keys != 2) gptr_mut<uint32_t>(d->rep_rows_out)[g] = (uint32_t) i;
                        if (m.dense_keys) { // the key column of the result, written where the key is in a register anyway
                           const long long key = kcur[u];
                           switch (m.direct_key_width) {
                              case 4: gptr_mut<int32_t>(d->direct_keys_out)[g] = (int32_t) key; break;
                              case 8: gptr_mut<int64_t>(d->direct_keys_out)[g] = (int64_t) key; break;
                              default:
                                 gptr_mut<int64_t>(d->direct_keys_out)[2 * g] = (int64_t) key;
                                 gptr_mut<int64_t>(d->direct_keys_out)[2 * g + 1] = key < 0 ? -1 : 0;
                           }
                        }
                     }
                  } else if (true_head && g != ~0ull) {
                     gptr_mut<unsigned long long>(d->g_keys)[g] = (h & 0xFFFFFFFF00000000ull) | (unsigned long long) ((uint32_t) i + 1u);
                  }
               }
               if (m.dense_out) {
                  // a group inside one run: reduced into the head lane's private words and written out as the group's final row.  A group
                  // that crosses a chunk boundary: atomics into the slot of the chunk that holds its first row — this chunk for a run that
                  // begins here, an earlier one for the run that arrives at lane 0 (normally the previous chunk; a group longer than a
                  // chunk finds its first chunk by bisection of the group numbers) — finished by k_gb_finalize_cross
                  const uint64_t chunk = (i - lane) >> 6;
                  unsigned long long loc[GB_MAX_WORDS];
                  LDB_UNROLL
                  for (int w = 0; w < nw; w++) loc[w] = m.word_init[w];
                  uint64_t slot = chunk;
                  if (head && !plain && !true_head && g != ~0ull) { // the run continuing from the previous chunk (lane 0)
                     slot = chunk - 1;
                     if (gprev[u] == gbase[u]) { // no group begins in the previous chunk either: the largest chunk c with chunk_off[c] <= g
                        const uint32_t* co = gptr<uint32_t>(d->chunk_off);
                        uint64_t lo_c = 0, hi_c = chunk - 1;
                        while (lo_c < hi_c) {
                           const uint64_t mid = (lo_c + hi_c + 1) >> 1;
                           if ((uint64_t) co[mid] <= g) lo_c = mid;
                           else hi_c = mid - 1;
                        }
                        slot = lo_c;
                     }
                  }
                  if (head && !plain && true_head && g != ~0ull) gptr_mut<uint8_t>(d->cross_flags)[chunk] = 1;
                  // every head lane reduces its run into its OWN words (plain stores into identity-initialised words: registers in the
                  // specialised kernel); then the words become the group's output row, or are merged into the crossing group's slot
                  const Sink ls{loc, 1, true};
                  d_accumulate_runs(m, d, rvv[u], rvalidv[u], i, pass, head, lane, run_end, ls);
                  if (head && g != ~0ull) {
                     if (plain) {
                        d_finalize_group(m, d, loc, 1, g, (uint32_t) i);
                     } else {
                        const Sink dst{g_acc + slot, g_cap};
                        d_combine(m, ls, dst);
                     }
                  }
                  continue;
               }
            } else if (head) {
               g = m.direct ? d_direct_slot(m, d, i) : d_global_slot(m, d, h, i);
            }
            const bool apply = head && g != ~0ull;
            Sink s{g_acc + (apply ? g : 0), g_cap, plain};
            d_accumulate_runs(m, d, rvv[u], rvalidv[u], i, pass, apply, lane, run_end, s);
         }
         continue;
      }
#pragma unroll
      for (int u = 0; u < ROWS; u++) {
      if (!passv[u]) continue;
      const uint64_t i = i0 + (uint64_t) u * nthreads;
      const uint64_t h = hv[u];
      const RowVals& rv = rvv[u];
      const uint32_t rvalid = rvalidv[u];
      int32_t lslot = -1;
      if (use_lds) lslot = d_lds_slot(m, keys, l_keys, S, R, rep, h, i);
      if (lslot >= 0) {
         Sink s{l_acc + (uint32_t) lslot * R + rep, SR};
         d_accumulate(m, d, rv, rvalid, i, s);
      } else {
         uint64_t g = d_global_slot(m, d, h, i);
         if (g != ~0ull) {
            Sink s{g_acc + g, g_cap};
            d_accumulate(m, d, rv, rvalid, i, s);
         }
      }
      }
   }
   if (!use_lds) return;
   __syncthreads();
   // Fold the replicas into replica 0 inside the workgroup first (LDS atomics): a global flush of
   // every replica would put R x #workgroups same-address atomics on each hot group's global
   // slot — for a key-less SUM (Q6) that serialisation alone cost more than the scan.
   if (R > 1) {
      for (uint32_t k = threadIdx.x; k < SR; k += GB_BLOCK) {
         const uint32_t r = k & (R - 1);
         if (r == 0) continue;
         unsigned long long w = l_keys[k];
         if (w == 0) continue;
         const uint64_t i = (uint64_t) ((uint32_t) w - 1u);
         const uint64_t h = m.keyless ? 0 : d_hash_keys(keys, i);
         int32_t t = d_lds_slot(m, keys, l_keys, S, R, 0, h, i);
         if (t < 0) continue; // replica 0 full: this entry goes to the global table below
         Sink src{l_acc + k, SR};
         Sink dst{l_acc + (uint32_t) t * R, SR};
         d_combine(m, src, dst);
         l_keys[k] = 0;
      }
      __syncthreads();
   }
   // flush: every remaining occupied (slot, replica) → global table
   for (uint32_t k = threadIdx.x; k < SR; k += GB_BLOCK) {
      unsigned long long w = l_keys[k];
      if (w == 0) continue;
      Sink src{l_acc + k, SR};
      uint64_t g = 0;
      if (!m.keyless) {
         uint64_t i = (uint64_t) ((uint32_t) w - 1u);
         uint64_t h = d_hash_keys(keys, i);
         g = d_global_slot(m, d, h, i);
         if (g == ~0ull) continue;
      }
      Sink dst{g_acc + g, g_cap};
      d_combine(m, src, dst);
   }
}

// dense_sorted pass 1: key changes per 64-row chunk (→ exclusive scan → DGroupBy::chunk_off)
__device__ __forceinline__ void gb_sorted_heads_body(const DGroupBy& m, const DGroupBy* __restrict__ d, uint32_t* __restrict__ chunk_cnt) {
   const uint64_t n = d->n_rows, n_chunks = (n + 63) / 64;
   const uint32_t lane = threadIdx.x & 63;
   const KV keys(m.keys, d->keys);
   const uint64_t wave = (blockIdx.x * (uint64_t) blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
   // four chunks per wave and iteration: their key loads are independent and issue back to back (one chunk per iteration left a single
   // pair of loads in flight per wave: 2.4 GB of keys at 2.9 TB/s)
   for (uint64_t c0 = wave * 4; c0 < n_chunks; c0 += n_waves * 4) {
      bool th[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const uint64_t i = (c0 + u) * 64 + lane;
         th[u] = i < n && (i == 0 || !d_keys_equal(keys, i - 1, keys, i, true));
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const uint64_t mask = __ballot(th[u]);
         if (lane == 0 && c0 + u < n_chunks) chunk_cnt[c0 + u] = (uint32_t) __popcll(mask);
      }
   }
}
