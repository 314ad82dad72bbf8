// ldb_gbhost.hip — fused scan + predicate + hash group-by aggregation: host side, AOT kernels,
// and the hook into the run-time specialiser.  Device code: ldb_gb_kernel.h.
// Replaces (reference): the generated pipeline body around PreAggregationHashtableFragment
// (LookupPreAggrHtFragment, src/compiler/Conversion/SubOpToControlFlow/SubOpToControlFlow.cpp:3065-3157;
// ReduceOpLowering :3719-3768; PreAggregationHashtableFragment::insert,
// src/runtime/PreAggregationHashtable.cpp:46-60), PreAggregationHashtable::merge (:76-158),
// Hashtable (src/runtime/Hashtable.cpp) and SimpleState for key-less aggregates
// (src/runtime/SimpleState.cpp:8-30).
//
// MI355X design (not a port of the pointer-chained CPU tables):
//   * the reference's per-worker 1024-slot pre-aggregation cache becomes a per-workgroup table in
//     LDS (160 KB/CU): open addressing, slot word = {hash tag : 32 | representative row+1 : 32}
//     claimed by one 64-bit LDS CAS; accumulators are 64-bit words updated with LDS atomics.
//   * low-cardinality group-bys (TPC-H Q1: 4 groups) would serialise 64 lanes on a handful of
//     LDS addresses, so the table is REPLICATED R times and lane l works on replica l % R.
//   * 128-bit decimal SUMs are two 64-bit words: atomic add on the low word returns the old value,
//     the carry (old + v < old) is added to the high word — exact because additions commute and
//     every wrap of the low word is observed by exactly one lane.
//   * at the end each workgroup flushes its LDS groups into one global open-addressing table
//     (global 64-bit CAS + atomics); rows whose group does not fit LDS go to the global table
//     directly.  A final kernel compacts occupied slots into dense result columns.
//   * keys are never copied into the table: a slot names a representative input row; equality
//     is checked against that row's key columns (any key type, incl. strings), and the output key
//     columns are a gather of the representative rows (late materialisation).
//   * like the reference (which JIT-compiles each pipeline with LLVM), large inputs run a kernel
//     specialised at run time on the descriptor (hiprtc, ldb_jit.hip); small inputs and any
//     specialisation failure use the generic ahead-of-time kernel below — both are this source.
#include "ldb_internal.h"
#include "ldb_chain.h"
#include "ldb_gb_kernel.h"
#include "ldb_jit.h"
#include <cmath>
#include <algorithm>
#include <memory>

extern __shared__ __attribute__((aligned(16))) unsigned long long gb_lds_dyn[];

// generic ahead-of-time kernel: metadata and addresses both come from the descriptor in memory
__global__ __launch_bounds__(GB_BLOCK) void k_groupby(const DGroupBy* __restrict__ d) { gb_body<1>(*d, d, gb_lds_dyn); }

__global__ void k_gb_sorted_heads(const DGroupBy* __restrict__ d, uint32_t* __restrict__ chunk_cnt) { gb_sorted_heads_body(*d, d, chunk_cnt); }

__global__ void k_gb_init(uint64_t* keys, uint64_t* acc, const DGroupBy* __restrict__ d) {
   const uint64_t cap = d->g_cap;
   const int nw = d->n_words;
   for (uint64_t p = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; p < cap; p += (uint64_t) gridDim.x * blockDim.x) {
      if (keys) keys[p] = (d->keyless && p == 0) ? 1ull : 0ull; // (dense_out: no slot words at all)
      for (int w = 0; w < nw; w++) acc[(uint64_t) w * cap + p] = d->word_init[w];
   }
}

// ---------------------------------------------------------------- partitioned counting (direct slots, COUNT(*) only)
// Q13's shape: 148 M rows → a row count per customer, 10 M random keys.  The direct path pays one global atomic per
// row into a 128 MB table (23 G atomics/s: 6.3 ms).  Here the rows' slots are first radix-partitioned by slot range —
// a histogram pass and a scatter pass with workgroup-private LDS cursors, 4 bytes per row each way — and then every
// partition (16 K consecutive slots) is counted by ONE workgroup in 64 KB of LDS and written to the table with plain,
// coalesced stores: no global atomics, ~2.5 GB of sequential traffic instead of 148 M random read-modify-writes.
// Semantics of the reference unchanged (PreAggregationHashtable.cpp:46-158 reduces thread-local fragments per
// partition; this is the same idea with LDS as the fragment).
#define GBP_BLOCK 256
#define GBP_SBLOCK 1024 // histogram / scatter: few, large workgroups → long per-(workgroup, partition) runs, full cache lines
#define GBP_SHIFT 14
#define GBP_MAX_PARTS 4096
__global__ __launch_bounds__(GBP_SBLOCK) void k_gbp_hist(const DGroupBy* __restrict__ d, uint32_t nparts, uint64_t rows_per_wg, uint32_t* __restrict__ hist) {
   __shared__ uint32_t h[GBP_MAX_PARTS];
   for (uint32_t p = threadIdx.x; p < nparts; p += GBP_SBLOCK) h[p] = 0;
   __syncthreads();
   const uint64_t n = d->n_rows, b = blockIdx.x * rows_per_wg, e = b + rows_per_wg < n ? b + rows_per_wg : n;
   for (uint64_t i = b + threadIdx.x; i < e; i += GBP_SBLOCK) atomicAdd(&h[(uint32_t) (d_direct_slot(*d, d, i) >> GBP_SHIFT)], 1u);
   __syncthreads();
   for (uint32_t p = threadIdx.x; p < nparts; p += GBP_SBLOCK) hist[(uint64_t) p * gridDim.x + blockIdx.x] = h[p];
}
__global__ __launch_bounds__(GBP_SBLOCK) void k_gbp_scatter(const DGroupBy* __restrict__ d, uint32_t nparts, uint64_t rows_per_wg, const uint32_t* __restrict__ offs, uint32_t* __restrict__ slots_out) {
   __shared__ uint32_t cur[GBP_MAX_PARTS];
   for (uint32_t p = threadIdx.x; p < nparts; p += GBP_SBLOCK) cur[p] = offs[(uint64_t) p * gridDim.x + blockIdx.x];
   __syncthreads();
   const uint64_t n = d->n_rows, b = blockIdx.x * rows_per_wg, e = b + rows_per_wg < n ? b + rows_per_wg : n;
   for (uint64_t i = b + threadIdx.x; i < e; i += GBP_SBLOCK) {
      const uint32_t slot = (uint32_t) d_direct_slot(*d, d, i);
      slots_out[atomicAdd(&cur[slot >> GBP_SHIFT], 1u)] = slot;
   }
}
// the rows' direct slots as a dense 32-bit array: the input of the write-combining partition (ldb_wc.hip)
__global__ void k_gbp_slots(const DGroupBy* __restrict__ d, uint32_t* __restrict__ slots) {
   const uint64_t n = d->n_rows;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) slots[i] = (uint32_t) d_direct_slot(*d, d, i);
}
__global__ __launch_bounds__(GBP_BLOCK) void k_gbp_count(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ offs, uint32_t grid0, uint32_t nparts, uint64_t n_total,
                                                         uint64_t* __restrict__ counters) {
   __shared__ uint32_t c[1u << GBP_SHIFT];
   const uint32_t p = blockIdx.x;
   for (uint32_t j = threadIdx.x; j < (1u << GBP_SHIFT); j += GBP_BLOCK) c[j] = 0;
   __syncthreads();
   const uint64_t b = offs[(uint64_t) p * grid0], e = p + 1 < nparts ? (uint64_t) offs[(uint64_t) (p + 1) * grid0] : n_total;
   for (uint64_t i = b + threadIdx.x; i < e; i += GBP_BLOCK) atomicAdd(&c[slots[i] & ((1u << GBP_SHIFT) - 1)], 1u);
   __syncthreads();
   uint64_t* out = counters + ((uint64_t) p << GBP_SHIFT);
   for (uint32_t j = threadIdx.x; j < (1u << GBP_SHIFT); j += GBP_BLOCK) out[j] = c[j];
}

// the general case of the partitioned path: ANY aggregates over direct slots (Q15: SUM of a 128-bit product per supplier over 23 M random
// keys — 13 G global atomics / s = 4.6 ms).  The (slot, row) pairs are partitioned by slot range like the counts above; ONE workgroup then owns
// 2^shift consecutive slots, keeps their accumulator words in LDS (word w of local slot j at lds[w << shift | j]), folds its rows in with LDS
// atomics — the row's columns are gathered by row number — and writes the finished words to the table with plain coalesced stores.  The
// reference's merge works the same way: one task per partition of the pre-aggregation fragments (PreAggregationHashtable.cpp:76-158).
extern __shared__ __attribute__((aligned(16))) unsigned long long gbp_lds[];
__global__ __launch_bounds__(GBP_BLOCK) void k_gbp_agg(const DGroupBy* __restrict__ d, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ offs,
                                                       uint32_t grid0, uint32_t nparts, uint64_t n_total, uint32_t shift) {
   const uint32_t P = 1u << shift;
   const int nw = d->n_words;
   for (int w = 0; w < nw; w++) {
      const unsigned long long init = d->word_init[w];
      for (uint32_t j = threadIdx.x; j < P; j += GBP_BLOCK) gbp_lds[((uint32_t) w << shift) + j] = init;
   }
   __syncthreads();
   const uint32_t p = blockIdx.x;
   const uint64_t b = offs[(uint64_t) p * grid0], e = p + 1 < nparts ? (uint64_t) offs[(uint64_t) (p + 1) * grid0] : n_total;
   for (uint64_t i = b + threadIdx.x; i < e; i += GBP_BLOCK) {
      const uint32_t slot = slots[i] & (P - 1);
      const uint64_t row = rows[i];
      RowVals rv;
      uint32_t rvalid;
      d_load_vals(*d, d, row, rv, rvalid);
      const Sink s{gbp_lds + slot, P};
      d_accumulate(*d, d, rv, rvalid, row, s);
   }
   __syncthreads();
   unsigned long long* acc = (unsigned long long*) d->g_acc;
   const uint64_t cap = d->g_cap;
   for (int w = 0; w < nw; w++)
      for (uint32_t j = threadIdx.x; j < P; j += GBP_BLOCK) acc[(uint64_t) w * cap + ((uint64_t) p << shift) + j] = gbp_lds[((uint32_t) w << shift) + j];
}

// compact occupied slots → dense outputs
// occupied slots per 64-slot chunk (→ exclusive scan → output position of every group).  A cursor
// atomic per wave instead is one contended address: ~10 ns each in the L2 — 84 ms for the 8.4 M
// chunks of Q18's 150 M-group table.
__global__ void k_gb_occupancy(const uint64_t* __restrict__ g_keys, uint64_t cap, uint32_t* __restrict__ pop) {
   for (uint64_t p = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; p < ((cap + 63) & ~63ull); p += (uint64_t) gridDim.x * blockDim.x) {
      const uint64_t occ = __ballot(p < cap && g_keys[p] != 0);
      if ((threadIdx.x & 63) == 0) pop[p >> 6] = (uint32_t) __popcll(occ);
   }
}
// the same + the exclusive scan of the chunk counts in ONE chained launch (ldb_chain.h): a tile is 256 chunks (16 384 slots), a
// wave takes 64 of them with independent loads; off[c] = occupied slots before chunk c, *total = number of groups
#define GBO_TILE_CHUNKS 256
__global__ __launch_bounds__(256) void k_gb_occupancy_scan(const uint64_t* __restrict__ occ_words, uint64_t cap, uint64_t n_chunks, uint64_t n_tiles, uint32_t* __restrict__ off,
                                                           unsigned long long* __restrict__ total, unsigned long long* __restrict__ status, unsigned long long* __restrict__ ticket,
                                                           unsigned long long ticket_base, unsigned long long epoch) {
   __shared__ unsigned long long s_tile;
   __shared__ uint32_t s_wave[4];
   __shared__ uint32_t s_prefix;
   __shared__ uint32_t s_cnt[GBO_TILE_CHUNKS];
   if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1ull) - ticket_base;
   __syncthreads();
   const uint64_t tile = s_tile;
   const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const uint64_t chunk0 = tile * GBO_TILE_CHUNKS + (uint64_t) wave * 64;
#pragma unroll 8
   for (uint32_t k = 0; k < 64; k++) {
      const uint64_t p = (chunk0 + k) * 64 + lane;
      const uint64_t m = __ballot(p < cap && occ_words[p] != 0);
      if (lane == 0) s_cnt[wave * 64 + k] = (uint32_t) __popcll(m);
   }
   __syncthreads();
   const uint32_t own = s_cnt[threadIdx.x];
   uint32_t agg;
   uint32_t excl = d_block_scan256<uint32_t>(own, s_wave, &agg);
   if (threadIdx.x < 64) {
      const uint32_t prefix = d_chain_prefix<uint32_t, 32>(status, tile, epoch, agg, threadIdx.x);
      if (threadIdx.x == 0) s_prefix = prefix;
   }
   __syncthreads();
   const uint64_t c = tile * GBO_TILE_CHUNKS + threadIdx.x;
   if (c < n_chunks) off[c] = s_prefix + excl;
   if (total && tile == n_tiles - 1 && threadIdx.x == 0) *total = (unsigned long long) s_prefix + agg;
}
__global__ void k_gb_finalize(const DGroupBy* __restrict__ d, uint32_t* __restrict__ rep_rows, const uint32_t* __restrict__ chunk_off) {
   const uint64_t cap = d->g_cap;
   // occupancy: the slot word, or in direct-address mode the group's row counter
   const uint64_t* occ = d->direct ? (const uint64_t*) d->g_acc + (uint64_t) d->direct_word * cap : (const uint64_t*) d->g_keys;
   for (uint64_t p = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; p < ((cap + 63) & ~63ull); p += (uint64_t) gridDim.x * blockDim.x) {
      // the grid stride is a multiple of the wave size: a wave covers one aligned 64-slot chunk
      uint64_t w = p < cap ? occ[p] : 0;
      const uint64_t occ = __ballot(w != 0);
      if (w == 0) continue;
      const uint32_t lane = threadIdx.x & 63;
      uint64_t g = (uint64_t) chunk_off[p >> 6] + (uint64_t) __popcll(occ & ((1ull << lane) - 1ull));
      uint32_t rep = (uint32_t) w - 1u; // (meaningless in direct mode: no aggregate reads the representative row there)
      if (d->direct) { // the key IS the slot number
         const long long key = (long long) d->kmin + (long long) p;
         switch (d->direct_key_width) {
            case 4: ((int32_t*) d->direct_keys_out)[g] = (int32_t) key; break;
            case 8: ((int64_t*) d->direct_keys_out)[g] = (int64_t) key; break;
            default:
               ((int64_t*) d->direct_keys_out)[2 * g] = (int64_t) key;
               ((int64_t*) d->direct_keys_out)[2 * g + 1] = key < 0 ? -1 : 0;
         }
      } else {
         rep_rows[g] = rep;
      }
      d_finalize_group(*d, d, (const unsigned long long*) d->g_acc + p, cap, g, rep);
   }
}
// dense_out: the groups that cross a chunk boundary.  One lane per chunk: a flagged chunk holds the first row of a group that continues
// into the next chunk; its accumulators are the chunk's slot (word w at g_acc[w * n_chunks + chunk]), its number is the last group that
// begins in the chunk, its representative row was written by its first row.
__global__ void k_gb_finalize_cross(const DGroupBy* __restrict__ d, uint64_t n_chunks, const unsigned long long* __restrict__ d_groups) {
   const uint8_t* flags = (const uint8_t*) d->cross_flags;
   const uint32_t* chunk_off = (const uint32_t*) d->chunk_off;
   for (uint64_t c = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; c + 1 < n_chunks; c += (uint64_t) gridDim.x * blockDim.x) {
      if (!flags[c]) continue;
      const uint64_t g = (uint64_t) chunk_off[c + 1] - 1ull;
      if (g >= d->dense_groups) continue; // (mis-speculated replay)
      d_finalize_group(*d, d, (const unsigned long long*) d->g_acc + c, n_chunks, g, d->dense_keys == 2 ? 0u : ((const uint32_t*) d->rep_rows_out)[g]);
   }
   // the output arrays were sized from a count that may have been REPLAYED: should the real one be smaller (the execution is void and will be
   // repeated, but its consumers are queued already), the representative rows behind it must still be row numbers and the keys behind it keys
   // of the column's range (a table built over them indexes by key - min): row 0 / the first group's key.  Normally an empty range.
   for (uint64_t g = (uint64_t) *d_groups + blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; g < d->dense_groups; g += (uint64_t) gridDim.x * blockDim.x) {
      if (d->dense_keys != 2) ((uint32_t*) d->rep_rows_out)[g] = 0;
      if (d->dense_keys) {
         switch (d->direct_key_width) {
            case 4: ((int32_t*) d->direct_keys_out)[g] = ((const int32_t*) d->direct_keys_out)[0]; break;
            case 8: ((int64_t*) d->direct_keys_out)[g] = ((const int64_t*) d->direct_keys_out)[0]; break;
            default:
               ((int64_t*) d->direct_keys_out)[2 * g] = ((const int64_t*) d->direct_keys_out)[0];
               ((int64_t*) d->direct_keys_out)[2 * g + 1] = ((const int64_t*) d->direct_keys_out)[1];
         }
      }
   }
}

// per-group validity bytes → Arrow bitmap; *nulls += number of zero bytes (one atomic per wave of a
// bounded grid)
// (the row count is read on the device: the launch is queued before the host knows the number of groups; `cap` = the rows `bytes` and `bitmap` were
// allocated for — under a replay that is the RECORDED group count, and a mis-speculated execution may really have more groups: it is repeated
// after the end-of-trace comparison, but until then nothing may be read or written past the allocations)
__global__ void k_pack_valid_bytes(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ bitmap, const unsigned long long* __restrict__ d_n, uint64_t cap, unsigned long long* __restrict__ nulls) {
   const uint64_t n = min((uint64_t) *d_n, cap);
   uint64_t nb = (n + 7) / 8;
   unsigned int zeros = 0;
   for (uint64_t b = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; b < nb; b += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++)
         if (b * 8 + k < n) {
            if (bytes[b * 8 + k])
               m |= (uint8_t) (1u << k);
            else
               zeros++;
         }
      bitmap[b] = m;
   }
   for (int off = 32; off > 0; off >>= 1) zeros += __shfl_down(zeros, off);
   if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(nulls, (unsigned long long) zeros);
}

// ---------------------------------------------------------------- host side
int32_t ldb_rel_select(ldb_ctx* ctx, ldb_rel* in, uint32_t* sel, int64_t n_sel, ldb_rel** out);
int32_t ldb_gather_columns(ldb_ctx* ctx, const ldb_rel* r, const ldb_colref* refs, int32_t n_cols, ldb_column* outs);

static uint64_t next_pow2_u64(uint64_t v) {
   uint64_t p = 1;
   while (p < v) p <<= 1;
   return p;
}

struct GbBuilder {
   ldb_rel* in;
   DGroupBy* h;
   std::vector<ldb_colref> col_refs;
   int32_t add_col(ldb_colref ref, int32_t* idx) {
      for (size_t k = 0; k < col_refs.size(); k++)
         if (col_refs[k].side == ref.side && col_refs[k].col == ref.col) {
            *idx = (int32_t) k;
            return LDB_OK;
         }
      if (col_refs.size() >= GB_MAX_COLS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: more than %d distinct aggregate input columns", GB_MAX_COLS);
      DCol dc;
      LDB_TRY(ldb_make_dcol(in, ref, &dc));
      if (dc.type == LDB_T_UTF8) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: string column in an aggregate expression");
      h->cols[col_refs.size()] = dc;
      *idx = (int32_t) col_refs.size();
      col_refs.push_back(ref);
      h->n_cols = (int32_t) col_refs.size();
      return LDB_OK;
   }
   int32_t conv_expr(const ldb_expr* e, DExprG* out, bool* nullable) {
      memset(out, 0, sizeof(*out));
      if (e->n_terms < 0 || e->n_terms > LDB_MAX_TERMS) LDB_FAIL(LDB_ERR_INVALID, "expression: %d terms (max %d)", e->n_terms, LDB_MAX_TERMS);
      out->n_terms = e->n_terms;
      out->is_float = e->is_float;
      for (int t = 0; t < e->n_terms; t++) {
         const ldb_term& tm = e->t[t];
         if (tm.n_factors < 0 || tm.n_factors > LDB_MAX_FACTORS) LDB_FAIL(LDB_ERR_INVALID, "expression: %d factors (max %d)", tm.n_factors, LDB_MAX_FACTORS);
         if (tm.div_pow10 < 0 || tm.div_pow10 > 38) LDB_FAIL(LDB_ERR_INVALID, "expression: div_pow10 %d", tm.div_pow10);
         out->t[t].n_factors = tm.n_factors;
         out->t[t].negate = tm.negate;
         out->t[t].div_pow10 = tm.div_pow10;
         // fits64: a bound on |Π (a + b * col)| from the bytes each column is stored in and its decimal precision (no statistics needed)
         long double bound = 1.0L;
         bool provable = !e->is_float && tm.n_factors > 0 && ldb_option("gb_fits64", 1) != 0;
         for (int f = 0; f < tm.n_factors; f++) {
            out->t[t].f[f].has_col = tm.f[f].has_col;
            out->t[t].f[f].a = tm.f[f].a;
            out->t[t].f[f].b = tm.f[f].b;
            if (tm.f[f].has_col) {
               int32_t idx;
               LDB_TRY(add_col(tm.f[f].col, &idx));
               out->t[t].f[f].col_idx = idx;
               const ldb_column& c = in->sides[(size_t) tm.f[f].col.side].table->cols[(size_t) tm.f[f].col.col];
               const ldb_rel_side& sd = in->sides[(size_t) tm.f[f].col.side];
               // NULL-able: the column has a validity bitmap, or its side can carry outer-join padding (LDB_NULL_ROW row ids)
               *nullable = *nullable || c.validity != nullptr || (sd.rowids && sd.may_null);
               bool colf = c.type.type == LDB_T_FLOAT64 || c.type.type == LDB_T_FLOAT32;
               if (colf && !e->is_float) LDB_FAIL(LDB_ERR_INVALID, "expression: float column in an integer expression (set is_float)");
               long double cb = c.width >= 8 ? 9.3e18L : (long double) (1ull << (8 * c.width - 1)); // |v| <= 2^(8w-1)
               if (c.type.type == LDB_T_DECIMAL128) {
                  long double pb = 1.0L;
                  for (int k = 0; k < c.type.precision; k++) pb *= 10.0L;
                  cb = std::min(cb, pb);
                  if (c.type.precision >= 19) provable = false; // (a wide column is read as 128 bits)
               }
               if (colf) provable = false;
               bound *= std::max(1.0L, fabsl((long double) tm.f[f].a) + fabsl((long double) tm.f[f].b) * cb);
            } else {
               bound *= std::max(1.0L, fabsl((long double) tm.f[f].a));
            }
         }
         out->t[t].fits64 = provable && bound < 4.0e18L ? 1 : 0; // < 2^62; every factor's bound is >= 1, so every partial product is bounded by it too
      }
      return LDB_OK;
   }
};

static bool same_acc(const DAcc& a, const DAcc& b) {
   if (a.kind != b.kind || a.n_cpreds != b.n_cpreds || a.count_rows != b.count_rows) return false;
   for (int p = 0; p < a.n_cpreds; p++)
      if (a.cpred[p] != b.cpred[p]) return false;
   if (a.kind == ACC_COUNT && a.count_rows) return true;
   return memcmp(&a.e, &b.e, sizeof(DExprG)) == 0;
}

extern "C" int32_t ldb_gpu_groupby(ldb_ctx* ctx, ldb_rel* in, const ldb_filter_desc* preds, int32_t n_preds, const ldb_colref* keys, int32_t n_keys,
                                   const ldb_agg_spec* aggs, int32_t n_aggs, int64_t est_groups, ldb_table** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "groupby: NULL argument");
   if (n_preds < 0 || n_preds > LDB_MAX_PREDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: %d predicates (max %d)", n_preds, LDB_MAX_PREDS);
   if (in->pending.size() + (size_t) n_preds > LDB_MAX_PREDS) LDB_TRY(ldb_rel_force(ctx, in)); // else: a lazy input's conjuncts are fused below
   if (n_keys == 0 && !in->pending.empty()) // key-less ANY needs a representative row that passed the filter: materialise a lazy input
      for (int32_t a = 0; a < n_aggs; a++)
         if (aggs[a].fn == LDB_AGG_ANY) {
            LDB_TRY(ldb_rel_force(ctx, in));
            break;
         }
   // one integer key over a range too large for the caches, many rows, a lazy filter in front: the filter is evaluated FIRST — the partitioned
   // aggregation below (k_gbp_agg) wants the surviving rows as a dense list, and a selective filter (Q15: 23 M of 600 M rows) then decides
   // whether it is worth it at all.  The fused alternative pays one global atomic per accumulator and surviving row.
   if (n_keys == 1 && n_preds == 0 && !in->pending.empty() && in->n_rows >= ldb_option("gb_partition_min_rows", 8ll << 20) && ldb_option("gb_partition", 1) != 0 &&
       ldb_option("gb_partition_values", 1) != 0 && ldb_option("gb_direct", 1) != 0 && est_groups >= (1 << 19)) {
      const ldb_rel_side& ks = in->sides[(size_t) keys[0].side];
      const ldb_column& kc = ks.table->cols[(size_t) keys[0].col];
      int64_t lo = 0, hi = -1;
      if (!kc.validity && !ks.rowids && (kc.width == 4 || kc.width == 8) && ldb_column_range(ctx, ks.table, keys[0].col, &lo, &hi) == LDB_OK && hi >= lo) {
         const unsigned __int128 range = (unsigned __int128) ((__int128) hi - lo) + 1;
         if (range >= ((unsigned __int128) 1 << 19) && range <= (unsigned __int128) est_groups * 2 && range <= ((unsigned __int128) 1 << 30)) LDB_TRY(ldb_rel_force(ctx, in));
      }
   }
   if (n_aggs < 0 || n_aggs > GB_MAX_OUT) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: %d aggregates (max %d)", n_aggs, GB_MAX_OUT);
   if (in->n_rows >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: too many rows");
   auto hp = std::make_unique<DGroupBy>();
   DGroupBy* h = hp.get();
   memset(h, 0, sizeof(*h));
   h->n_rows = (uint64_t) in->n_rows;
   h->n_preds = n_preds;
   for (int32_t p = 0; p < n_preds; p++) LDB_TRY(ldb_make_dpred(in, &preds[p], &h->preds[p]));
   for (auto& dp : in->pending) h->preds[h->n_preds++] = dp;
   n_preds = h->n_preds;
   ldb_order_preds(h->preds, n_preds); // cheap conjuncts first, same-column neighbours marked
   h->batch_rows = n_preds >= 2 ? 8 : 4;
   LDB_TRY(ldb_make_dkeys_dict(in, keys, n_keys, &h->keys)); // (dictionary codes stand in for low-cardinality strings: hashing and equality only)
   h->keyless = n_keys == 0;
   GbBuilder b{in, h, {}};

   auto add_acc = [&](DAcc& a, int words, uint64_t init, int32_t* idx) -> int32_t {
      for (int k = 0; k < h->n_accs; k++)
         if (same_acc(h->accs[k], a)) {
            *idx = k;
            return LDB_OK;
         }
      if (h->n_accs >= GB_MAX_ACCS || h->n_words + words > GB_MAX_WORDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: too many distinct accumulators");
      a.word = h->n_words;
      h->word_init[h->n_words] = init;
      for (int k = 1; k < words; k++) h->word_init[h->n_words + k] = 0;
      h->n_words += words;
      h->accs[h->n_accs] = a;
      *idx = h->n_accs++;
      return LDB_OK;
   };

   // output table layout: keys then aggregates
   auto res = std::make_unique<ldb_table>();
   res->ctx = ctx;
   res->name = "groupby";
   h->n_outs = n_aggs;
   struct OutInfo {
      ldb_coltype type;
      int width;
   };
   std::vector<OutInfo> oinfo((size_t) n_aggs);
   for (int32_t a = 0; a < n_aggs; a++) {
      const ldb_agg_spec& sp = aggs[a];
      DOut& o = h->outs[a];
      memset(&o, 0, sizeof(o));
      o.fn = sp.fn;
      o.wide = sp.wide;
      o.avg_pow10 = sp.avg_pow10;
      o.acc = -1;
      o.cnt_acc = -1;
      o.cnt_rows_acc = -1;
      o.cnt_pass_acc = -1;
      o.is_float = sp.arg.is_float && sp.fn != LDB_AGG_COUNT && sp.fn != LDB_AGG_COUNT_STAR;
      if (sp.n_preds < 0 || sp.n_preds > LDB_MAX_AGG_PREDS) LDB_FAIL(LDB_ERR_INVALID, "groupby: aggregate %d has %d predicates", a, sp.n_preds);
      if (sp.avg_pow10 < 0 || sp.avg_pow10 > 38) LDB_FAIL(LDB_ERR_INVALID, "groupby: avg_pow10 %d", sp.avg_pow10);
      DAcc acc;
      memset(&acc, 0, sizeof(acc));
      acc.n_cpreds = sp.n_preds;
      for (int p = 0; p < sp.n_preds; p++) {
         DPred dp;
         LDB_TRY(ldb_make_dpred(in, &sp.preds[p], &dp));
         int found = -1;
         for (int k = 0; k < h->n_cpreds; k++)
            if (!memcmp(&h->cpreds[k], &dp, sizeof(DPred))) found = k;
         if (found < 0) {
            if (h->n_cpreds >= GB_MAX_CPREDS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: too many conditional-aggregate predicates");
            h->cpreds[h->n_cpreds] = dp;
            found = h->n_cpreds++;
         }
         acc.cpred[p] = found;
      }
      bool nullable = false;
      if (sp.fn != LDB_AGG_COUNT_STAR) LDB_TRY(b.conv_expr(&sp.arg, &acc.e, &nullable));
      o.e = acc.e;
      // counter of contributing rows: AVG divisor, and NULL-ness of SUM/MIN/MAX results
      auto need_counter = [&](int32_t* idx) -> int32_t {
         DAcc c = acc;
         if (sp.fn == LDB_AGG_AVG && sp.has_count_expr) {
            // merging partial (sum, count) states: the divisor is SUM(count_expr)
            bool cn = false;
            c.kind = ACC_SUM64;
            c.count_rows = 0;
            LDB_TRY(b.conv_expr(&sp.count_expr, &c.e, &cn));
            if (c.e.is_float) LDB_FAIL(LDB_ERR_INVALID, "groupby: AVG count_expr must be an integer expression");
            return add_acc(c, 1, 0, idx);
         }
         c.kind = ACC_COUNT;
         c.count_rows = (!nullable && sp.fn != LDB_AGG_COUNT) ? 1 : 0;
         if (c.count_rows) memset(&c.e, 0, sizeof(c.e));
         return add_acc(c, 1, 0, idx);
      };
      switch (sp.fn) {
         case LDB_AGG_COUNT_STAR: {
            acc.kind = ACC_COUNT;
            acc.count_rows = 1;
            LDB_TRY(add_acc(acc, 1, 0, &o.acc));
            break;
         }
         case LDB_AGG_COUNT: {
            acc.kind = ACC_COUNT;
            acc.count_rows = nullable ? 0 : 1;
            if (acc.count_rows) memset(&acc.e, 0, sizeof(acc.e));
            LDB_TRY(add_acc(acc, 1, 0, &o.acc));
            break;
         }
         case LDB_AGG_SUM:
         case LDB_AGG_AVG: {
            if (sp.arg.is_float) {
               acc.kind = ACC_SUMF64;
               LDB_TRY(add_acc(acc, 1, 0, &o.acc));
            } else if (sp.wide) {
               acc.kind = ACC_SUM128;
               LDB_TRY(add_acc(acc, 2, 0, &o.acc));
            } else {
               acc.kind = ACC_SUM64;
               LDB_TRY(add_acc(acc, 1, 0, &o.acc));
            }
            // a plain SUM over no non-NULL input is NULL; a conditional SUM (case … else 0) gets a
            // non-NULL 0 from every row failing its predicates (see DOut::cnt_rows_acc)
            if (sp.fn == LDB_AGG_AVG || nullable || (h->keyless && sp.n_preds == 0)) LDB_TRY(need_counter(&o.cnt_acc));
            if (sp.fn == LDB_AGG_SUM && sp.n_preds > 0 && (nullable || h->keyless)) {
               DAcc rows;
               memset(&rows, 0, sizeof(rows));
               rows.kind = ACC_COUNT;
               rows.count_rows = 1;
               LDB_TRY(add_acc(rows, 1, 0, &o.cnt_rows_acc));
               if (nullable) {
                  DAcc passing = acc; // same predicates, counts rows
                  passing.kind = ACC_COUNT;
                  passing.count_rows = 1;
                  memset(&passing.e, 0, sizeof(passing.e));
                  LDB_TRY(add_acc(passing, 1, 0, &o.cnt_pass_acc));
               }
            }
            break;
         }
         case LDB_AGG_MIN:
         case LDB_AGG_MAX: {
            if (sp.wide && !sp.arg.is_float) { // low word, signed high word, lock word (d_sink_minmax128)
               acc.kind = sp.fn == LDB_AGG_MIN ? ACC_MIN128 : ACC_MAX128;
               LDB_TRY(add_acc(acc, 3, 0, &o.acc));
               const int w0 = h->accs[o.acc].word;
               h->word_init[w0] = sp.fn == LDB_AGG_MIN ? ~0ull : 0ull;
               h->word_init[w0 + 1] = sp.fn == LDB_AGG_MIN ? (uint64_t) INT64_MAX : (uint64_t) INT64_MIN;
               h->word_init[w0 + 2] = 0;
            } else if (sp.arg.is_float) {
               double init = sp.fn == LDB_AGG_MIN ? __builtin_inf() : -__builtin_inf();
               uint64_t bits;
               memcpy(&bits, &init, 8);
               acc.kind = sp.fn == LDB_AGG_MIN ? ACC_MINF64 : ACC_MAXF64;
               LDB_TRY(add_acc(acc, 1, bits, &o.acc));
            } else {
               acc.kind = sp.fn == LDB_AGG_MIN ? ACC_MIN64 : ACC_MAX64;
               LDB_TRY(add_acc(acc, 1, sp.fn == LDB_AGG_MIN ? (uint64_t) INT64_MAX : (uint64_t) INT64_MIN, &o.acc));
            }
            if (nullable || h->keyless || sp.n_preds) LDB_TRY(need_counter(&o.cnt_acc));
            break;
         }
         case LDB_AGG_ANY: // evaluated on the group's representative row
            // the key-less group is pre-seeded with row 0 as its representative, which need not pass a fused filter
            if (h->keyless && h->n_preds > 0) LDB_FAIL(LDB_ERR_UNSUPPORTED, "groupby: ANY in a key-less aggregation with fused predicates (filter the relation first)");
            break;
         default: LDB_FAIL(LDB_ERR_INVALID, "groupby: unknown aggregate function %d", sp.fn);
      }
      ldb_coltype ot{};
      ot.type = sp.out_type;
      ot.precision = sp.out_precision;
      ot.scale = sp.out_scale;
      ot.nullable = 1;
      if (o.is_float) ot.type = LDB_T_FLOAT64;
      if (sp.fn == LDB_AGG_COUNT || sp.fn == LDB_AGG_COUNT_STAR) ot.type = LDB_T_INT64;
      int w = ldb_width_of(ot, 0);
      if (w != 4 && w != 8 && w != 16) LDB_FAIL(LDB_ERR_INVALID, "groupby: aggregate %d output type %d unsupported", a, ot.type);
      o.out_width = w;
      oinfo[(size_t) a] = {ot, w};
   }
   // at least one word so that every slot has something to initialise
   if (h->n_words == 0) {
      DAcc c;
      memset(&c, 0, sizeof(c));
      c.kind = ACC_COUNT;
      c.count_rows = 1;
      int32_t idx;
      LDB_TRY(add_acc(c, 1, 0, &idx));
   }

   // ---- geometry
   int nw = h->n_words;
   const size_t slot_bytes = 8 * (size_t) (1 + nw);
   const size_t lds_budget = 60 * 1024;
   uint64_t est = est_groups > 0 ? (uint64_t) est_groups : 0;
   if (h->keyless) est = 1;
   uint32_t S, R;
   h->use_lds = 1;
   if (h->keyless) {
      S = 1;
      R = 64;
   } else if (est == 0) {
      S = 1;
      while ((size_t) S * 2 * slot_bytes <= lds_budget && S < 4096) S <<= 1;
      R = 1;
   } else {
      S = (uint32_t) std::max<uint64_t>(8, next_pow2_u64(est * 2));
      if ((size_t) S * slot_bytes > lds_budget) {
         // more groups than LDS can pre-aggregate: the hit rate of a small cache is poor when
         // the input is not clustered, so go straight to the global table
         uint32_t smax = 1;
         while ((size_t) smax * 2 * slot_bytes <= lds_budget) smax <<= 1;
         S = smax;
         if (est > (uint64_t) smax * 4) h->use_lds = 0;
         R = 1;
      } else {
         R = 1;
         while (R < 64 && (size_t) S * (R * 2) * slot_bytes <= lds_budget) R <<= 1;
      }
   }
   h->lds_slots = S;
   h->lds_reps = R;
   size_t lds_bytes = h->use_lds ? (size_t) S * R * slot_bytes : 0;

   uint64_t cap_guess = est ? est * 2 : std::min<uint64_t>((uint64_t) in->n_rows * 2, 1ull << 22);
   uint64_t cap = std::max<uint64_t>(1024, next_pow2_u64(cap_guess));
   const uint64_t cap_max = next_pow2_u64(std::max<uint64_t>(1024, (uint64_t) in->n_rows * 2));
   if (cap > cap_max) cap = cap_max;

   // ordered global slots (DGroupBy::ordered_slots): one integer key with a 32-bit value range
   unsigned __int128 key_range = 0;
   const bool gb_ordered = ldb_option("gb_ordered", 1) != 0;
   if (gb_ordered && !h->use_lds && n_keys == 1 && in->n_rows > 0) {
      int64_t lo = 0, hi = -1;
      const ldb_rel_side& ks = in->sides[(size_t) keys[0].side];
      if (!ks.table->cols[(size_t) keys[0].col].skewed && ldb_column_range(ctx, ks.table, keys[0].col, &lo, &hi) == LDB_OK && hi >= lo &&
          (unsigned __int128) ((__int128) hi - lo) < ((unsigned __int128) 1 << 32)) {
         h->ordered_slots = 1;
         h->kmin = lo;
         key_range = (unsigned __int128) ((__int128) hi - lo) + 1;
      }
   }
   // sorted key column, no filter (DGroupBy::dense_sorted): number the groups by key changes
   void* direct_keys = nullptr; // the output key column where the kernel writes it itself (direct slots; sorted keys with dense_keys)
   const ldb_column* direct_src = nullptr;
   uint32_t* chunk_off = nullptr;
   uint64_t sorted_groups = 0, sorted_chunks = 0; // dense_sorted: number of groups / of 64-row chunks
   uint64_t* d_sorted_groups = nullptr; // the same count on the device (an arena word)
   const bool gb_sorted = ldb_option("gb_sorted", 1) != 0;
   if (gb_sorted && !h->use_lds && n_keys == 1 && h->n_preds == 0 && in->n_rows > 0) {
      const ldb_rel_side& ks = in->sides[(size_t) keys[0].side];
      bool sorted = false;
      if (!ks.rowids) LDB_TRY(ldb_column_sorted(ctx, ks.table, keys[0].col, &sorted));
      if (sorted) {
         const int64_t n_chunks = (in->n_rows + 63) / 64;
         uint32_t* chunk_cnt;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &chunk_cnt, 4 * (size_t) n_chunks));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &chunk_off, 4 * (size_t) n_chunks));
         LdbDesc<DGroupBy> dh_desc(ctx);
         LDB_TRY(dh_desc.upload(h, sizeof(*h)));
         DGroupBy* dh = dh_desc.p;
         {
            const int hgrid = ldb_grid_for(ctx, in->n_rows, 256, 8);
            hipFunction_t spec = nullptr;
            std::string why;
            if (ldb_jit_wanted(in->n_rows)) spec = ldb_jit_groupby_kernel(ctx->device, h, "k_gb_sorted_heads_spec", &why);
            LdbProf prof_(ctx, "k_gb_sorted_heads");
            if (spec) {
               void* params[] = {(void*) &dh, (void*) &chunk_cnt};
               LDB_HIP(hipModuleLaunchKernel(spec, (unsigned) hgrid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
            } else {
               hipLaunchKernelGGL(k_gb_sorted_heads, dim3(hgrid), dim3(256), 0, ctx->stream, dh, chunk_cnt);
            }
         }
         uint64_t* d_groups;
         LDB_TRY(ldb_counters(ctx, 1, &d_groups));
         LDB_TRY(ldb_exclusive_scan_u32(ctx, chunk_cnt, chunk_off, n_chunks, d_groups));
         uint64_t groups = 0;
         LDB_TRY(ldb_read_u64(ctx, d_groups, &groups));
         dh_desc.release();
         ldb_dev_free(ctx, chunk_cnt);
         h->dense_sorted = 1;
         h->ordered_slots = 0;
         h->chunk_off = (uint64_t) chunk_off;
         cap = std::max<uint64_t>(1, groups); // the dense group array: exactly one slot per group
         sorted_groups = groups;
         sorted_chunks = (uint64_t) n_chunks;
         d_sorted_groups = d_groups;
         // groups inside one wave leave the kernel as final rows; the table shrinks to one slot per CHUNK for the groups that cross a
         // chunk boundary (DGroupBy::dense_out)
         h->dense_out = ldb_option("gb_dense_out", 1) != 0 ? 1 : 0;
         if (h->dense_out) cap = std::max<uint64_t>(1, sorted_chunks);
         // … and the key column itself: the first row of a group has its key in a register (no representative rows + gather afterwards: Q18's
         // 150 M keys, k_gather 0.72 ms); representative rows are still written when an ANY aggregate reads them
         const ldb_column& skc = ks.table->cols[(size_t) keys[0].col];
         if (h->dense_out && ldb_option("gb_dense_keys", 1) != 0 && !skc.validity && (skc.width == 4 || skc.width == 8 || skc.width == 16) && skc.type.type != LDB_T_UTF8 &&
             h->keys.cols[0].width == skc.width) {
            bool any_fn = false;
            for (int32_t a = 0; a < n_aggs; a++) any_fn = any_fn || aggs[a].fn == LDB_AGG_ANY;
            h->dense_keys = any_fn ? 1 : 2;
            h->direct_key_width = skc.width;
            direct_src = &skc;
         }
      }
   }
   // direct-address slots (DGroupBy::direct): one NOT NULL integer key whose value range is at most twice
   // the expected number of groups — the table is indexed by key - kmin
   if (ldb_option("gb_direct", 1) != 0 && h->ordered_slots && !h->dense_sorted && n_keys == 1) {
      const ldb_rel_side& ks = in->sides[(size_t) keys[0].side];
      const ldb_column& kc = ks.table->cols[(size_t) keys[0].col];
      bool any_fn = false;
      for (int32_t a = 0; a < n_aggs; a++) any_fn = any_fn || aggs[a].fn == LDB_AGG_ANY;
      const uint64_t expect = std::max<uint64_t>(est ? est : (uint64_t) in->n_rows, 1024);
      if (!any_fn && !kc.validity && !(ks.rowids && ks.may_null) && (kc.width == 4 || kc.width == 8 || kc.width == 16) && key_range <= (unsigned __int128) expect * 2 &&
          key_range <= ((unsigned __int128) 1 << 30)) {
         DAcc rows;
         memset(&rows, 0, sizeof(rows));
         rows.kind = ACC_COUNT;
         rows.count_rows = 1;
         int32_t idx;
         LDB_TRY(add_acc(rows, 1, 0, &idx)); // (an existing COUNT(*) accumulator is reused)
         nw = h->n_words;
         h->direct = 1;
         h->direct_word = h->accs[idx].word;
         h->direct_key_width = kc.width;
         direct_src = &kc;
         cap = std::max<uint64_t>(1024, next_pow2_u64((uint64_t) key_range));
      }
   }
   // One control block per call, read back ONCE per attempt: [0] the kernel's overflow / long-probe
   // flags, [1] the number of groups (total of the occupancy scan), [2 + a] NULLs of aggregate a.
   // The finalisation (occupancy → scan → k_gb_finalize → validity packing) is queued right behind
   // the aggregation kernel without waiting for its flags; an overflow (rare: the estimate was far
   // too low) throws that work away and retries with a larger table.
   const size_t ctl_bytes = 8 * (size_t) (2 + GB_MAX_OUT);
   unsigned long long* d_ctl = nullptr; // zeroed arena words, fresh ones per attempt (ldb_counters)
   unsigned long long ctl[2 + GB_MAX_OUT];
   LdbDesc<DGroupBy> d_desc(ctx); // (one per attempt: given back after the attempt's control words are read, and on every error return)
   DGroupBy* d = nullptr;
   uint64_t n_groups = 0;
   uint32_t* rep_rows = nullptr;
   std::vector<void*> out_vals((size_t) n_aggs, nullptr);
   std::vector<uint8_t*> out_valid((size_t) n_aggs, nullptr);
   std::vector<uint8_t*> bitmaps((size_t) n_aggs, nullptr);
   auto drop_outputs = [&]() {
      ldb_dev_free(ctx, rep_rows);
      rep_rows = nullptr;
      ldb_dev_free(ctx, direct_keys);
      direct_keys = nullptr;
      for (int32_t a = 0; a < n_aggs; a++) {
         ldb_dev_free(ctx, out_vals[(size_t) a]);
         ldb_dev_free(ctx, out_valid[(size_t) a]);
         ldb_dev_free(ctx, bitmaps[(size_t) a]);
         out_vals[(size_t) a] = nullptr;
         out_valid[(size_t) a] = nullptr;
         bitmaps[(size_t) a] = nullptr;
      }
   };
   for (;;) {
      h->g_cap = cap;
      h->kmult = h->ordered_slots ? (uint64_t) ((((unsigned __int128) cap) << 32) / key_range) : 0;
      uint64_t *gk = nullptr, *ga;
      uint8_t* cross = nullptr;
      LdbBufs attempt(ctx); // the table of this attempt: released when the attempt ends, on every path (an error return included)
      if (!h->dense_out) LDB_TRY(attempt.alloc(&gk, 8 * (size_t) cap));
      LDB_TRY(attempt.alloc(&ga, 8 * (size_t) cap * (size_t) nw));
      h->g_keys = (uint64_t) gk;
      h->g_acc = (uint64_t) ga;
      LDB_TRY(ldb_counters(ctx, 2 + GB_MAX_OUT, (uint64_t**) &d_ctl));
      h->g_flags = (uint64_t) d_ctl;
      // upper bound of groups = min(cap, rows) (1 for keyless); the sorted path knows the exact number
      const uint64_t max_groups = std::max<uint64_t>(1, h->keyless ? 1 : h->dense_sorted ? sorted_groups : std::min<uint64_t>(cap, (uint64_t) in->n_rows));
      h->dense_groups = h->dense_sorted ? max_groups : 0;
      if (h->dense_out) {
         LDB_TRY(attempt.alloc(&cross, (size_t) sorted_chunks + 8));
         LDB_HIP(hipMemsetAsync(cross, 0, (size_t) sorted_chunks + 8, ctx->stream));
         h->cross_flags = (uint64_t) cross;
      }
      if (h->direct || h->dense_keys) {
         LDB_TRY(ldb_dev_alloc(ctx, &direct_keys, (size_t) h->direct_key_width * (size_t) max_groups));
         h->direct_keys_out = (uint64_t) direct_keys;
      }
      if (!h->direct && h->dense_keys != 2) {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &rep_rows, 4 * (size_t) max_groups));
         h->rep_rows_out = (uint64_t) rep_rows;
         // ordered slots give up on long probe runs, which depends on the insertion order: should a REPLAYED execution
         // (ldb_readback) meet that where the recorded one did not, its group count is too high until the trace ends and the
         // execution is repeated — the representative rows beyond the real count must then still be valid row numbers
         if (h->ordered_slots && ctx->trace_mode == 2) LDB_HIP(hipMemsetAsync(rep_rows, 0, 4 * (size_t) max_groups, ctx->stream));
      }
      for (int32_t a = 0; a < n_aggs; a++) {
         LDB_TRY(ldb_dev_alloc(ctx, &out_vals[(size_t) a], (size_t) oinfo[(size_t) a].width * (size_t) max_groups));
         h->outs[a].out_values = (uint64_t) out_vals[(size_t) a];
         h->outs[a].out_valid = 0;
         if (h->outs[a].cnt_acc >= 0 || h->outs[a].cnt_rows_acc >= 0 || h->outs[a].fn == LDB_AGG_ANY) {
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &out_valid[(size_t) a], (size_t) max_groups));
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &bitmaps[(size_t) a], (size_t) ((max_groups + 7) / 8 + 1)));
            h->outs[a].out_valid = (uint64_t) out_valid[(size_t) a];
         }
      }
      LDB_TRY(d_desc.upload(h, sizeof(*h)));
      d = d_desc.p;
      hipLaunchKernelGGL(k_gb_init, dim3(ldb_grid_for(ctx, (int64_t) cap, 256, 8)), dim3(256), 0, ctx->stream, gk, ga, d);
      // COUNT(*) per key over direct slots, many rows into a table far beyond the L2: partition, then count in LDS
      const bool partitioned = h->direct && h->n_words == 1 && h->n_accs == 1 && h->n_preds == 0 && h->n_cpreds == 0 && cap >= (1ull << 20) && (cap >> GBP_SHIFT) <= GBP_MAX_PARTS &&
         in->n_rows >= ldb_option("gb_partition_min_rows", 8ll << 20) && (uint64_t) in->n_rows < (1ull << 32) && ldb_option("gb_partition", 1) != 0;
      // the same for arbitrary aggregates (k_gbp_agg): as many slots per partition as their accumulator words fit into 60 KB of LDS
      uint32_t pv_shift = 0;
      while (pv_shift < 16 && ((size_t) 8 * (size_t) nw << (pv_shift + 1)) <= 60 * 1024) pv_shift++;
      bool wide_minmax = false;
      for (int a = 0; a < h->n_accs; a++) wide_minmax = wide_minmax || h->accs[a].kind == ACC_MIN128 || h->accs[a].kind == ACC_MAX128;
      const bool part_values = !partitioned && h->direct && h->n_preds == 0 && !wide_minmax && cap >= (1ull << 20) && pv_shift >= 8 && (cap >> pv_shift) <= GBP_MAX_PARTS &&
         in->n_rows >= ldb_option("gb_partition_min_rows", 8ll << 20) && (uint64_t) in->n_rows < (1ull << 32) && ldb_option("gb_partition", 1) != 0 && ldb_option("gb_partition_values", 1) != 0;
      if (partitioned) {
         const uint32_t nparts = (uint32_t) (cap >> GBP_SHIFT);
         const uint32_t g0 = (uint32_t) std::max<int64_t>(1, std::min<int64_t>((int64_t) ctx->cus, (in->n_rows + 16383) / 16384));
         const uint64_t rows_per_wg = ((uint64_t) in->n_rows + g0 - 1) / g0;
         uint32_t *hist = nullptr, *offs = nullptr, *slots = nullptr;
         const size_t hn = (size_t) nparts * g0;
         if (ldb_option("gb_partition_wc", 1) != 0 && nparts > 64) {
            // write-combining two-pass partition (ldb_wc.hip): the slots are written once as a dense array, tile-sorted by
            // partition in LDS and stored in full-line runs — the one-pass scatter below keeps nparts open 4-byte streams per
            // workgroup (Q13: 1 024), most of whose lines leave the L2 half written (2.4 ms for 148 M rows)
            uint32_t *raw, *part = nullptr, chunks = 1;
            LdbBufs tmpb(ctx);
            LDB_TRY(tmpb.alloc(&raw, 4 * (size_t) in->n_rows));
            LDB_TRY(tmpb.alloc(&slots, 4 * (size_t) in->n_rows));
            {
               LdbProf prof_(ctx, "k_gbp_slots");
               hipLaunchKernelGGL(k_gbp_slots, dim3(ldb_grid_for(ctx, in->n_rows, 256, 8)), dim3(256), 0, ctx->stream, (const DGroupBy*) d, raw);
            }
            LDB_TRY(ldb_wc_partition(ctx, raw, nullptr, (uint64_t) in->n_rows, 0u, (uint32_t) (cap - 1), GBP_SHIFT, nparts, slots, nullptr, &part, &chunks, "k_gbp_hist", "k_gbp_scatter"));
            {
               LdbProf prof_(ctx, "k_gbp_count");
               hipLaunchKernelGGL(k_gbp_count, dim3(nparts), dim3(GBP_BLOCK), 0, ctx->stream, (const uint32_t*) slots, (const uint32_t*) part, chunks, nparts, (uint64_t) in->n_rows,
                                  ga + (uint64_t) h->direct_word * cap);
            }
            ldb_dev_free(ctx, part);
            slots = nullptr; // (owned by tmpb)
         } else {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &hist, 4 * hn));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &offs, 4 * hn));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &slots, 4 * (size_t) in->n_rows));
         {
            LdbProf prof_(ctx, "k_gbp_hist");
            hipLaunchKernelGGL(k_gbp_hist, dim3(g0), dim3(GBP_SBLOCK), 0, ctx->stream, (const DGroupBy*) d, nparts, rows_per_wg, hist);
         }
         LDB_TRY(ldb_exclusive_scan_u32(ctx, hist, offs, (int64_t) hn, nullptr));
         {
            LdbProf prof_(ctx, "k_gbp_scatter");
            hipLaunchKernelGGL(k_gbp_scatter, dim3(g0), dim3(GBP_SBLOCK), 0, ctx->stream, (const DGroupBy*) d, nparts, rows_per_wg, (const uint32_t*) offs, slots);
         }
         {
            LdbProf prof_(ctx, "k_gbp_count");
            hipLaunchKernelGGL(k_gbp_count, dim3(nparts), dim3(GBP_BLOCK), 0, ctx->stream, (const uint32_t*) slots, (const uint32_t*) offs, g0, nparts, (uint64_t) in->n_rows,
                               ga + (uint64_t) h->direct_word * cap);
         }
         }
         ldb_dev_free(ctx, hist);
         ldb_dev_free(ctx, offs);
         ldb_dev_free(ctx, slots);
      } else if (part_values) {
         // any aggregates over direct slots too many for the caches, many rows: partition (slot, row) pairs, aggregate every slot range in LDS
         const uint32_t nparts = (uint32_t) (cap >> pv_shift);
         uint32_t *raw, *slots, *prow, *part = nullptr, chunks = 1;
         LdbBufs tmpb(ctx);
         LDB_TRY(tmpb.alloc(&raw, 4 * (size_t) in->n_rows));
         LDB_TRY(tmpb.alloc(&slots, 4 * (size_t) in->n_rows));
         LDB_TRY(tmpb.alloc(&prow, 4 * (size_t) in->n_rows));
         {
            LdbProf prof_(ctx, "k_gbp_slots");
            hipLaunchKernelGGL(k_gbp_slots, dim3(ldb_grid_for(ctx, in->n_rows, 256, 8)), dim3(256), 0, ctx->stream, (const DGroupBy*) d, raw);
         }
         LDB_TRY(ldb_wc_partition(ctx, raw, nullptr, (uint64_t) in->n_rows, 0u, (uint32_t) (cap - 1), pv_shift, nparts, slots, prow, &part, &chunks, "k_gbp_hist", "k_gbp_scatter"));
         {
            LdbProf prof_(ctx, "k_gbp_agg");
            hipLaunchKernelGGL(k_gbp_agg, dim3(nparts), dim3(GBP_BLOCK), (size_t) 8 * (size_t) nw << pv_shift, ctx->stream, (const DGroupBy*) d, (const uint32_t*) slots, (const uint32_t*) prow,
                               (const uint32_t*) part, chunks, nparts, (uint64_t) in->n_rows, pv_shift);
         }
         ldb_dev_free(ctx, part);
      } else if (in->n_rows) {
         int per_cu = lds_bytes > 40 * 1024 ? 2 : 4;
         // (the sorted, LDS-free path was swept over 2 / 3 / 4 / 6 / 8 resident workgroups per CU: 5.52 / 4.22 / 3.90 / 3.70 / 3.83 ms on one box, but 6 gave 4.05 ms on the
         // next one where 4 gives ≈ 3.95: inside the box-to-box spread, so the default stays)
         if (const int64_t forced = ldb_option("gb_wgs_per_cu", 0)) per_cu = (int) std::max<int64_t>(1, std::min<int64_t>(forced, 16)); // experiments (Q1's line re-fetches)
         int grid = ldb_grid_for(ctx, in->n_rows, GB_BLOCK, per_cu);
         hipFunction_t spec = nullptr;
         if (ldb_jit_wanted(in->n_rows)) {
            std::string why;
            spec = ldb_jit_groupby(ctx->device, h, &why);
            if (!spec) ldb_set_error("groupby: specialised kernel unavailable (%s); using the generic kernel", why.c_str());
         }
         const char* prof_name = h->direct ? "k_groupby_direct" : "k_groupby"; // (the same kernel; the name tells the profile which slot scheme ran)
         if (spec) {
            LdbProf prof_(ctx, prof_name);
            void* params[] = {(void*) &d};
            LDB_HIP(hipModuleLaunchKernel(spec, (unsigned) grid, 1, 1, GB_BLOCK, 1, 1, (unsigned) lds_bytes, ctx->stream, params, nullptr));
         } else {
            LdbProf prof_(ctx, prof_name);
            hipLaunchKernelGGL(k_groupby, dim3(grid), dim3(GB_BLOCK), lds_bytes, ctx->stream, d);
         }
      }
      LDB_HIP(hipGetLastError());
      // ---- finalize (speculative: valid only if the flags come back clean)
      if (h->dense_out) { // every group inside one wave is final already: only the chunk-crossing groups are left, the count is known
         LdbProf prof_(ctx, "k_gb_finalize");
         hipLaunchKernelGGL(k_gb_finalize_cross, dim3(ldb_grid_for(ctx, (int64_t) sorted_chunks, 256, 8)), dim3(256), 0, ctx->stream, (const DGroupBy*) d, sorted_chunks,
                            (const unsigned long long*) d_sorted_groups);
         for (int32_t a = 0; a < n_aggs; a++)
            if (out_valid[(size_t) a])
               hipLaunchKernelGGL(k_pack_valid_bytes, dim3(ldb_grid_for(ctx, (int64_t) max_groups, 256 * 8, 4)), dim3(256), 0, ctx->stream, out_valid[(size_t) a], bitmaps[(size_t) a],
                                  (const unsigned long long*) d_sorted_groups, (uint64_t) max_groups, d_ctl + 2 + a);
         LDB_HIP(hipGetLastError());
      } else {
         LdbProf prof_(ctx, "k_gb_finalize");
         const int64_t n_chunks = (int64_t) ((cap + 63) / 64);
         uint32_t *pop = nullptr, *off;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &off, 4 * (size_t) n_chunks));
         const int fgrid = ldb_grid_for(ctx, (int64_t) cap, 256, 8);
         const uint64_t* occ = h->direct ? (const uint64_t*) h->g_acc + (uint64_t) h->direct_word * cap : (const uint64_t*) h->g_keys;
         if (ldb_option("scan_single_pass", 1) != 0) { // occupied slots per chunk + their exclusive scan + the group count: one chained launch
            const uint64_t n_tiles = ((uint64_t) n_chunks + GBO_TILE_CHUNKS - 1) / GBO_TILE_CHUNKS;
            ChainCall c;
            LDB_TRY(ldb_chain_begin(ctx, n_tiles, false, &c));
            hipLaunchKernelGGL(k_gb_occupancy_scan, dim3((unsigned) n_tiles), dim3(256), 0, ctx->stream, occ, cap, (uint64_t) n_chunks, n_tiles, off, d_ctl + 1, c.status, c.ticket, c.ticket_base,
                               c.epoch);
            if (hipGetLastError() != hipSuccess) return ldb_chain_failed(ctx);
         } else {
            LDB_TRY(ldb_dev_alloc(ctx, (void**) &pop, 4 * (size_t) n_chunks));
            hipLaunchKernelGGL(k_gb_occupancy, dim3(fgrid), dim3(256), 0, ctx->stream, occ, cap, pop);
            LDB_TRY(ldb_exclusive_scan_u32(ctx, pop, off, n_chunks, (uint64_t*) (d_ctl + 1)));
         }
         hipLaunchKernelGGL(k_gb_finalize, dim3(fgrid), dim3(256), 0, ctx->stream, d, rep_rows, (const uint32_t*) off);
         for (int32_t a = 0; a < n_aggs; a++)
            if (out_valid[(size_t) a])
               hipLaunchKernelGGL(k_pack_valid_bytes, dim3(ldb_grid_for(ctx, (int64_t) max_groups, 256 * 8, 4)), dim3(256), 0, ctx->stream, out_valid[(size_t) a], bitmaps[(size_t) a],
                                  (const unsigned long long*) (d_ctl + 1), (uint64_t) max_groups, d_ctl + 2 + a);
         LDB_HIP(hipGetLastError());
         ldb_dev_free(ctx, pop);
         ldb_dev_free(ctx, off);
      }
      static_assert(sizeof(ctl) <= 64 * sizeof(int64_t), "control block must fit the pinned scratch words");
      LDB_TRY(LDB_READBACK(ctx, ctl, d_ctl, ctl_bytes));
      if (h->dense_out) ctl[1] = sorted_groups; // (no occupancy scan ran: the heads pass counted the groups)
      d_desc.release();
      const uint64_t flags = (uint64_t) ctl[0];
      if ((flags & 3) == 0) break;
      drop_outputs();
      if ((flags & 2) && h->ordered_slots) { // long probe runs: this key distribution needs hashed slots
         h->ordered_slots = 0;
         in->sides[(size_t) keys[0].side].table->cols[(size_t) keys[0].col].skewed = true;
         if ((flags & 1) == 0) continue;
      }
      // global table overflowed: retry larger (the estimate was too low)
      if (cap >= cap_max) {
         ldb_dev_free(ctx, chunk_off);
         LDB_FAIL(LDB_ERR_HIP, "groupby: global table overflow at maximum capacity");
      }
      cap = std::min(cap * 8, cap_max);
   }
   ldb_dev_free(ctx, chunk_off);
   n_groups = (uint64_t) ctl[1];
   for (int32_t a = 0; a < n_aggs; a++) {
      ldb_dev_free(ctx, out_valid[(size_t) a]);
      out_valid[(size_t) a] = nullptr;
   }

   // ---- result table: key columns = gather of representative rows, then aggregates
   res->n_rows = (int64_t) n_groups;
   res->cols.resize((size_t) (n_keys + n_aggs));
   if (h->direct || h->dense_keys) { // the key column was written by k_gb_finalize / by the sorted aggregation itself
      ldb_column& kc = res->cols[0];
      kc.name = direct_src->name;
      kc.type = direct_src->type;
      kc.width = direct_src->width;
      kc.values = direct_keys;
      kc.value_bytes = (int64_t) n_groups * kc.width;
      kc.owned = true;
      if (direct_src->has_range) { // the groups' keys are values of the source column: its cached range stays a valid superset
         kc.has_range = true;
         kc.vmin = direct_src->vmin;
         kc.vmax = direct_src->vmax;
      }
      ldb_dev_free(ctx, rep_rows); // (sorted keys with an ANY aggregate: the representative rows were only read by the finalisation)
      rep_rows = nullptr;
   } else {
      ldb_rel* reps = nullptr;
      LDB_TRY(ldb_rel_select(ctx, in, rep_rows, (int64_t) n_groups, &reps));
      const int32_t s = n_keys ? ldb_gather_columns(ctx, reps, keys, n_keys, res->cols.data()) : LDB_OK;
      ldb_gpu_rel_release(ctx, reps);
      if (s != LDB_OK) return s;
   }
   for (int32_t a = 0; a < n_aggs; a++) {
      ldb_column& c = res->cols[(size_t) (n_keys + a)];
      char nm[32];
      snprintf(nm, sizeof(nm), "agg%d", a);
      c.name = nm;
      c.type = oinfo[(size_t) a].type;
      c.width = oinfo[(size_t) a].width;
      c.values = out_vals[(size_t) a];
      c.value_bytes = (int64_t) n_groups * c.width;
      c.owned = true;
      if (bitmaps[(size_t) a]) { // drop the bitmap when nothing is NULL
         const int64_t nulls = (int64_t) ctl[2 + a];
         if (nulls) {
            c.validity = bitmaps[(size_t) a];
            c.null_count = nulls;
         } else {
            ldb_dev_free(ctx, bitmaps[(size_t) a]);
         }
      }
   }
   LDB_HIP(hipGetLastError());
   *out = res.release();
   return LDB_OK;
}
