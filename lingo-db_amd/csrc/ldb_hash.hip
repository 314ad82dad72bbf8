// ldb_hash.hip — db.hash over key columns, bit-identical to the reference's compiled hash.
// Replaces (reference): HashLowering (src/compiler/Conversion/DBToStd/LowerToStd.cpp:1065-1152),
// Hash64Lowering / HashCombineLowering / VarLenTryCheapHashLowering / HashVarLenLowering
// (src/compiler/Conversion/UtilToLLVM/LowerToLLVM.cpp:372-391,493-524).
#include "ldb_internal.h"
#include "ldb_keys.h"
#include <memory>

__global__ void k_hash_keys(const DKeys* __restrict__ d, uint64_t* __restrict__ out, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) out[i] = d_hash_keys(*d, i);
}

int32_t ldb_make_dkeys(const ldb_rel* r, const ldb_colref* keys, int32_t n_keys, DKeys* out) {
   if (n_keys < 0 || n_keys > LDB_MAX_KEYS) LDB_FAIL(LDB_ERR_UNSUPPORTED, "%d key columns (max %d)", n_keys, LDB_MAX_KEYS);
   memset(out, 0, sizeof(*out));
   out->n_keys = n_keys;
   for (int32_t k = 0; k < n_keys; k++) LDB_TRY(ldb_make_dcol(r, keys[k], &out->cols[k]));
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_hash_keys(ldb_ctx* ctx, ldb_rel* in, const ldb_colref* keys, int32_t n_keys, ldb_table** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "hash_keys: NULL argument");
   LDB_TRY(ldb_rel_force(ctx, in));
   DKeys h;
   LDB_TRY(ldb_make_dkeys(in, keys, n_keys, &h));
   ldb_coltype t = {LDB_T_INT64, 0, 0, 0};
   const char* nm = "hash";
   ldb_table* res;
   LDB_TRY(ldb_gpu_table_alloc(ctx, "hash", 1, &t, &nm, in->n_rows, nullptr, 0, &res));
   DKeys* d;
   LDB_TRY(ldb_dev_upload(ctx, &h, sizeof(h), (void**) &d));
   if (in->n_rows) hipLaunchKernelGGL(k_hash_keys, dim3(ldb_grid_for(ctx, in->n_rows, 256, 8)), dim3(256), 0, ctx->stream, d, (uint64_t*) res->cols[0].values, (uint64_t) in->n_rows);
   LDB_HIP(hipGetLastError());
   ldb_dev_free(ctx, d);
   *out = res;
   return LDB_OK;
}

// ---------------------------------------------------------------- scalar functions as computed columns
// extract(year from date): DateRuntime::extractYear (reference src/runtime/DateRuntime.cpp:99-101),
// i.e. the civil (proleptic Gregorian, UTC) year of a date32 day number; result type i64.
__device__ __forceinline__ int64_t d_extract_year(int64_t days) {
   const int64_t z = days + 719468; // days since 0000-03-01
   const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
   const int64_t doe = z - era * 146097;
   const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
   const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
   const int64_t mp = (5 * doy + 2) / 153;
   return yoe + era * 400 + (mp >= 10 ? 1 : 0);
}
__global__ void k_map_column(DCol col, int fn, uint64_t n, int64_t* __restrict__ out, uint8_t* __restrict__ valid_bytes) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t row = d_phys_row(col, i);
      const bool ok = d_valid(col, row);
      int64_t v = 0;
      if (ok) v = d_extract_year(d_load_i64(col, row)); // fn == LDB_FN_EXTRACT_YEAR (the only one so far)
      out[i] = v;
      if (valid_bytes) valid_bytes[i] = ok ? 1 : 0;
   }
}
__global__ void k_pack_bytes_to_bits(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ bitmap, uint64_t n) {
   const uint64_t nb = (n + 7) / 8;
   for (uint64_t b = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; b < nb; b += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++)
         if (b * 8 + k < n && bytes[b * 8 + k]) m |= (uint8_t) (1u << k);
      bitmap[b] = m;
   }
}

extern "C" int32_t ldb_gpu_map_column(ldb_ctx* ctx, ldb_rel* in, ldb_colref col, int32_t fn, const char* name, ldb_table** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "map_column: NULL argument");
   if (fn != LDB_FN_EXTRACT_YEAR) LDB_FAIL(LDB_ERR_UNSUPPORTED, "map_column: unknown function %d", fn);
   LDB_TRY(ldb_rel_force(ctx, in));
   DCol dc;
   LDB_TRY(ldb_make_dcol(in, col, &dc));
   if (dc.type != LDB_T_DATE32) LDB_FAIL(LDB_ERR_INVALID, "map_column: extract(year) needs a date32 column");
   ldb_coltype t = {LDB_T_INT64, 0, 0, 0};
   const char* nm = name ? name : "year";
   ldb_table* res;
   LDB_TRY(ldb_gpu_table_alloc(ctx, "mapped", 1, &t, &nm, in->n_rows, nullptr, 0, &res));
   const int64_t n = in->n_rows;
   const bool nullable = dc.validity || dc.rowids;
   uint8_t* vb = nullptr;
   if (nullable) LDB_TRY(ldb_dev_alloc(ctx, (void**) &vb, (size_t) (n ? n : 1)));
   const int grid = ldb_grid_for(ctx, n, 256, 8);
   if (n) hipLaunchKernelGGL(k_map_column, dim3(grid), dim3(256), 0, ctx->stream, dc, fn, (uint64_t) n, (int64_t*) res->cols[0].values, vb);
   if (nullable) { // NULL in → NULL out
      uint8_t* bm;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &bm, (size_t) ((n + 7) / 8 + 1)));
      if (n) hipLaunchKernelGGL(k_pack_bytes_to_bits, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) vb, bm, (uint64_t) n);
      res->cols[0].validity = bm;
      res->cols[0].null_count = -1; // unknown (Arrow convention)
      ldb_dev_free(ctx, vb);
   }
   LDB_HIP(hipGetLastError());
   *out = res;
   return LDB_OK;
}

// a relation extended by a dense table of exactly its row count (a computed column): the new side
// is the last one, with identity row ids
extern "C" int32_t ldb_gpu_rel_zip(ldb_ctx* ctx, ldb_rel* in, const ldb_table* t, ldb_rel** out) {
   if (!ctx || !in || !t || !out) LDB_FAIL(LDB_ERR_INVALID, "rel_zip: NULL argument");
   LDB_TRY(ldb_rel_force(ctx, in));
   if (t->n_rows != in->n_rows) LDB_FAIL(LDB_ERR_INVALID, "rel_zip: table has %ld rows, relation %ld", (long) t->n_rows, (long) in->n_rows);
   if (in->sides.size() + 1 > LDB_MAX_SIDES) LDB_FAIL(LDB_ERR_UNSUPPORTED, "rel_zip: more than %d sides (materialize first)", LDB_MAX_SIDES);
   ldb_rel* r = ldb_rel_new(ctx);
   r->n_rows = in->n_rows;
   for (auto& s : in->sides) {
      ldb_rel_side ns{s.table, nullptr, false};
      if (s.rowids) {
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &ns.rowids, 4 * (size_t) (in->n_rows ? in->n_rows : 1)));
         if (in->n_rows) LDB_HIP(hipMemcpyAsync(ns.rowids, s.rowids, 4 * (size_t) in->n_rows, hipMemcpyDeviceToDevice, ctx->stream));
         ns.owned = true;
      }
      r->sides.push_back(ns);
   }
   r->sides.push_back(ldb_rel_side{t, nullptr, false});
   *out = r;
   return LDB_OK;
}
