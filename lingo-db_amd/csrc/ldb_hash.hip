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
   LdbDesc<DKeys> d_desc(ctx);
   LDB_TRY(d_desc.upload(&h, sizeof(h)));
   DKeys* d = d_desc.p;
   if (in->n_rows) hipLaunchKernelGGL(k_hash_keys, dim3(ldb_grid_for(ctx, in->n_rows, 256, 8)), dim3(256), 0, ctx->stream, d, (uint64_t*) res->cols[0].values, (uint64_t) in->n_rows);
   LDB_HIP(hipGetLastError());
   d_desc.release();
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

// literal * num / den over decimal (or integer) columns, e.g. Q14's 100.00 * sum(..) / sum(..):
// DecimalMulOpLowering (reference LowerToStd.cpp:653-677) then DecimalOpScaledLowering (:631-651),
//   out = ((((num * mul) sdiv 10^mul_div) * 10^pow10) sdiv den   in wrapping 128-bit arithmetic.
__global__ void k_map_muldiv(DCol num, DCol den, i128 mul, int mul_div, int pow10, uint64_t n, i128* __restrict__ out, uint8_t* __restrict__ valid_bytes) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t rn = d_phys_row(num, i), rd = d_phys_row(den, i);
      bool ok = d_valid(num, rn) && d_valid(den, rd);
      i128 v = 0;
      if (ok) {
         const i128 d = d_load_i128(den, rd);
         if (d == 0) {
            ok = false; // the reference's sdiv by zero is undefined; NULL here
         } else {
            i128 prod = (i128) ((u128) d_load_i128(num, rn) * (u128) mul);
            if (mul_div > 0) prod = d_sdiv128(prod, d_pow10(mul_div));
            v = d_sdiv128((i128) ((u128) prod * (u128) d_pow10(pow10)), d);
         }
      }
      out[i] = v;
      valid_bytes[i] = ok ? 1 : 0;
   }
}
extern "C" int32_t ldb_gpu_map_muldiv(ldb_ctx* ctx, ldb_rel* in, ldb_colref num, int64_t mul_lo, int64_t mul_hi, int32_t mul_div_pow10, int32_t pow10, ldb_colref den,
                                      int32_t out_precision, int32_t out_scale, const char* name, ldb_table** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "map_muldiv: NULL argument");
   if (pow10 < 0 || pow10 > 38 || mul_div_pow10 < 0 || mul_div_pow10 > 38) LDB_FAIL(LDB_ERR_INVALID, "map_muldiv: exponents %d, %d", mul_div_pow10, pow10);
   LDB_TRY(ldb_rel_force(ctx, in));
   DCol dn, dd;
   LDB_TRY(ldb_make_dcol(in, num, &dn));
   LDB_TRY(ldb_make_dcol(in, den, &dd));
   auto numeric = [](const DCol& c) { return c.type == LDB_T_DECIMAL128 || c.type == LDB_T_INT64 || c.type == LDB_T_INT32 || c.type == LDB_T_INT16 || c.type == LDB_T_INT8; };
   if (!numeric(dn) || !numeric(dd)) LDB_FAIL(LDB_ERR_INVALID, "map_muldiv: decimal or integer columns expected");
   ldb_coltype t = {LDB_T_DECIMAL128, out_precision, out_scale, 0};
   const char* nm = name ? name : "ratio";
   ldb_table* res;
   LDB_TRY(ldb_gpu_table_alloc(ctx, "mapped", 1, &t, &nm, in->n_rows, nullptr, 0, &res));
   const int64_t n = in->n_rows;
   uint8_t *vb, *bm;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &vb, (size_t) (n ? n : 1)));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &bm, (size_t) ((n + 7) / 8 + 1)));
   const int grid = ldb_grid_for(ctx, n, 256, 8);
   const i128 mul = (i128) (((u128) (uint64_t) mul_hi << 64) | (uint64_t) mul_lo);
   if (n) {
      hipLaunchKernelGGL(k_map_muldiv, dim3(grid), dim3(256), 0, ctx->stream, dn, dd, mul, (int) mul_div_pow10, (int) pow10, (uint64_t) n, (i128*) res->cols[0].values, vb);
      hipLaunchKernelGGL(k_pack_bytes_to_bits, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) vb, bm, (uint64_t) n);
   }
   res->cols[0].validity = bm;
   res->cols[0].null_count = -1; // unknown (Arrow convention)
   ldb_dev_free(ctx, vb);
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
      ldb_rel_side ns{s.table, nullptr, false, s.may_null};
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
