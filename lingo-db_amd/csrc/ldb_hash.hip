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
