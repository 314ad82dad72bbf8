// ldb_expr.hip — general scalar projection: computed columns from expression programs, substr.
// Replaces (reference): the per-tuple scalar code the DB dialect lowers inside `subop.map`
// (src/compiler/Conversion/DBToStd/LowerToStd.cpp): DecimalBinOpLowering / DecimalMulOpLowering /
// DecimalOpScaledLowering (:622-699), the comparison lowerings (:374-466), `scf.if` on
// db.derive_truth for CASE (:1022-1045), NULL propagation (NullHandler), and the runtime call
// StringRuntime::substr (src/runtime/StringRuntime.cpp:292-319).
//
// An expression arrives as a POSTFIX program (ldb_xinstr[]) over the columns of a relation.  It is
// evaluated per row on a small stack of nullable 128-bit integers in wrapping arithmetic — the
// value domain of every integer / decimal / date / bool the generated code computes with; the
// caller has already fixed the scales (casts are MUL_POW10 / SDIV_POW10), exactly as the frontend
// inserts db.cast before db.add / db.compare (sql_analyzer.cpp:3058-3159).
#include "ldb_internal.h"
#include "ldb_device.h"
#include <memory>

#include "ldb_expr_kernel.h"
#include "ldb_jit.h"

__global__ void k_pack_bytes_to_bits_x(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ bitmap, uint64_t n) {
   const uint64_t nb = (n + 7) / 8;
   for (uint64_t b = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; b < nb; b += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++)
         if (b * 8 + k < n && bytes[b * 8 + k]) m |= (uint8_t) (1u << k);
      bitmap[b] = m;
   }
}

__global__ void k_map_expr(const DXProg* __restrict__ prog, uint64_t n, void* __restrict__ out, uint8_t* __restrict__ valid_bytes) { map_expr_body(*prog, prog, n, out, valid_bytes); }
// run-time specialised variant (hiprtc): the program — opcodes, constants, column types — as compile-time constants, so the
// interpreter loop unrolls into straight-line code and the value stack lives in registers
static const char* XPR_SPEC_SRC =
   "extern \"C\" __global__ void k_map_expr_spec(const DXProg* __restrict__ prog, uint64_t n, void* __restrict__ out, uint8_t* __restrict__ valid_bytes) {\n"
   "   map_expr_body(LDB_META, prog, n, out, valid_bytes);\n"
   "}\n";
bool ldb_expr_jit_check(std::string* log) { // coalesce(count, 0) * 2 <= 10 over a nullable int64 column
   auto m = std::make_unique<DXProg>();
   memset(m.get(), 0, sizeof(DXProg));
   const int32_t ops[] = {LDB_X_COL, LDB_X_CONST, LDB_X_COALESCE, LDB_X_CONST, LDB_X_MUL, LDB_X_CONST, LDB_X_CMP};
   m->n = 7;
   m->out_width = 1;
   for (int k = 0; k < 7; k++) m->ins[k].op = ops[k];
   m->ins[0].col.type = LDB_T_INT64;
   m->ins[0].col.width = 8;
   m->ins[0].col.validity = 1;
   m->ins[3].lo = 2;
   m->ins[5].lo = 10;
   m->ins[6].arg = LDB_F_LTE;
   return ldb_jit_compile_only("ldb_expr_kernel.h", "DXProg", XPR_SPEC_SRC, m.get(), sizeof(DXProg), log);
}

extern "C" int32_t ldb_gpu_map_expr(ldb_ctx* ctx, ldb_rel* in, const ldb_xinstr* prog, int32_t n_instr, ldb_coltype out_type, const char* name, ldb_table** out) {
   if (!ctx || !in || !prog || !out) LDB_FAIL(LDB_ERR_INVALID, "map_expr: NULL argument");
   if (n_instr < 1 || n_instr > LDB_MAX_XPROG) LDB_FAIL(LDB_ERR_UNSUPPORTED, "map_expr: %d instructions (max %d)", n_instr, LDB_MAX_XPROG);
   switch (out_type.type) {
      case LDB_T_INT32:
      case LDB_T_INT64:
      case LDB_T_DATE32:
      case LDB_T_DECIMAL128:
      case LDB_T_BOOL8: break;
      default: LDB_FAIL(LDB_ERR_UNSUPPORTED, "map_expr: result type %d (integer, date, decimal or bool expected)", out_type.type);
   }
   LDB_TRY(ldb_rel_force(ctx, in));
   auto hp = std::make_unique<DXProg>();
   memset(hp.get(), 0, sizeof(DXProg));
   hp->n = n_instr;
   // verify the program on the host: stack depth, operand types
   int depth = 0;
   for (int32_t k = 0; k < n_instr; k++) {
      DXInstr& x = hp->ins[k];
      x.op = prog[k].op;
      x.arg = prog[k].arg;
      x.lo = (uint64_t) prog[k].lo;
      x.hi = prog[k].hi;
      int pops = 0, pushes = 1;
      switch (x.op) {
         case LDB_X_COL: {
            LDB_TRY(ldb_make_dcol(in, prog[k].col, &x.col));
            if (x.col.type == LDB_T_UTF8 || x.col.type == LDB_T_FLOAT32 || x.col.type == LDB_T_FLOAT64) LDB_FAIL(LDB_ERR_UNSUPPORTED, "map_expr: instruction %d: integer / decimal / date / bool columns only", k);
            break;
         }
         case LDB_X_CONST:
         case LDB_X_ROW: break;
         case LDB_X_ADD:
         case LDB_X_SUB:
         case LDB_X_MUL:
         case LDB_X_SDIV:
         case LDB_X_AND:
         case LDB_X_OR:
         case LDB_X_COALESCE: pops = 2; break;
         case LDB_X_CMP:
            if (x.arg < LDB_F_EQ || x.arg > LDB_F_GTE) LDB_FAIL(LDB_ERR_INVALID, "map_expr: instruction %d: bad comparison %d", k, x.arg);
            pops = 2;
            break;
         case LDB_X_MUL_POW10:
         case LDB_X_SDIV_POW10:
            if (x.arg < 0 || x.arg > 38) LDB_FAIL(LDB_ERR_INVALID, "map_expr: instruction %d: exponent %d", k, x.arg);
            pops = 1;
            break;
         case LDB_X_NEG:
         case LDB_X_NOT:
         case LDB_X_ISNULL: pops = 1; break;
         case LDB_X_SELECT: pops = 3; break;
         default: LDB_FAIL(LDB_ERR_INVALID, "map_expr: instruction %d: unknown op %d", k, x.op);
      }
      if (depth < pops) LDB_FAIL(LDB_ERR_INVALID, "map_expr: instruction %d pops %d of %d stack entries", k, pops, depth);
      depth += pushes - pops;
      if (depth > XSTACK) LDB_FAIL(LDB_ERR_UNSUPPORTED, "map_expr: stack deeper than %d", XSTACK);
   }
   if (depth != 1) LDB_FAIL(LDB_ERR_INVALID, "map_expr: the program leaves %d values (1 expected)", depth);
   const char* nm = name ? name : "expr";
   ldb_table* res;
   LDB_TRY(ldb_gpu_table_alloc(ctx, "mapped", 1, &out_type, &nm, in->n_rows, nullptr, 0, &res));
   hp->out_width = res->cols[0].width;
   const int64_t n = in->n_rows;
   uint8_t *vb, *bm;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &vb, (size_t) (n ? n : 1)));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &bm, (size_t) ((n + 7) / 8 + 1)));
   const int grid = ldb_grid_for(ctx, n, 256, 8);
   if (n) {
      LdbDesc<DXProg> d_desc(ctx);
      LDB_TRY(d_desc.upload(hp.get(), sizeof(DXProg)));
      DXProg* d = d_desc.p;
      {
         hipFunction_t spec = nullptr;
         if (ldb_jit_wanted(n)) {
            auto meta = std::make_unique<DXProg>();
            memcpy(meta.get(), hp.get(), sizeof(DXProg));
            for (int k = 0; k < LDB_MAX_XPROG; k++) ldb_jit_strip_col(meta->ins[k].col);
            std::string why;
            spec = ldb_jit_kernel(ctx->device, "ldb_expr_kernel.h", "DXProg", XPR_SPEC_SRC, "k_map_expr_spec", meta.get(), sizeof(DXProg), &why);
         }
         LdbProf prof_(ctx, "k_map_expr");
         if (spec) {
            uint64_t nn = (uint64_t) n;
            void* ov = res->cols[0].values;
            void* params[] = {(void*) &d, (void*) &nn, (void*) &ov, (void*) &vb};
            LDB_HIP(hipModuleLaunchKernel(spec, (unsigned) grid, 1, 1, 256, 1, 1, 0, ctx->stream, params, nullptr));
         } else {
            hipLaunchKernelGGL(k_map_expr, dim3(grid), dim3(256), 0, ctx->stream, (const DXProg*) d, (uint64_t) n, res->cols[0].values, vb);
         }
      }
      hipLaunchKernelGGL(k_pack_bytes_to_bits_x, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) vb, bm, (uint64_t) n);
      d_desc.release();
   }
   res->cols[0].validity = bm;
   res->cols[0].null_count = -1; // unknown (Arrow convention)
   res->cols[0].type.nullable = 1;
   ldb_dev_free(ctx, vb);
   LDB_HIP(hipGetLastError());
   *out = res;
   return LDB_OK;
}

// ---------------------------------------------------------------- substr
// StringRuntime::substr(str, from, len) (reference src/runtime/StringRuntime.cpp:292-319): positions
// count UTF-8 CHARACTERS from 1; positions before the string "count towards the length"; from / to
// beyond the end are truncated to it (charIndexToByteIndex, :102-135).
__device__ __forceinline__ uint32_t d_char_to_byte(const uint8_t* s, uint32_t byte_len, uint64_t char_index, uint32_t known_byte, uint64_t known_char) {
   for (; known_byte < byte_len; known_byte++) {
      if ((s[known_byte] >> 6) != 2) { // not a continuation byte
         if (known_char == char_index) return known_byte;
         known_char++;
      }
   }
   return byte_len;
}
__device__ __forceinline__ void d_substr_range(const uint8_t* s, uint32_t len, int64_t from, int64_t for_len, uint32_t* b0, uint32_t* b1) {
   const int64_t leg_len = for_len > 0 ? for_len : 0;
   uint64_t leg_from = (uint64_t) (from > 1 ? from : 1);
   const int64_t to_raw = from + leg_len;
   uint64_t leg_to = to_raw > (int64_t) leg_from ? (uint64_t) to_raw : leg_from;
   leg_from--;
   leg_to--;
   *b0 = d_char_to_byte(s, len, leg_from, 0, 0);
   *b1 = d_char_to_byte(s, len, leg_to, *b0, leg_from);
}
__global__ void k_substr_lens(DCol col, int64_t from, int64_t for_len, uint64_t n, int64_t* __restrict__ lens, uint8_t* __restrict__ valid_bytes) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t row = d_phys_row(col, i);
      const bool ok = d_valid(col, row);
      int64_t l = 0;
      if (ok) {
         uint32_t len, b0, b1;
         const uint8_t* s = d_load_str(col, row, &len);
         d_substr_range(s, len, from, for_len, &b0, &b1);
         l = (int64_t) (b1 - b0);
      }
      lens[i] = l;
      if (valid_bytes) valid_bytes[i] = ok ? 1 : 0;
   }
}
__global__ void k_substr_fill(DCol col, int64_t from, int64_t for_len, uint64_t n, const int64_t* __restrict__ offs, uint8_t* __restrict__ out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      const uint32_t row = d_phys_row(col, i);
      if (!d_valid(col, row)) continue;
      uint32_t len, b0, b1;
      const uint8_t* s = d_load_str(col, row, &len);
      d_substr_range(s, len, from, for_len, &b0, &b1);
      const int64_t o = offs[i];
      for (uint32_t b = b0; b < b1; b++) out[o + (b - b0)] = s[b];
   }
}

extern "C" int32_t ldb_gpu_map_substr(ldb_ctx* ctx, ldb_rel* in, ldb_colref col, int64_t from, int64_t for_len, const char* name, ldb_table** out) {
   if (!ctx || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "map_substr: NULL argument");
   LDB_TRY(ldb_rel_force(ctx, in));
   DCol dc;
   LDB_TRY(ldb_make_dcol(in, col, &dc));
   if (dc.type != LDB_T_UTF8) LDB_FAIL(LDB_ERR_INVALID, "map_substr: utf8 column expected");
   const int64_t n = in->n_rows;
   const bool nullable = dc.validity || dc.rowids;
   const int grid = ldb_grid_for(ctx, n, 256, 8);
   int64_t* lens;
   uint8_t* vb = nullptr;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &lens, 8 * (size_t) (n + 1)));
   if (nullable) LDB_TRY(ldb_dev_alloc(ctx, (void**) &vb, (size_t) (n ? n : 1)));
   if (n) hipLaunchKernelGGL(k_substr_lens, dim3(grid), dim3(256), 0, ctx->stream, dc, from, for_len, (uint64_t) n, lens, vb);
   int64_t* offs;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &offs, 8 * (size_t) (n + 1)));
   LDB_TRY(ldb_exclusive_scan_i64(ctx, lens, offs, n, offs + n));
   uint64_t total = 0;
   LDB_TRY(ldb_read_u64(ctx, offs + n, &total));
   ldb_dev_free(ctx, lens);
   ldb_coltype t = {LDB_T_UTF8, 0, 0, nullable ? 1 : 0};
   const char* nm = name ? name : "substr";
   const int64_t cap = (int64_t) total;
   ldb_table* res;
   LDB_TRY(ldb_gpu_table_alloc(ctx, "mapped", 1, &t, &nm, n, &cap, 0, &res));
   LDB_HIP(hipMemcpyAsync(res->cols[0].offsets, offs, 8 * (size_t) (n + 1), hipMemcpyDeviceToDevice, ctx->stream));
   res->cols[0].value_bytes = cap;
   if (n) hipLaunchKernelGGL(k_substr_fill, dim3(grid), dim3(256), 0, ctx->stream, dc, from, for_len, (uint64_t) n, (const int64_t*) offs, (uint8_t*) res->cols[0].values);
   ldb_dev_free(ctx, offs);
   if (nullable) {
      uint8_t* bm;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &bm, (size_t) ((n + 7) / 8 + 1)));
      if (n) hipLaunchKernelGGL(k_pack_bytes_to_bits_x, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) vb, bm, (uint64_t) n);
      res->cols[0].validity = bm;
      res->cols[0].null_count = -1;
      ldb_dev_free(ctx, vb);
   }
   LDB_HIP(hipGetLastError());
   *out = res;
   return LDB_OK;
}
