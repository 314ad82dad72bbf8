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

#define XSTACK 8
struct DXInstr {
   int32_t op;
   int32_t arg;
   DCol col;
   uint64_t lo;
   int64_t hi;
};
struct DXProg {
   int32_t n;
   int32_t out_width; // 1, 4, 8 or 16 bytes per output value
   DXInstr ins[LDB_MAX_XPROG];
};

__global__ void k_pack_bytes_to_bits_x(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ bitmap, uint64_t n) {
   const uint64_t nb = (n + 7) / 8;
   for (uint64_t b = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; b < nb; b += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++)
         if (b * 8 + k < n && bytes[b * 8 + k]) m |= (uint8_t) (1u << k);
      bitmap[b] = m;
   }
}

__global__ void k_map_expr(const DXProg* __restrict__ prog, uint64_t n, void* __restrict__ out, uint8_t* __restrict__ valid_bytes) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      i128 st[XSTACK];
      bool nul[XSTACK];
      int sp = 0;
      const int np = prog->n;
      for (int k = 0; k < np; k++) {
         const DXInstr& x = prog->ins[k];
         switch (x.op) {
            case LDB_X_COL: {
               const uint32_t row = d_phys_row(x.col, i);
               nul[sp] = !d_valid(x.col, row);
               st[sp] = nul[sp] ? (i128) 0 : d_load_i128(x.col, row);
               sp++;
               break;
            }
            case LDB_X_CONST:
               st[sp] = (i128) (((u128) (uint64_t) x.hi << 64) | x.lo);
               nul[sp] = false;
               sp++;
               break;
            case LDB_X_ADD:
            case LDB_X_SUB:
            case LDB_X_MUL:
            case LDB_X_SDIV: {
               const i128 b = st[--sp], a = st[sp - 1];
               const bool nb = nul[sp];
               bool nn = nul[sp - 1] || nb;
               i128 r = 0;
               if (!nn) {
                  if (x.op == LDB_X_ADD) r = (i128) ((u128) a + (u128) b);
                  else if (x.op == LDB_X_SUB) r = (i128) ((u128) a - (u128) b);
                  else if (x.op == LDB_X_MUL) r = (i128) ((u128) a * (u128) b);
                  else if (b == 0) nn = true; // arith.divsi by zero is undefined in the reference: NULL here
                  else r = d_sdiv128(a, b);
               }
               st[sp - 1] = r;
               nul[sp - 1] = nn;
               break;
            }
            case LDB_X_MUL_POW10: st[sp - 1] = (i128) ((u128) st[sp - 1] * (u128) d_pow10(x.arg)); break;
            case LDB_X_SDIV_POW10: st[sp - 1] = d_sdiv128(st[sp - 1], d_pow10(x.arg)); break;
            case LDB_X_NEG: st[sp - 1] = (i128) ((u128) 0 - (u128) st[sp - 1]); break;
            case LDB_X_CMP: { // arg = ldb_filter_op comparison; NULL if an operand is NULL
               const i128 b = st[--sp], a = st[sp - 1];
               nul[sp - 1] = nul[sp - 1] || nul[sp];
               st[sp - 1] = d_cmp_vals<i128>(x.arg, a, b) ? 1 : 0;
               break;
            }
            case LDB_X_AND: { // three-valued: false wins over NULL
               const i128 b = st[--sp], a = st[sp - 1];
               const bool na = nul[sp - 1], nb = nul[sp];
               const bool fa = !na && a == 0, fb = !nb && b == 0;
               nul[sp - 1] = !(fa || fb) && (na || nb);
               st[sp - 1] = (fa || fb || na || nb) ? 0 : 1;
               break;
            }
            case LDB_X_OR: { // three-valued: true wins over NULL
               const i128 b = st[--sp], a = st[sp - 1];
               const bool na = nul[sp - 1], nb = nul[sp];
               const bool ta = !na && a != 0, tb = !nb && b != 0;
               nul[sp - 1] = !(ta || tb) && (na || nb);
               st[sp - 1] = (ta || tb) ? 1 : 0;
               break;
            }
            case LDB_X_NOT: st[sp - 1] = st[sp - 1] == 0 ? 1 : 0; break;
            case LDB_X_SELECT: { // cond a b → cond (true and not NULL, db.derive_truth) ? a : b
               const i128 b = st[--sp], a = st[--sp];
               const bool nb = nul[sp + 1], na = nul[sp];
               const bool c = !nul[sp - 1] && st[sp - 1] != 0;
               st[sp - 1] = c ? a : b;
               nul[sp - 1] = c ? na : nb;
               break;
            }
            case LDB_X_ISNULL:
               st[sp - 1] = nul[sp - 1] ? 1 : 0;
               nul[sp - 1] = false;
               break;
            default: { // LDB_X_COALESCE: a b → a unless NULL
               const i128 b = st[--sp];
               const bool nb = nul[sp];
               if (nul[sp - 1]) {
                  st[sp - 1] = b;
                  nul[sp - 1] = nb;
               }
               break;
            }
         }
      }
      const i128 v = nul[0] ? (i128) 0 : st[0];
      switch (prog->out_width) {
         case 1: ((uint8_t*) out)[i] = v != 0 ? 1 : 0; break;
         case 4: ((int32_t*) out)[i] = (int32_t) v; break;
         case 8: ((int64_t*) out)[i] = (int64_t) v; break;
         default: ((i128*) out)[i] = v; break;
      }
      valid_bytes[i] = nul[0] ? 0 : 1;
   }
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
         case LDB_X_CONST: break;
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
      DXProg* d;
      LDB_TRY(ldb_dev_upload(ctx, hp.get(), sizeof(DXProg), (void**) &d));
      {
         LdbProf prof_(ctx, "k_map_expr");
         hipLaunchKernelGGL(k_map_expr, dim3(grid), dim3(256), 0, ctx->stream, (const DXProg*) d, (uint64_t) n, res->cols[0].values, vb);
      }
      hipLaunchKernelGGL(k_pack_bytes_to_bits_x, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t*) vb, bm, (uint64_t) n);
      ldb_dev_free(ctx, d);
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
