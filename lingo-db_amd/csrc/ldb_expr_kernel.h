// ldb_expr_kernel.h — device code of the scalar-expression interpreter (ldb_gpu_map_expr): a postfix program over the
// columns of a relation, evaluated per row on a small stack of nullable 128-bit integers.  Compiled ahead of time
// (generic: the program is read from memory) and, for large inputs, at run time with the program as a compile-time
// constant (ldb_jit.hip) — same source.  Reference: the per-tuple scalar code the DB dialect lowers inside `subop.map`
// (src/compiler/Conversion/DBToStd/LowerToStd.cpp:374-466, 622-699, 1022-1045).
#pragma once
#include "ldb_device.h"

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

__device__ __forceinline__ void map_expr_body(const DXProg& m, const DXProg* __restrict__ d, uint64_t n, void* __restrict__ out, uint8_t* __restrict__ valid_bytes) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      i128 st[XSTACK];
      bool nul[XSTACK];
      int sp = 0;
      const int np = m.n;
      LDB_UNROLL
      for (int k = 0; k < LDB_MAX_XPROG; k++) {
         if (k >= np) break;
         const DXInstr& x = m.ins[k];
         switch (x.op) {
            case LDB_X_COL: {
               const CV col(x.col, d->ins[k].col);
               const uint32_t row = d_phys_row(col, i);
               nul[sp] = !d_valid(col, row);
               st[sp] = nul[sp] ? (i128) 0 : d_load_i128(col, row);
               sp++;
               break;
            }
            case LDB_X_CONST:
               st[sp] = (i128) (((u128) (uint64_t) x.hi << 64) | x.lo);
               nul[sp] = false;
               sp++;
               break;
            case LDB_X_ROW:
               st[sp] = (i128) i;
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
      switch (m.out_width) {
         case 1: ((uint8_t*) out)[i] = v != 0 ? 1 : 0; break;
         case 4: ((int32_t*) out)[i] = (int32_t) v; break;
         case 8: ((int64_t*) out)[i] = (int64_t) v; break;
         default: ((i128*) out)[i] = v; break;
      }
      valid_bytes[i] = nul[0] ? 0 : 1;
   }
}

