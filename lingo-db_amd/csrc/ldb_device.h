// ldb_device.h — device-side scalar semantics shared by all kernels (gfx950 / wave64).
// Each helper cites the reference lowering whose result it must reproduce bit-exactly.
//
// Column / predicate / key-set VIEWS (CV, PV, KV) pair a metadata source `m` with an address
// source `p`.  In ahead-of-time kernels both are the same descriptor in memory; in a run-time
// specialised kernel (ldb_jit.hip) `m` is a constexpr copy, so every `c.m.type` / `p.m.op`
// test folds at compile time while the addresses still come from the launch's descriptor.
#pragma once
#include "ldb_devtypes.h"
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef __int128 i128;
typedef unsigned __int128 u128;

#define LDB_WAVE 64
// loops over descriptor counts: fully unrolled when the descriptor is a compile-time constant
#ifdef LDB_JIT_SPECIALIZED
#define LDB_UNROLL _Pragma("unroll")
#else
#define LDB_UNROLL
#endif

// Addresses travel as uint64_t; casting through address space 1 tells the compiler the memory is
// global, so it emits global_load / global_atomic instead of flat instructions.
#define LDB_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const T* gptr(uint64_t a) {
   return (const T*) (const LDB_GLOBAL T*) a;
}
template <typename T>
__device__ __forceinline__ T* gptr_mut(uint64_t a) {
   return (T*) (LDB_GLOBAL T*) a;
}

struct CV {
   const DCol& m;
   const DCol& p;
   __device__ __forceinline__ CV(const DCol& c) : m(c), p(c) {}
   __device__ __forceinline__ CV(const DCol& m_, const DCol& p_) : m(m_), p(p_) {}
};
struct PV {
   const DPred& m;
   const DPred& p;
   __device__ __forceinline__ PV(const DPred& c) : m(c), p(c) {}
   __device__ __forceinline__ PV(const DPred& m_, const DPred& p_) : m(m_), p(p_) {}
   __device__ __forceinline__ CV col() const { return CV(m.col, p.col); }
   __device__ __forceinline__ CV rhs() const { return CV(m.rhs, p.rhs); }
};

__device__ __forceinline__ uint64_t d_bswap64(uint64_t x) { return __builtin_bswap64(x); }

// util.hash64 (Hash64Lowering, reference src/compiler/Conversion/UtilToLLVM/LowerToLLVM.cpp:493-503):
// m = x * 0x9E3779B97F4A7C55; m ^ bswap64(m)
__device__ __forceinline__ uint64_t d_hash64(uint64_t x) {
   uint64_t m = x * 11400714819323198549ull;
   return m ^ d_bswap64(m);
}
// util.hash_combine(new, total) = new ^ bswap64(total) (HashCombineLowering, LowerToLLVM.cpp:505-512)
__device__ __forceinline__ uint64_t d_hash_combine(uint64_t h_new, uint64_t total) { return h_new ^ d_bswap64(total); }

// ---- XXH64, seed 0 (hashVarLenData → llvm::xxHash64, reference src/runtime/Hash.cpp:13-16)
#define DXP1 11400714785074694791ULL
#define DXP2 14029467366897019727ULL
#define DXP3 1609587929392839161ULL
#define DXP4 9650029242287828579ULL
#define DXP5 2870177450012600261ULL
__device__ __forceinline__ uint64_t d_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t d_rd64(const uint8_t* p) {
   uint64_t v = 0;
#pragma unroll
   for (int i = 0; i < 8; i++) v |= (uint64_t) p[i] << (8 * i);
   return v;
}
__device__ __forceinline__ uint32_t d_rd32(const uint8_t* p) {
   return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24);
}
__device__ __forceinline__ uint64_t d_xround(uint64_t acc, uint64_t in) {
   acc += in * DXP2;
   acc = d_rotl64(acc, 31);
   return acc * DXP1;
}
__device__ __forceinline__ uint64_t d_xmerge(uint64_t acc, uint64_t val) {
   val = d_xround(0, val);
   acc ^= val;
   return acc * DXP1 + DXP4;
}
__device__ inline uint64_t d_xxh64(const uint8_t* p, uint64_t len) {
   const uint8_t* end = p + len;
   uint64_t h;
   if (len >= 32) {
      const uint8_t* limit = end - 32;
      uint64_t v1 = DXP1 + DXP2, v2 = DXP2, v3 = 0, v4 = 0 - DXP1;
      do {
         v1 = d_xround(v1, d_rd64(p));
         v2 = d_xround(v2, d_rd64(p + 8));
         v3 = d_xround(v3, d_rd64(p + 16));
         v4 = d_xround(v4, d_rd64(p + 24));
         p += 32;
      } while (p <= limit);
      h = d_rotl64(v1, 1) + d_rotl64(v2, 7) + d_rotl64(v3, 12) + d_rotl64(v4, 18);
      h = d_xmerge(h, v1);
      h = d_xmerge(h, v2);
      h = d_xmerge(h, v3);
      h = d_xmerge(h, v4);
   } else {
      h = DXP5;
   }
   h += len;
   while (p + 8 <= end) {
      h ^= d_xround(0, d_rd64(p));
      h = d_rotl64(h, 27) * DXP1 + DXP4;
      p += 8;
   }
   if (p + 4 <= end) {
      h ^= (uint64_t) d_rd32(p) * DXP1;
      h = d_rotl64(h, 23) * DXP2 + DXP3;
      p += 4;
   }
   while (p < end) {
      h ^= (*p) * DXP5;
      h = d_rotl64(h, 11) * DXP1;
      p++;
   }
   h ^= h >> 33;
   h *= DXP2;
   h ^= h >> 29;
   h *= DXP3;
   h ^= h >> 32;
   return h;
}
// VarLenTryCheapHash (LowerToLLVM.cpp:372-391): len <= 12 → hash64(first64) ^ bswap(hash64(last64))
// over the 16-byte VarLen32 image {len:u32, bytes[12] zero padded} (helpers.h:194-209); else XXH64.
__device__ inline uint64_t d_hash_varlen(const uint8_t* p, uint32_t len) {
   if (len > 12) return d_xxh64(p, len);
   uint64_t first64 = len, last64 = 0;
#pragma unroll
   for (int i = 0; i < 4; i++)
      if ((uint32_t) i < len) first64 |= (uint64_t) p[i] << (32 + 8 * i);
#pragma unroll
   for (int i = 0; i < 8; i++)
      if ((uint32_t) (i + 4) < len) last64 |= (uint64_t) p[i + 4] << (8 * i);
   return d_hash64(first64) ^ d_bswap64(d_hash64(last64));
}

// ---- column access (LoadArrowOpLowering, reference src/compiler/Conversion/DBToStd/LowerToStd.cpp:111-209)
// (presence of row ids / a validity bitmap is metadata: the specialised kernel's constexpr copy
// keeps non-zero flags in those fields, the addresses themselves come from c.p)
__device__ __forceinline__ uint32_t d_phys_row(CV c, uint64_t i) { return c.m.rowids ? gptr<uint32_t>(c.p.rowids)[i] : (uint32_t) i; }
__device__ __forceinline__ bool d_valid(CV c, uint32_t row) {
   if (c.m.rowids && row == LDB_NULL_ROW) return false; // only row-id vectors can carry outer-join padding
   if (!c.m.validity) return true;
   return (gptr<uint8_t>(c.p.validity)[row >> 3] >> (row & 7)) & 1;
}
// 64-bit view of a fixed-width value (sign-extended); decimal128 with p<19 is truncated to
// i64 exactly as the generated code does (LowerToStd.cpp:128-132)
__device__ __forceinline__ int64_t d_load_i64(CV c, uint32_t row) {
   switch (c.m.width) {
      case 4: return gptr<int32_t>(c.p.values)[row];
      case 8: return gptr<int64_t>(c.p.values)[row];
      case 16: return gptr<int64_t>(c.p.values)[2 * (uint64_t) row];
      case 2: return gptr<int16_t>(c.p.values)[row];
      default: return c.m.type == LDB_T_BOOL8 ? (gptr<uint8_t>(c.p.values)[row] ? 1 : 0) : gptr<int8_t>(c.p.values)[row];
   }
}
__device__ __forceinline__ bool d_is_wide(CV c) { return c.m.type == LDB_T_DECIMAL128 && c.m.precision >= 19; }
__device__ __forceinline__ bool d_is_flt(CV c) { return c.m.type == LDB_T_FLOAT64 || c.m.type == LDB_T_FLOAT32; }
__device__ __forceinline__ i128 d_load_i128(CV c, uint32_t row) {
   if (c.m.width == 16) {
      const uint64_t* p = gptr<uint64_t>(c.p.values) + 2 * (uint64_t) row;
      uint64_t lo = p[0], hi = p[1];
      if (c.m.precision < 19) return (i128) (int64_t) lo;
      return (i128) (((u128) hi << 64) | lo);
   }
   return (i128) d_load_i64(c, row);
}
__device__ __forceinline__ double d_load_f64(CV c, uint32_t row) {
   if (c.m.type == LDB_T_FLOAT64) return gptr<double>(c.p.values)[row];
   if (c.m.type == LDB_T_FLOAT32) return gptr<float>(c.p.values)[row];
   return (double) d_load_i64(c, row);
}
__device__ __forceinline__ const uint8_t* d_load_str(CV c, uint32_t row, uint32_t* len) {
   const int64_t* o = gptr<int64_t>(c.p.offsets);
   int64_t b = o[row], e = o[row + 1];
   *len = (uint32_t) (e - b);
   return gptr<uint8_t>(c.p.values) + b;
}
// std::string_view compare: unsigned bytewise then length (reference StringRuntime.cpp:242-256)
__device__ inline int d_str_cmp(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
   uint32_t m = la < lb ? la : lb;
   for (uint32_t i = 0; i < m; i++) {
      if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
   }
   return la < lb ? -1 : (la > lb ? 1 : 0);
}
// SQL LIKE with the reference's semantics (StringRuntime::like → iterativeLike,
// src/runtime/StringRuntime.cpp:28-93, escape '\\'): a character = UTF-8 lead byte + continuation
// bytes, two characters are equal when their lead bytes are, '_' = one character, '%' = any number;
// an escape right after a %/_ run is stepped over and the character behind it matches like an
// unescaped one; a pattern ending in a lone escape never matches.  The reference recurses at every
// '%'; here one restart point (the latest '%') is kept — for LIKE patterns retrying from the most
// recent '%' is equivalent — so no device stack is needed.
__device__ __forceinline__ uint32_t d_utf8_len(const uint8_t* p, uint32_t left) {
   uint32_t k = 1;
   while (k < left && (p[k] >> 6) == 2) k++;
   return k;
}
// Bytes of one string through an 8-byte window: a matcher that walks the string byte by byte then
// issues one (unaligned) 8-byte global load per 8 bytes instead of 8 dependent byte loads — the
// LIKE scan over 20 M part names was latency-bound on those (5.3 ms).  The window may read up to
// 7 bytes past the string: every device buffer carries that slack (ldb_dev_alloc).
struct d_bytes8 {
   const uint8_t* b;
   uint32_t wbase = 0xFFFFFFFFu;
   uint64_t w = 0;
   __device__ __forceinline__ explicit d_bytes8(const uint8_t* base) : b(base) {}
   __device__ __forceinline__ uint8_t operator[](uint32_t k) {
      if ((k & ~7u) != wbase) {
         wbase = k & ~7u;
         typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
         w = *(const LDB_GLOBAL u64_unaligned*) (b + wbase);
      }
      return (uint8_t) (w >> (8 * (k & 7u)));
   }
};
// the same interface over bytes staged in LDS (see d_eval_conj_batch: a wave copies the contiguous
// bytes of its 64 strings into LDS with coalesced loads, then every lane matches its own string)
#define LDB_LDS __attribute__((address_space(3)))
struct d_bytes_lds {
   const LDB_LDS uint8_t* b;
   __device__ __forceinline__ explicit d_bytes_lds(const uint8_t* base) : b((const LDB_LDS uint8_t*) base) {}
   __device__ __forceinline__ uint8_t operator[](uint32_t k) const { return b[k]; }
};
// plain bytes behind a generic pointer (the LIKE pattern in the descriptor: every access is a
// memory load — the staged path copies the pattern into LDS instead, see d_eval_conj_batch)
struct d_bytes_mem {
   const uint8_t* b;
   __device__ __forceinline__ explicit d_bytes_mem(const uint8_t* base) : b(base) {}
   __device__ __forceinline__ uint8_t operator[](uint32_t k) const { return b[k]; }
};
template <typename S>
__device__ __forceinline__ uint32_t d_utf8_len_at(S& s, uint32_t at, uint32_t sl) {
   uint32_t k = 1;
   while (at + k < sl && (s[at + k] >> 6) == 2) k++;
   return k;
}
// (not inlined: a batch evaluator calls it once per batch row, and four to eight inlined copies of
// the matcher made a 25 k-line scan kernel that no longer fits the instruction cache)
template <typename S, typename P>
__device__ __attribute__((noinline)) bool d_like_on(S& s, uint32_t sl, P& p, uint32_t pl) {
   uint32_t si = 0, pi = 0, star_p = 0, star_s = 0;
   bool have_star = false;
   for (;;) {
      bool mismatch = false;
      if (pi < pl && si < sl) {
         const uint8_t pc = p[pi];
         if (pc == '%') {
            pi++;
            while (pi < pl && (p[pi] == '%' || p[pi] == '_')) {
               if (p[pi] == '_') {
                  if (si >= sl) return false;
                  si += d_utf8_len_at(s, si, sl);
               }
               pi++;
            }
            if (pi >= pl) return true;
            if (p[pi] == '\\') {
               pi += d_utf8_len_at(p, pi, pl);
               if (pi >= pl) return false;
            }
            have_star = true;
            star_p = pi;
            star_s = si;
            continue;
         } else if (pc == '\\') {
            uint32_t q = pi + d_utf8_len_at(p, pi, pl);
            if (q >= pl || p[q] != s[si]) {
               mismatch = true;
            } else {
               si += d_utf8_len_at(s, si, sl);
               pi = q + d_utf8_len_at(p, q, pl);
            }
         } else if (pc == '_' || pc == s[si]) {
            si += d_utf8_len_at(s, si, sl);
            pi += d_utf8_len_at(p, pi, pl);
         } else {
            mismatch = true;
         }
      } else if (si >= sl) {
         while (pi < pl && p[pi] == '%') pi++;
         return pi >= pl;
      } else {
         mismatch = true; // pattern used up, string is not
      }
      if (mismatch) {
         if (!have_star) return false;
         star_s += d_utf8_len_at(s, star_s, sl);
         // skip ahead to the next place the literal behind the '%' can start.  Byte-wise is exact:
         // a character boundary whose lead byte differs would fail its first comparison anyway, and
         // continuation bytes (0x80-0xBF) never equal an ASCII or lead byte.  Restarting the general
         // loop at every position cost ~55 instructions per position ('%green%' over 20 M part
         // names: 4.2 ms); this loop costs a handful.
         const uint8_t c = p[star_p];
         if (c != '_' && c != '%' && c != '\\' && (c < 0x80 || c >= 0xC0))
            while (star_s < sl && s[star_s] != c) star_s++;
         if (star_s >= sl) return false;
         si = star_s;
         pi = star_p;
      }
   }
}

__device__ inline bool d_like(const uint8_t* sptr, uint32_t sl, const uint8_t* pattern, uint32_t pl) {
   d_bytes8 s(sptr);
   d_bytes_mem p(pattern);
   return d_like_on(s, sl, p, pl);
}

__device__ __forceinline__ bool d_cmp_apply(int op, int c3) {
   switch (op) {
      case LDB_F_EQ: return c3 == 0;
      case LDB_F_NEQ: return c3 != 0;
      case LDB_F_LT: return c3 < 0;
      case LDB_F_LTE: return c3 <= 0;
      case LDB_F_GT: return c3 > 0;
      default: return c3 >= 0;
   }
}
template <typename T>
__device__ __forceinline__ bool d_cmp_vals(int op, T a, T b) {
   switch (op) {
      case LDB_F_EQ: return a == b;
      case LDB_F_NEQ: return a != b;
      case LDB_F_LT: return a < b;
      case LDB_F_LTE: return a <= b;
      case LDB_F_GT: return a > b;
      default: return a >= b;
   }
}

// can ANY value in [zlo, zhi] satisfy `value op c`?  (zone maps: per-chunk min / max, the reference's counterpart is the
// per-chunk statistics a scan uses to skip chunks)
__device__ __forceinline__ bool d_zone_may_pass(int op, int64_t zlo, int64_t zhi, int64_t c) {
   switch (op) {
      case LDB_F_EQ: return zlo <= c && c <= zhi;
      case LDB_F_NEQ: return !(zlo == c && zhi == c);
      case LDB_F_LT: return zlo < c;
      case LDB_F_LTE: return zlo <= c;
      case LDB_F_GT: return zhi > c;
      default: return zhi >= c; // LDB_F_GTE
   }
}
// One conjunct on logical row i.  Semantics: Filter impls of reference
// src/runtime/storage/Restrictions.cpp:67-321 (native-type compare, decimals as __int128,
// strings as string_view, IN = membership); NULL operands fail.  All branches on p.m.* are
// wave-uniform (descriptor in scalar registers / constant memory, or folded when specialised).
__device__ __forceinline__ bool d_eval_pred(PV p, uint64_t i) {
   const CV col = p.col();
   uint32_t row = d_phys_row(col, i);
   bool valid = d_valid(col, row);
   if (p.m.op == LDB_F_NOTNULL) return valid;
   if (!valid) return false;
   const int type = p.m.col.type;
   if (p.m.rhs_kind == LDB_RHS_COLUMN) {
      const CV rhs = p.rhs();
      uint32_t row2 = d_phys_row(rhs, i);
      if (!d_valid(rhs, row2)) return false;
      if (type == LDB_T_UTF8) {
         uint32_t la, lb;
         const uint8_t* a = d_load_str(col, row, &la);
         const uint8_t* b = d_load_str(rhs, row2, &lb);
         return d_cmp_apply(p.m.op, d_str_cmp(a, la, b, lb));
      }
      if (d_is_flt(col) || d_is_flt(rhs)) return d_cmp_vals<double>(p.m.op, d_load_f64(col, row), d_load_f64(rhs, row2));
      if (d_is_wide(col) || d_is_wide(rhs)) return d_cmp_vals<i128>(p.m.op, d_load_i128(col, row), d_load_i128(rhs, row2));
      return d_cmp_vals<int64_t>(p.m.op, d_load_i64(col, row), d_load_i64(rhs, row2));
   }
   if (type == LDB_T_UTF8) {
      uint32_t la;
      const uint8_t* a = d_load_str(col, row, &la);
      if (p.m.op == LDB_F_IN) {
         for (int k = 0; k < p.m.n_in; k++) {
            uint32_t lb = (uint32_t) (p.m.in_off[k + 1] - p.m.in_off[k]);
            if (d_str_cmp(a, la, (const uint8_t*) p.m.in_blob + p.m.in_off[k], lb) == 0) return true;
         }
         return false;
      }
      if (p.m.op == LDB_F_LIKE || p.m.op == LDB_F_NOT_LIKE) return d_like(a, la, (const uint8_t*) p.m.str, (uint32_t) p.m.str_len) == (p.m.op == LDB_F_LIKE);
      return d_cmp_apply(p.m.op, d_str_cmp(a, la, (const uint8_t*) p.m.str, (uint32_t) p.m.str_len));
   }
   if (d_is_flt(col)) {
      double a = d_load_f64(col, row);
      if (p.m.op == LDB_F_IN) {
         for (int k = 0; k < p.m.n_in; k++)
            if (a == __longlong_as_double((long long) p.m.in_lo[k])) return true;
         return false;
      }
      return d_cmp_vals<double>(p.m.op, a, p.m.f);
   }
   if (d_is_wide(col)) {
      i128 a = d_load_i128(col, row);
      if (p.m.op == LDB_F_IN) {
         for (int k = 0; k < p.m.n_in; k++)
            if (a == (i128) (((u128) (uint64_t) p.m.in_hi[k] << 64) | p.m.in_lo[k])) return true;
         return false;
      }
      return d_cmp_vals<i128>(p.m.op, a, (i128) (((u128) (uint64_t) p.m.hi << 64) | p.m.lo));
   }
   // narrow integer path: the constant is a 128-bit value; a column value (fits i64) compares
   // against it exactly after placing the constant relative to the i64 range
   if (p.m.zmin) { // zone map (only ever attached to dense columns: row == i)
      const uint64_t z = i >> LDB_ZONE_SHIFT;
      if (!d_zone_may_pass(p.m.op, gptr<int64_t>(p.p.zmin)[z], gptr<int64_t>(p.p.zmax)[z], (int64_t) p.m.lo)) return false;
   }
   int64_t a = d_load_i64(col, row);
   if (p.m.op == 100) { // LDB_F_CODESET: dictionary code of a utf8 column against the set of codes the string predicate accepts
      const uint32_t c = (uint32_t) a;
      return c < 1024u && (((uint8_t) p.m.in_blob[c >> 3] >> (c & 7u)) & 1u);
   }
   if (p.m.op == LDB_F_IN) {
      for (int k = 0; k < p.m.n_in; k++) {
         bool fits = (p.m.in_hi[k] == ((int64_t) p.m.in_lo[k] >> 63));
         if (fits && a == (int64_t) p.m.in_lo[k]) return true;
      }
      return false;
   }
   if (p.m.hi == ((int64_t) p.m.lo >> 63)) return d_cmp_vals<int64_t>(p.m.op, a, (int64_t) p.m.lo);
   return d_cmp_apply(p.m.op, p.m.hi < 0 ? 1 : -1); // constant below / above every int64
}

// A "simple" conjunct: narrow integer-like column (ints, date32, char(1), decimal p<19) against a
// constant with a comparison operator — the shape of almost every pushed-down TPC-H filter.
__device__ __forceinline__ bool d_pred_is_simple(const DPred& pm) {
   const int t = pm.col.type;
   const bool narrow = t == LDB_T_INT8 || t == LDB_T_INT16 || t == LDB_T_INT32 || t == LDB_T_INT64 || t == LDB_T_DATE32 || t == LDB_T_CHAR4 || t == LDB_T_BOOL8 ||
                       (t == LDB_T_DECIMAL128 && pm.col.precision < 19);
   return narrow && pm.rhs_kind == LDB_RHS_INT && pm.op != LDB_F_IN && pm.op != LDB_F_NOTNULL;
}

__device__ __forceinline__ bool d_type_is_narrow_int(const DCol& c) {
   const int t = c.type;
   return t == LDB_T_INT8 || t == LDB_T_INT16 || t == LDB_T_INT32 || t == LDB_T_INT64 || t == LDB_T_DATE32 || t == LDB_T_CHAR4 || t == LDB_T_BOOL8 ||
          (t == LDB_T_DECIMAL128 && c.precision < 19);
}
// column-vs-column comparison of two dense (no row ids, no NULLs) narrow integer columns
__device__ __forceinline__ bool d_pred_is_colcol_dense(const DPred& pm) {
   return pm.rhs_kind == LDB_RHS_COLUMN && pm.op != LDB_F_IN && pm.op != LDB_F_NOTNULL && d_type_is_narrow_int(pm.col) && d_type_is_narrow_int(pm.rhs) && !pm.col.rowids &&
          !pm.col.validity && !pm.rhs.rowids && !pm.rhs.validity;
}

// A conjunction over U rows of the same thread, predicate-major, load phase separated from the
// compare phase: the U loads of one conjunct are independent and issue back to back under their
// own exec masks (memory-level parallelism: one memory round trip per conjunct column for the
// whole batch) instead of one load → wait → compare → branch chain per row and conjunct.
// (Branching on pass[u] around the loads instead gets jump-threaded into one code path per
// pass/fail combination: a 25k-line kernel that ran 2x slower than the row-major one.)
// Consecutive simple conjuncts on the same column (`same_col`, set by the host: l_shipdate >= a
// AND l_shipdate < b) share one load.  Rows with pass[u] == false are not loaded; a wave whose
// rows all failed skips the loads (execz).  Non-simple shapes go through d_eval_pred.
// `stage` (optional): LDS_STR_STAGE bytes of LDS owned by the calling WAVE, given only by callers
// whose lanes hold 64 CONSECUTIVE rows in every batch slot (rows[u] = r0 + lane).  LIKE conjuncts
// over a dense utf8 column then stage the wave's strings — contiguous in the value buffer — with
// coalesced 8-byte loads and match from LDS: one memory round trip per 64 strings instead of ~5
// dependent ones per string (the matcher walks its string through an 8-byte window otherwise).
#define LDS_STR_STAGE 6144
// layout of a wave's stage: [0, LDS_STR_STAGE + 32) the strings (+ readable slack for the 24-byte
// windows of the position-parallel matcher), then one match bitmap per pattern segment, then the pattern
#define LDS_LIKE_STRIDE (LDS_STR_STAGE / 8 + 8)
#define LDS_STR_BITS (LDS_STR_STAGE + 32)
#define LDS_STR_PAT (LDS_STR_BITS + LDB_LIKE_MAX_SEG * LDS_LIKE_STRIDE)
#define LDS_STR_BYTES (LDS_STR_PAT + 64)

// (round 6, measured on Q13's scan — 3.33 ms with the funnel-shift compares below: v_mqsad_u32_u8 pieces 3.57 ms, a rarest-byte SWAR prefilter 4.07 ms,
// first segment by position + later segments row-cooperative 3.89 ms.  The instruction count of the MQSAD form is a third, its issue rate is not;
// kept as a compile-time alternative: LDB_JIT_DEFINES=-DLDB_LIKE_MQSAD=1)
#ifndef LDB_LIKE_MQSAD
#define LDB_LIKE_MQSAD 0
#endif
typedef unsigned int ldb_u32x4 __attribute__((ext_vector_type(4)));
// the first 16 bytes of pattern segment [off, off + len) as two little-endian words and their masks
__device__ __forceinline__ void d_like_seg_words(const char* str, int off, int len, uint64_t (&pat)[2], uint64_t (&msk)[2]) {
   pat[0] = pat[1] = msk[0] = msk[1] = 0;
#pragma unroll
   for (int k = 0; k < 16; k++)
      if (k < len) {
         pat[k >> 3] |= (uint64_t) (uint8_t) str[off + k] << (8 * (k & 7));
         msk[k >> 3] |= 0xFFull << (8 * (k & 7));
      }
}
// Position-parallel LIKE for "simple" patterns (host: ldb_like_plan; all lanes of the wave call it).
// `stage` holds `span` contiguous bytes: the strings of the wave's 64 rows; this lane's row is
// [row_off, row_off + row_len).  Phase A: the lanes split the POSITIONS — lane l takes bytes
// 8(l + 64t) … +7, reads three 8-byte words and compares the eight byte-shifted windows with every
// segment: one match bit per position and segment, exactly balanced, no divergence, ~13 instructions
// per byte (the row-per-lane matcher walks its string byte by byte through the general pattern
// automaton: ~200 lane-instructions per byte on TPC-H Q13's o_comment once divergence is counted).
// Phase B: each lane places the segments of its row left to right with find-first-set over the
// bitmaps (greedy leftmost placement decides %A%B% patterns; an anchored first / last segment has
// one admissible position).  Bits computed from bytes beyond `span` are garbage but lie beyond every
// row's last admissible position.
__device__ __forceinline__ bool d_like_simple_wave(const DPred& m, uint8_t* stage, uint32_t span, uint32_t row_off, uint32_t row_len, bool active) {
   const uint32_t lane = threadIdx.x & 63;
   const int nseg = m.n_in;
   const LDB_LDS uint8_t* text = (const LDB_LDS uint8_t*) stage;
   LDB_LDS uint8_t* bits = (LDB_LDS uint8_t*) stage + LDS_STR_BITS;
   const uint32_t nb = (span + 7) >> 3;
   for (uint32_t b = lane; b < nb; b += 64) {
      const uint64_t w0 = *(const LDB_LDS uint64_t*) (text + 8 * b), w1 = *(const LDB_LDS uint64_t*) (text + 8 * b + 8), w2 = *(const LDB_LDS uint64_t*) (text + 8 * b + 16);
      LDB_UNROLL
      for (int j = 0; j < LDB_LIKE_MAX_SEG; j++) {
         if (j >= nseg) break;
         uint64_t pat[2], msk[2];
         const int len = m.in_off[2 * j + 1];
         d_like_seg_words(m.str, m.in_off[2 * j], len, pat, msk);
         uint32_t found = 0;
#if LDB_LIKE_MQSAD
         // Round 6: v_mqsad_u32_u8 — four masked sums of absolute byte differences of a 4-byte reference against the four byte offsets of an
         // 8-byte window, accumulated — does the work of four shifted masked compares in one instruction: the segment is cut into 4-byte pieces
         // (zero-padded: a zero reference byte is masked out, and no pattern byte is zero), the window of piece p at offsets 4g … 4g + 3 is simply
         // the register pair (t[p + g], t[p + g + 1]) of the block's 24 text bytes, no shifting; sum == 0 ⇔ the segment matches at that offset.
         // Before: eight funnel-shifted 64-bit masked compares per segment and block (~13 instructions per text byte; the kernel was VALU-bound).
         const uint32_t t[6] = {(uint32_t) w0, (uint32_t) (w0 >> 32), (uint32_t) w1, (uint32_t) (w1 >> 32), (uint32_t) w2, (uint32_t) (w2 >> 32)};
         const uint32_t piece[4] = {(uint32_t) pat[0], (uint32_t) (pat[0] >> 32), (uint32_t) pat[1], (uint32_t) (pat[1] >> 32)};
         const int n_piece = (len + 3) >> 2;
#pragma unroll
         for (int g = 0; g < 2; g++) {
            ldb_u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int pc = 0; pc < 4; pc++)
               if (pc < n_piece) acc = __builtin_amdgcn_mqsad_u32_u8((uint64_t) t[pc + g] | ((uint64_t) t[pc + g + 1] << 32), piece[pc], acc);
            found |= (acc[0] == 0u ? 1u : 0u) << (4 * g) | (acc[1] == 0u ? 2u : 0u) << (4 * g) | (acc[2] == 0u ? 4u : 0u) << (4 * g) | (acc[3] == 0u ? 8u : 0u) << (4 * g);
         }
#else
#pragma unroll
         for (int o = 0; o < 8; o++) {
            const uint64_t x0 = o ? (w0 >> (8 * o)) | (w1 << (64 - 8 * o)) : w0;
            bool eq = (x0 & msk[0]) == pat[0];
            if (len > 8) {
               const uint64_t x1 = o ? (w1 >> (8 * o)) | (w2 << (64 - 8 * o)) : w1;
               eq = eq && (x1 & msk[1]) == pat[1];
            }
            found |= (eq ? 1u : 0u) << o;
         }
#endif
         bits[j * LDS_LIKE_STRIDE + b] = (uint8_t) found;
      }
   }
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
   bool ok = active;
   if (active) {
      uint32_t pos = row_off;
      const uint32_t row_end = row_off + row_len;
      LDB_UNROLL
      for (int j = 0; j < LDB_LIKE_MAX_SEG; j++) {
         if (j >= nseg || !ok) break;
         const uint32_t len = (uint32_t) m.in_off[2 * j + 1];
         if (pos + len > row_end) {
            ok = false;
            break;
         }
         const uint32_t lo = (j == nseg - 1 && (m.lo & 2)) ? row_end - len : pos; // anchored end: the last segment closes the row
         const uint32_t hi = (j == 0 && (m.lo & 1)) ? pos : row_end - len; // anchored start: the first segment opens it
         if (lo > hi) {
            ok = false;
            break;
         }
         const LDB_LDS uint64_t* M = (const LDB_LDS uint64_t*) (bits + j * LDS_LIKE_STRIDE);
         uint32_t w = lo >> 6, at = 0xFFFFFFFFu;
         uint64_t word = M[w] & (~0ull << (lo & 63));
         for (;;) {
            if (word) {
               at = (w << 6) + (uint32_t) __builtin_ctzll(word);
               break;
            }
            w++;
            if ((w << 6) > hi) break;
            word = M[w];
         }
         if (at > hi) ok = false; // (no bit: at = 0xFFFFFFFF)
         pos = at + len;
      }
   }
   __builtin_amdgcn_wave_barrier(); // the next batch slot overwrites the stage
   return ok;
}
template <int U>
__device__ __forceinline__ void d_eval_conj_batch(const DPred* mp, const DPred* __restrict__ dp, int np, const uint64_t (&rows)[U], bool (&pass)[U], uint64_t n_rows = 0,
                                                  uint8_t* stage = nullptr) {
   int64_t val[U];
   bool ok[U];
#pragma unroll
   for (int u = 0; u < U; u++) {
      val[u] = 0;
      ok[u] = false;
   }
   LDB_UNROLL
   for (int p = 0; p < np; p++) {
      const PV pv(mp[p], dp[p]);
      if (d_pred_is_simple(mp[p])) {
         const bool reuse = p > 0 && mp[p].same_col && d_pred_is_simple(mp[p - 1]);
         if (mp[p].zmin) { // zone map attached (dense column, constant fits int64): rows of excluded zones fail before the load
            const int64_t* zlo = gptr<int64_t>(dp[p].zmin);
            const int64_t* zhi = gptr<int64_t>(dp[p].zmax);
#pragma unroll
            for (int u = 0; u < U; u++)
               if (pass[u]) {
                  const uint64_t z = rows[u] >> LDB_ZONE_SHIFT;
                  pass[u] = d_zone_may_pass(mp[p].op, zlo[z], zhi[z], (int64_t) mp[p].lo);
               }
         }
         if (!reuse) {
            const CV col = pv.col();
            if (!col.m.rowids && !col.m.validity) {
               // dense column: a failed row loads row 0 instead (one broadcast line, no branch), so
               // the batch's loads carry no control dependence at all.  Callers guarantee n >= 1.
#pragma unroll
               for (int u = 0; u < U; u++) {
                  ok[u] = true;
                  val[u] = d_load_i64(col, pass[u] ? (uint32_t) rows[u] : 0u);
               }
            } else {
#pragma unroll
               for (int u = 0; u < U; u++) {
                  ok[u] = false;
                  val[u] = 0;
                  if (pass[u]) {
                     uint32_t row = d_phys_row(col, rows[u]);
                     ok[u] = d_valid(col, row);
                     if (ok[u]) val[u] = d_load_i64(col, row);
                  }
               }
            }
         }
         const bool fits = mp[p].hi == ((int64_t) mp[p].lo >> 63);
#pragma unroll
         for (int u = 0; u < U; u++) {
            const bool c = fits ? d_cmp_vals<int64_t>(mp[p].op, val[u], (int64_t) mp[p].lo) : d_cmp_apply(mp[p].op, mp[p].hi < 0 ? 1 : -1);
            pass[u] = pass[u] & ok[u] & c; // unconditional: no branch to correlate with the load's
         }
      } else if (stage && (mp[p].op == LDB_F_LIKE || mp[p].op == LDB_F_NOT_LIKE) && mp[p].col.type == LDB_T_UTF8 && !mp[p].col.rowids && !mp[p].col.validity) {
         const CV col = pv.col();
         const int64_t* offs = gptr<int64_t>(col.p.offsets);
         const uint8_t* vals = gptr<uint8_t>(col.p.values);
         const uint32_t lane = threadIdx.x & 63;
         // the pattern goes to LDS too (behind the string area): read from the descriptor every
         // pattern byte is a memory load on the matcher's critical path — that, not the string
         // bytes, made the 20 M-name LIKE scan of Q9 take 4.9 ms
         const bool simple = mp[p].n_in > 0; // position-parallel matcher (d_like_simple_wave)
         if (!simple && lane < (uint32_t) mp[p].str_len) stage[LDS_STR_PAT + lane] = (uint8_t) mp[p].str[lane];
         // Software pipeline over the batch slots.  A slot is a dependent chain offsets → string bytes → LDS → match, and a wave that walks it
         // slot by slot pays two global round trips per 64 rows with nothing else in flight (Q13: 150 M comments at 2.2 TB/s, 0.27 of peak).
         // So: the offsets of ALL slots are loaded first (independent), and the string bytes of slot u + 1 travel into registers while slot
         // u is being matched out of LDS (up to LDS_STR_STAGE bytes per wave = 12 eight-byte words per lane).
         typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
         constexpr int NPRE = LDS_STR_STAGE / 512;
         int64_t ob[U], oe[U];
#pragma unroll
         for (int u = 0; u < U; u++) {
            const uint64_t row = rows[u] < n_rows ? rows[u] : n_rows; // offsets has n_rows + 1 entries
            ob[u] = offs[row];
            oe[u] = offs[row < n_rows ? row + 1 : n_rows];
         }
         uint64_t pre[NPRE];
#pragma unroll
         for (int t = 0; t < NPRE; t++) pre[t] = 0;
         auto fetch = [&](int u) __attribute__((always_inline)) {
            if (__ballot(pass[u]) == 0) return; // wave-uniform
            const int64_t begin = __shfl((long long) ob[u], 0), end = __shfl((long long) oe[u], 63);
            const uint64_t span = (uint64_t) (end - begin);
            if (span > LDS_STR_STAGE) return;
#pragma unroll
            for (int t = 0; t < NPRE; t++) {
               const uint64_t k = (uint64_t) lane * 8 + (uint64_t) t * 512;
               if (k < span) pre[t] = *(const LDB_GLOBAL u64_unaligned*) (vals + begin + k);
            }
         };
         fetch(0);
#pragma unroll
         for (int u = 0; u < U; u++) {
            if (__ballot(pass[u]) == 0) { // wave-uniform
               if (u + 1 < U) fetch(u + 1);
               continue;
            }
            const int64_t o = ob[u], o1 = oe[u];
            const int64_t begin = __shfl((long long) o, 0), end = __shfl((long long) o1, 63);
            const uint64_t span = (uint64_t) (end - begin);
            if (span <= LDS_STR_STAGE) {
#pragma unroll
               for (int t = 0; t < NPRE; t++) {
                  const uint64_t k = (uint64_t) lane * 8 + (uint64_t) t * 512;
                  if (k < span) *(uint64_t*) (stage + k) = pre[t];
               }
               if (u + 1 < U) fetch(u + 1); // the next slot's bytes are on their way while this one is matched
               __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
               __builtin_amdgcn_wave_barrier();
               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
               if (simple) {
                  const bool hit = d_like_simple_wave(mp[p], stage, (uint32_t) span, (uint32_t) (o - begin), (uint32_t) (o1 - o), pass[u]);
                  if (pass[u]) pass[u] = hit == (mp[p].op == LDB_F_LIKE);
                  continue;
               }
               if (pass[u]) {
                  d_bytes_lds s(stage + (o - begin)), pat(stage + LDS_STR_PAT);
                  pass[u] = d_like_on(s, (uint32_t) (o1 - o), pat, (uint32_t) mp[p].str_len) == (mp[p].op == LDB_F_LIKE);
               }
               __builtin_amdgcn_wave_barrier(); // the next batch slot overwrites the stage
            } else {
               if (u + 1 < U) fetch(u + 1);
               if (pass[u]) pass[u] = d_eval_pred(pv, rows[u]);
            }
         }
      } else if (d_pred_is_colcol_dense(mp[p])) {
         // column vs column, both dense narrow integers (l_commitdate < l_receiptdate): the same
         // branch-free scheme with two loads per row
         const CV a = pv.col(), b = pv.rhs();
#pragma unroll
         for (int u = 0; u < U; u++) {
            const uint32_t r = pass[u] ? (uint32_t) rows[u] : 0u;
            const int64_t x = d_load_i64(a, r), y = d_load_i64(b, r);
            pass[u] = pass[u] & d_cmp_vals<int64_t>(mp[p].op, x, y);
         }
      } else {
#pragma unroll
         for (int u = 0; u < U; u++)
            if (pass[u]) pass[u] = d_eval_pred(pv, rows[u]);
      }
   }
}

// db.hash of one key part folded into `total` (HashLowering::hashImpl, LowerToStd.cpp:1073-1132).
// Caller has checked validity (NULL parts are skipped).
__device__ __forceinline__ uint64_t d_hash_part(CV c, uint32_t row, uint64_t total) {
   switch (c.m.type) {
      case LDB_T_UTF8: {
         uint32_t len;
         const uint8_t* p = d_load_str(c, row, &len);
         return d_hash_combine(d_hash_varlen(p, len), total);
      }
      case LDB_T_FLOAT64: return d_hash_combine(d_hash64(gptr<uint64_t>(c.p.values)[row]), total);
      case LDB_T_FLOAT32: return d_hash_combine(d_hash64((uint64_t) (int64_t) gptr<int32_t>(c.p.values)[row]), total);
      case LDB_T_DATE32: // hashed in ns (LowerToStd.cpp:133-139)
         return d_hash_combine(d_hash64((uint64_t) ((int64_t) gptr<int32_t>(c.p.values)[row] * 86400000000000LL)), total);
      case LDB_T_BOOL8: return d_hash_combine(d_hash64(gptr<uint8_t>(c.p.values)[row] ? ~0ull : 0ull), total);
      case LDB_T_DECIMAL128:
         if (c.m.precision >= 19) { // two pieces: high then low (LowerToStd.cpp:1079-1090)
            i128 v = d_load_i128(c, row);
            uint64_t h1 = d_hash_combine(d_hash64((uint64_t) (v >> 64)), total);
            return d_hash_combine(d_hash64((uint64_t) v), h1);
         }
         [[fallthrough]];
      default: return d_hash_combine(d_hash64((uint64_t) d_load_i64(c, row)), total);
   }
}

// equality of one key part between two physical rows of (possibly different) columns
__device__ __forceinline__ bool d_key_part_equal(CV ca, uint32_t ra, CV cb, uint32_t rb) {
   if (ca.m.type == LDB_T_UTF8) {
      uint32_t la, lb;
      const uint8_t* a = d_load_str(ca, ra, &la);
      const uint8_t* b = d_load_str(cb, rb, &lb);
      if (la != lb) return false;
      for (uint32_t i = 0; i < la; i++)
         if (a[i] != b[i]) return false;
      return true;
   }
   if (d_is_flt(ca)) return d_load_f64(ca, ra) == d_load_f64(cb, rb);
   if (d_is_wide(ca) || d_is_wide(cb)) return d_load_i128(ca, ra) == d_load_i128(cb, rb);
   return d_load_i64(ca, ra) == d_load_i64(cb, rb);
}

// ---- 128-bit helpers (no libcalls on the device: no __divti3)
__device__ inline u128 d_udiv128(u128 n, u128 d) {
   if (d == 0) return 0;
   if (n < d) return 0;
   // shift-subtract long division; used only in per-group finalisation / clamped-scale terms
   u128 q = 0, r = 0;
   for (int i = 127; i >= 0; i--) {
      r = (r << 1) | ((n >> i) & 1);
      if (r >= d) {
         r -= d;
         q |= ((u128) 1 << i);
      }
   }
   return q;
}
// arith.divsi: truncating signed division
__device__ inline i128 d_sdiv128(i128 a, i128 b) {
   bool neg = (a < 0) != (b < 0);
   u128 ua = a < 0 ? (u128) 0 - (u128) a : (u128) a;
   u128 ub = b < 0 ? (u128) 0 - (u128) b : (u128) b;
   u128 q = d_udiv128(ua, ub);
   return neg ? (i128) ((u128) 0 - q) : (i128) q;
}
__device__ inline i128 d_pow10(int k) {
   i128 r = 1;
   for (int i = 0; i < k; i++) r *= 10;
   return r;
}

// wave-level helpers
__device__ __forceinline__ uint32_t d_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t d_rank_in(uint64_t mask) {
   return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}
