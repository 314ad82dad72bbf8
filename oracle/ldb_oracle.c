/*
 * ldb_oracle.c — CPU restatement of LingoDB's sub-operator hot path (see ldb_oracle.h).
 * TEST INFRASTRUCTURE ONLY — never linked into or called from the product path.
 *
 * Plain C11 + pthreads.  All arithmetic on decimals is __int128, wrapping, exactly as the
 * LLVM code the reference JIT-generates (arith.muli/addi/divsi on i64/i128).
 */
#define _GNU_SOURCE
#include "ldb_oracle.h"
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

/* ====================================================================== scalar spec (a5) */

/* util.hash64: m = x * 0x9E3779B97F4A7C55; m ^ bswap64(m)
 * (Hash64Lowering, src/compiler/Conversion/UtilToLLVM/LowerToLLVM.cpp:493-503; runtime twin
 *  dbHash64, src/runtime/Hash.cpp:25-28) */
uint64_t ora_hash64(int64_t v) {
   uint64_t m = 11400714819323198549ull * (uint64_t) v;
   return m ^ __builtin_bswap64(m);
}
/* util.hash_combine(h1=new, h2=total) = h1 ^ bswap64(h2)
 * (HashCombineLowering, LowerToLLVM.cpp:505-512; dbHashFoldPiece, Hash.cpp:30-32) */
uint64_t ora_hash_combine(uint64_t h_new, uint64_t total) {
   return h_new ^ __builtin_bswap64(total);
}

/* XXH64 (Yann Collet's published algorithm, the function behind llvm::xxHash64 of LLVM 20.1
 * that hashVarLenData calls with seed 0, src/runtime/Hash.cpp:13-16).  The dependency is not
 * in /root/reference; restated from the public specification (xxhash_spec.md, XXH64). */
#define XP1 11400714785074694791ULL
#define XP2 14029467366897019727ULL
#define XP3 1609587929392839161ULL
#define XP4 9650029242287828579ULL
#define XP5 2870177450012600261ULL
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) {
   uint64_t v;
   memcpy(&v, p, 8);
   return v;
}
static inline uint32_t rd32(const uint8_t* p) {
   uint32_t v;
   memcpy(&v, p, 4);
   return v;
}
static inline uint64_t xround(uint64_t acc, uint64_t in) {
   acc += in * XP2;
   acc = rotl64(acc, 31);
   return acc * XP1;
}
static inline uint64_t xmerge(uint64_t acc, uint64_t val) {
   val = xround(0, val);
   acc ^= val;
   return acc * XP1 + XP4;
}
uint64_t ora_xxh64(const void* data, uint64_t len, uint64_t seed) {
   const uint8_t* p = (const uint8_t*) data;
   const uint8_t* end = p + len;
   uint64_t h;
   if (len >= 32) {
      const uint8_t* limit = end - 32;
      uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
      do {
         v1 = xround(v1, rd64(p));
         v2 = xround(v2, rd64(p + 8));
         v3 = xround(v3, rd64(p + 16));
         v4 = xround(v4, rd64(p + 24));
         p += 32;
      } while (p <= limit);
      h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
      h = xmerge(h, v1);
      h = xmerge(h, v2);
      h = xmerge(h, v3);
      h = xmerge(h, v4);
   } else {
      h = seed + XP5;
   }
   h += len;
   while (p + 8 <= end) {
      h ^= xround(0, rd64(p));
      h = rotl64(h, 27) * XP1 + XP4;
      p += 8;
   }
   if (p + 4 <= end) {
      h ^= (uint64_t) rd32(p) * XP1;
      h = rotl64(h, 23) * XP2 + XP3;
      p += 4;
   }
   while (p < end) {
      h ^= (*p) * XP5;
      h = rotl64(h, 11) * XP1;
      p++;
   }
   h ^= h >> 33;
   h *= XP2;
   h ^= h >> 29;
   h *= XP3;
   h ^= h >> 32;
   return h;
}

/* 16-byte VarLen32 image of a short (<= 12 byte) string: len | bytes zero padded
 * (VarLen32 ctor, include/lingodb/runtime/helpers.h:194-209).  For len > 12 only len and the
 * 4-byte prefix are defined here (the pointer half is address dependent and never hashed). */
void ora_varlen32_image(const uint8_t* p, uint32_t len, uint8_t out[16]) {
   memset(out, 0, 16);
   memcpy(out, &len, 4);
   if (len <= 12) {
      memcpy(out + 4, p, len);
   } else {
      memcpy(out + 4, p, 4);
   }
}
/* VarLenTryCheapHash + HashVarLen (LowerToLLVM.cpp:372-391; dbHashVarLen32, Hash.cpp:45-56) */
uint64_t ora_hash_varlen(const uint8_t* p, uint32_t len) {
   if (len > 12) return ora_xxh64(p, len, 0);
   uint8_t img[16];
   ora_varlen32_image(p, len, img);
   uint64_t first64 = rd64(img), last64 = rd64(img + 8);
   return ora_hash64((int64_t) first64) ^ __builtin_bswap64(ora_hash64((int64_t) last64));
}
/* i128 key part: hash(high64) folded first, then hash(low64) (HashLowering,
 * src/compiler/Conversion/DBToStd/LowerToStd.cpp:1079-1090; Hash.cpp:126-137) */
uint64_t ora_hash_i128(uint64_t lo, int64_t hi, int first, uint64_t total) {
   uint64_t h1 = ora_hash64(hi);
   total = first ? h1 : ora_hash_combine(h1, total);
   return ora_hash_combine(ora_hash64((int64_t) lo), total);
}

/* bloomMasks: 16-bit words with 4 bits set, indexed by the top 11 hash bits
 * (src/runtime/helpers.cpp: 1820 distinct patterns in ascending order + 228 repeats;
 * helpers.h:326-346).  The table only decides which chains are skipped early — false
 * positives are re-checked on the keys — so its exact content never changes a result.  The
 * restatement generates the 1820 ascending patterns and repeats from the start. */
static uint16_t g_bloom[2048];
static pthread_once_t g_bloom_once = PTHREAD_ONCE_INIT;
static void bloom_init(void) {
   int n = 0;
   for (uint32_t x = 0; x < 65536 && n < 1820; x++)
      if (__builtin_popcount(x) == 4) g_bloom[n++] = (uint16_t) x;
   for (int i = 1820; i < 2048; i++) g_bloom[i] = g_bloom[(i - 1820) * 7 % 1820];
}
uint16_t ora_bloom_mask(uint32_t idx) {
   pthread_once(&g_bloom_once, bloom_init);
   return g_bloom[idx & 2047];
}

/* ====================================================================== column access (a4) */

static inline int col_valid(const ora_col* c, int64_t row) {
   return !c->validity || ((c->validity[row >> 3] >> (row & 7)) & 1);
}
static inline int is_string(const ora_col* c) { return c->type == LDB_T_UTF8; }
static inline int is_float(const ora_col* c) { return c->type == LDB_T_FLOAT64 || c->type == LDB_T_FLOAT32; }

/* LoadArrowOpLowering (LowerToStd.cpp:111-209): fixed-width load, sign extension; decimal128
 * with p < 19 is truncated to i64 (:128-132). */
static inline i128 load_int(const ora_col* c, int64_t row) {
   switch (c->type) {
      case LDB_T_INT8: return ((const int8_t*) c->values)[row];
      case LDB_T_BOOL8: return ((const uint8_t*) c->values)[row] ? 1 : 0;
      case LDB_T_INT16: return ((const int16_t*) c->values)[row];
      case LDB_T_INT32:
      case LDB_T_DATE32:
      case LDB_T_CHAR4: return ((const int32_t*) c->values)[row];
      case LDB_T_INT64: return ((const int64_t*) c->values)[row];
      case LDB_T_DECIMAL128: {
         if (c->width == 8) return ((const int64_t*) c->values)[row];
         i128 v;
         memcpy(&v, (const uint8_t*) c->values + row * 16, 16);
         if (c->precision < 19) v = (int64_t) v;
         return v;
      }
      default: return 0;
   }
}
static inline double load_f64(const ora_col* c, int64_t row) {
   if (c->type == LDB_T_FLOAT64) return ((const double*) c->values)[row];
   if (c->type == LDB_T_FLOAT32) return ((const float*) c->values)[row];
   return (double) load_int(c, row);
}
static inline const uint8_t* load_str(const ora_col* c, int64_t row, uint32_t* len) {
   int64_t b = c->offsets[row], e = c->offsets[row + 1];
   *len = (uint32_t) (e - b);
   return (const uint8_t*) c->values + b;
}
static inline const ora_col* rel_col(const ora_rel* r, ldb_colref ref) { return &r->tables[ref.side]->cols[ref.col]; }
/* physical row of logical row i on `side`; LDB_NULL_ROW for outer-join padding */
static inline int64_t rel_row(const ora_rel* r, int32_t side, int64_t i) {
   const uint32_t* ids = r->rowids[side];
   if (!ids) return i;
   return ids[i] == LDB_NULL_ROW ? -1 : (int64_t) ids[i];
}
static inline int ref_valid(const ora_rel* r, ldb_colref ref, int64_t i, int64_t* prow) {
   int64_t row = rel_row(r, ref.side, i);
   *prow = row;
   if (row < 0) return 0;
   return col_valid(rel_col(r, ref), row);
}

/* std::string_view three-way compare = unsigned bytewise, then length (StringRuntime.cpp:242-256) */
static inline int str_cmp(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
   uint32_t m = la < lb ? la : lb;
   int c = m ? memcmp(a, b, m) : 0;
   if (c) return c < 0 ? -1 : 1;
   return la < lb ? -1 : (la > lb ? 1 : 0);
}
static inline int cmp_apply(int op, int c3) {
   switch (op) {
      case LDB_F_EQ: return c3 == 0;
      case LDB_F_NEQ: return c3 != 0;
      case LDB_F_LT: return c3 < 0;
      case LDB_F_LTE: return c3 <= 0;
      case LDB_F_GT: return c3 > 0;
      case LDB_F_GTE: return c3 >= 0;
      default: return 0;
   }
}
static inline i128 make_i128(uint64_t lo, int64_t hi) { return (i128) (((u128) (uint64_t) hi << 64) | lo); }

/* One predicate on one logical row.  Pushed-down filter semantics: Filter impls of
 * src/runtime/storage/Restrictions.cpp:67-321 (SimpleTypeFilter compares in the column's
 * native type, decimals as __int128, strings as string_view, IN = membership); residual
 * column-vs-column compares follow db.cmp on the loaded values.  A NULL operand fails every
 * comparison (the reference emits a NOTNULL filter first, Pushdown.cpp:266-408). */
/* SQL LIKE as the reference evaluates it (StringRuntime::like → iterativeLike,
 * src/runtime/StringRuntime.cpp:28-93, 134-136; escape character '\\').  Restated from its
 * behaviour, quirks included:
 *   - a "character" is a UTF-8 lead byte plus its continuation bytes (nextChar :18-26);
 *   - two characters are equal when their LEAD bytes are equal (`*p == *s`, :84): continuation
 *     bytes are skipped, never compared;
 *   - '_' consumes one character, '%' any number; after a run of %/_ an escape character is
 *     stepped over and the character behind it is then matched like an unescaped one (:65-70 hand
 *     the recursion a pattern that starts behind the escape);
 *   - a pattern that ends in a lone escape never matches (:33-36, :67-69). */
static size_t like_char_len(const uint8_t* p, size_t left) { /* bytes of the character at p (left > 0) */
   size_t k = 1;
   while (k < left && (p[k] >> 6) == 2) k++;
   return k;
}
static int like_match(const uint8_t* s, size_t sl, const uint8_t* p, size_t pl) {
   while (pl > 0 && sl > 0) {
      if (*p == '\\') {
         size_t e = like_char_len(p, pl);
         p += e;
         pl -= e;
         if (pl == 0 || *p != *s) return 0;
         size_t a = like_char_len(s, sl), b = like_char_len(p, pl);
         s += a, sl -= a, p += b, pl -= b;
      } else if (*p == '%') {
         p++, pl--;
         while (pl > 0 && (*p == '%' || *p == '_')) { /* collapse the wildcard run */
            if (*p == '_') {
               if (sl == 0) return 0;
               size_t a = like_char_len(s, sl);
               s += a, sl -= a;
            }
            p++, pl--;
         }
         if (pl == 0) return 1;
         if (*p == '\\') {
            size_t e = like_char_len(p, pl);
            p += e, pl -= e;
            if (pl == 0) return 0;
         }
         while (sl > 0) { /* try the rest of the pattern at every remaining position */
            if (like_match(s, sl, p, pl)) return 1;
            size_t a = like_char_len(s, sl);
            s += a, sl -= a;
         }
         return 0;
      } else if (*p == '_' || *p == *s) {
         size_t a = like_char_len(s, sl), b = like_char_len(p, pl);
         s += a, sl -= a, p += b, pl -= b;
      } else {
         return 0;
      }
   }
   while (pl > 0 && *p == '%') p++, pl--;
   return sl == 0 && pl == 0;
}
int32_t ora_like(const uint8_t* s, int64_t sl, const uint8_t* p, int64_t pl) { return like_match(s, (size_t) sl, p, (size_t) pl); }

/* substring(str from `from` for `len`): StringRuntime::substr (src/runtime/StringRuntime.cpp:292-319) with
 * charIndexToByteIndex (:102-135): positions count UTF-8 characters from 1, positions before the string
 * count towards the length, from / to beyond the end are truncated to it.  Writes the byte range. */
static size_t char_to_byte(const uint8_t* s, size_t byte_len, size_t char_index, size_t known_byte, size_t known_char) {
   for (; known_byte < byte_len; known_byte++) {
      if ((s[known_byte] >> 6) != 2) { /* not a continuation byte */
         if (known_char == char_index) return known_byte;
         known_char++;
      }
   }
   return byte_len;
}
void ora_substr(const uint8_t* s, int64_t sl, int64_t from, int64_t len, int64_t* out_begin, int64_t* out_end) {
   int64_t legal_len = len > 0 ? len : 0;
   size_t legal_from = (size_t) (from > 1 ? from : 1);
   int64_t to_raw = from + legal_len;
   size_t legal_to = to_raw > (int64_t) legal_from ? (size_t) to_raw : legal_from;
   legal_from--;
   legal_to--;
   size_t b0 = char_to_byte(s, (size_t) sl, legal_from, 0, 0);
   size_t b1 = char_to_byte(s, (size_t) sl, legal_to, b0, legal_from);
   *out_begin = (int64_t) b0;
   *out_end = (int64_t) b1;
}

/* extract(year from date) on a date32 (days since 1970-01-01): DateRuntime::extractYear
 * (src/runtime/DateRuntime.cpp:99-101) = civil year of the day (proleptic Gregorian, UTC). */
int64_t ora_extract_year(int64_t days) {
   int64_t z = days + 719468; /* days since 0000-03-01 */
   int64_t era = (z >= 0 ? z : z - 146096) / 146097;
   int64_t doe = z - era * 146097; /* [0, 146096] */
   int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365; /* [0, 399] */
   int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100); /* [0, 365], March-based */
   int64_t mp = (5 * doy + 2) / 153; /* March = 0 */
   return yoe + era * 400 + (mp >= 10 ? 1 : 0);
}

/* literal * num / den on decimals (Q14's `100.00 * sum(..) / sum(..)`): DecimalMulOpLowering
 * (LowerToStd.cpp:653-677: sign-extend to the result width, multiply, sdiv by 10^(sL+sR-sRes) when
 * the result scale was clamped) followed by DecimalOpScaledLowering (:631-651:
 * (left * 10^(sRes + sR - sL)) sdiv right).  128-bit wrapping arithmetic (result types with
 * p >= 19); values as {lo, hi} words.  Returns 0 (no value) when den is 0: undefined there. */
static i128 pow10_i128(int k) {
   i128 r = 1;
   while (k-- > 0) r *= 10;
   return r;
}
int32_t ora_decimal_muldiv(const int64_t num[2], const int64_t mul[2], int32_t mul_div_pow10, int32_t pow10, const int64_t den[2], int64_t out[2]) {
   i128 n = (i128) (((u128) (uint64_t) num[1] << 64) | (uint64_t) num[0]);
   i128 m = (i128) (((u128) (uint64_t) mul[1] << 64) | (uint64_t) mul[0]);
   i128 d = (i128) (((u128) (uint64_t) den[1] << 64) | (uint64_t) den[0]);
   if (d == 0) return 0;
   i128 prod = (i128) ((u128) n * (u128) m);
   if (mul_div_pow10 > 0) prod = prod / pow10_i128(mul_div_pow10);
   i128 q = (i128) ((u128) prod * (u128) pow10_i128(pow10)) / d;
   out[0] = (int64_t) (uint64_t) q;
   out[1] = (int64_t) (q >> 64);
   return 1;
}

static int eval_pred(const ora_rel* r, const ldb_filter_desc* p, int64_t i) {
   const ora_col* c = rel_col(r, p->col);
   int64_t row;
   int valid = ref_valid(r, p->col, i, &row);
   if (p->op == LDB_F_NOTNULL) return valid;
   if (!valid) return 0;
   if (p->rhs_kind == LDB_RHS_COLUMN) {
      const ora_col* c2 = rel_col(r, p->rhs_col);
      int64_t row2;
      if (!ref_valid(r, p->rhs_col, i, &row2)) return 0;
      if (is_string(c)) {
         uint32_t la, lb;
         const uint8_t* a = load_str(c, row, &la);
         const uint8_t* b = load_str(c2, row2, &lb);
         return cmp_apply(p->op, str_cmp(a, la, b, lb));
      }
      if (is_float(c) || is_float(c2)) {
         double a = load_f64(c, row), b = load_f64(c2, row2);
         return cmp_apply(p->op, a < b ? -1 : (a > b ? 1 : 0));
      }
      i128 a = load_int(c, row), b = load_int(c2, row2);
      return cmp_apply(p->op, a < b ? -1 : (a > b ? 1 : 0));
   }
   if (is_string(c)) {
      uint32_t la;
      const uint8_t* a = load_str(c, row, &la);
      if (p->op == LDB_F_IN) {
         for (int k = 0; k < p->n_in; k++)
            if (str_cmp(a, la, (const uint8_t*) p->in_strs[k], (uint32_t) p->in_str_lens[k]) == 0) return 1;
         return 0;
      }
      if (p->op == LDB_F_LIKE || p->op == LDB_F_NOT_LIKE)
         return like_match(a, la, (const uint8_t*) p->str, (size_t) p->str_len) == (p->op == LDB_F_LIKE);
      return cmp_apply(p->op, str_cmp(a, la, (const uint8_t*) p->str, (uint32_t) p->str_len));
   }
   if (is_float(c)) {
      double a = load_f64(c, row);
      if (p->op == LDB_F_IN) {
         for (int k = 0; k < p->n_in; k++) {
            double b;
            memcpy(&b, &p->in_values[2 * k], 8);
            if (a == b) return 1;
         }
         return 0;
      }
      double b = p->value_f64;
      return cmp_apply(p->op, a < b ? -1 : (a > b ? 1 : 0));
   }
   i128 a = load_int(c, row);
   if (p->op == LDB_F_IN) {
      for (int k = 0; k < p->n_in; k++)
         if (a == make_i128((uint64_t) p->in_values[2 * k], p->in_values[2 * k + 1])) return 1;
      return 0;
   }
   i128 b = make_i128(p->value_lo, p->value_hi);
   return cmp_apply(p->op, a < b ? -1 : (a > b ? 1 : 0));
}

/* ====================================================================== worker pool */
/* Stand-in for the fiber scheduler (include/lingodb/scheduler/Scheduler.h:29-42): N threads
 * pull work units from an atomic cursor (the reference hands out units per worker and steals,
 * LingoDBTable.cpp:409-454; the hand-out order never affects results). */
typedef void (*unit_fn)(void* arg, int worker, int64_t unit);
typedef struct {
   unit_fn fn;
   void* arg;
   _Atomic int64_t next;
   int64_t n_units;
} pool_job;
typedef struct {
   pool_job* job;
   int worker;
} pool_arg;
static void* pool_main(void* a) {
   pool_arg* pa = (pool_arg*) a;
   for (;;) {
      int64_t u = atomic_fetch_add(&pa->job->next, 1);
      if (u >= pa->job->n_units) break;
      pa->job->fn(pa->job->arg, pa->worker, u);
   }
   return NULL;
}
static void run_units(int threads, int64_t n_units, unit_fn fn, void* arg) {
   pool_job job = {fn, arg, 0, n_units};
   if (threads <= 1 || n_units <= 1) {
      pool_arg pa = {&job, 0};
      pool_main(&pa);
      return;
   }
   if (threads > 256) threads = 256;
   if ((int64_t) threads > n_units) threads = (int) n_units; /* never more workers than units (the merge has 64 partitions) */
   pthread_t th[256];
   pool_arg pas[256];
   for (int t = 0; t < threads; t++) {
      pas[t].job = &job;
      pas[t].worker = t;
      pthread_create(&th[t], NULL, pool_main, &pas[t]);
   }
   for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}
int32_t ora_num_cores(void) { return (int32_t) sysconf(_SC_NPROCESSORS_ONLN); }

#define MORSEL 20000 /* splitSize, src/runtime/storage/LingoDBTable.cpp:364 */

/* ====================================================================== scan + filter (a2, a3) */
/* ScanBatchesTask::unitRun (LingoDBTable.cpp:382-407) + Restrictions::applyFilters
 * (Restrictions.cpp:365-390): per unit, each filter compacts a uint16 selection vector
 * (ping-pong between two buffers), early exit when empty.  Returns the selection length and
 * leaves the final vector in *sel. */
static int32_t morsel_filter(const ora_rel* r, const ldb_filter_desc* preds, int32_t n_preds, int64_t base, int32_t len,
                             uint16_t* sv1, uint16_t* sv2, uint16_t** sel) {
   uint16_t* cur = sv1;
   uint16_t* nxt = sv2;
   int32_t n = len;
   for (int32_t i = 0; i < len; i++) cur[i] = (uint16_t) i; /* defaultSelectionVector */
   for (int32_t f = 0; f < n_preds; f++) {
      uint16_t* w = nxt;
      for (int32_t i = 0; i < n; i++) {
         uint16_t idx = cur[i];
         *w = idx;
         w += eval_pred(r, &preds[f], base + idx); /* branch-free compaction (:176-189) */
      }
      n = (int32_t) (w - nxt);
      uint16_t* t = cur;
      cur = nxt;
      nxt = t;
      if (n == 0) break;
   }
   *sel = cur;
   return n;
}

typedef struct {
   const ora_rel* in;
   const ldb_filter_desc* preds;
   int32_t n_preds;
   uint32_t* out; /* staged at the unit's own offset, compacted afterwards */
   int64_t* unit_counts;
} scan_job;
static void scan_unit(void* a, int worker, int64_t u) {
   (void) worker;
   scan_job* j = (scan_job*) a;
   int64_t base = u * MORSEL;
   int32_t len = (int32_t) (j->in->n_rows - base < MORSEL ? j->in->n_rows - base : MORSEL);
   uint16_t sv1[MORSEL], sv2[MORSEL];
   uint16_t* sel;
   int32_t n = morsel_filter(j->in, j->preds, j->n_preds, base, len, sv1, sv2, &sel);
   for (int32_t i = 0; i < n; i++) j->out[base + i] = (uint32_t) (base + sel[i]);
   j->unit_counts[u] = n;
}
/* Output = LOGICAL row numbers of `in` that pass, ascending. */
int64_t ora_scan_filter(const ora_rel* in, const ldb_filter_desc* preds, int32_t n_preds, uint32_t* out_rows, int32_t threads) {
   int64_t n_units = (in->n_rows + MORSEL - 1) / MORSEL;
   if (n_units == 0) return 0;
   int64_t* counts = (int64_t*) calloc((size_t) n_units, sizeof(int64_t));
   uint32_t* staged = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) (in->n_rows ? in->n_rows : 1));
   scan_job j = {in, preds, n_preds, staged, counts};
   run_units(threads, n_units, scan_unit, &j);
   int64_t total = 0;
   for (int64_t u = 0; u < n_units; u++) {
      if (out_rows) memcpy(out_rows + total, staged + u * MORSEL, sizeof(uint32_t) * (size_t) counts[u]);
      total += counts[u];
   }
   free(staged);
   free(counts);
   return total;
}

/* ====================================================================== db.hash over key columns (a5) */
/* HashLowering::hashImpl per key part (LowerToStd.cpp:1073-1132): integers sign-extended to
 * 64 bit; date32 in the runtime unit ns (days * 86 400 000 000 000, LowerToStd.cpp:133-139);
 * decimal p>=19 as two pieces; floats bit-cast; strings via VarLenTryCheapHash; NULL parts
 * leave the running hash unchanged (0 if first). */
static inline uint64_t hash_part(const ora_col* c, int64_t row, uint64_t total) {
   switch (c->type) {
      case LDB_T_UTF8: {
         uint32_t len;
         const uint8_t* p = load_str(c, row, &len);
         return ora_hash_combine(ora_hash_varlen(p, len), total);
      }
      case LDB_T_FLOAT64: {
         int64_t bits;
         memcpy(&bits, (const uint8_t*) c->values + row * 8, 8);
         return ora_hash_combine(ora_hash64(bits), total);
      }
      case LDB_T_FLOAT32: {
         int32_t bits;
         memcpy(&bits, (const uint8_t*) c->values + row * 4, 4);
         return ora_hash_combine(ora_hash64((int64_t) bits), total);
      }
      case LDB_T_DATE32: {
         int64_t ns = (int64_t) ((const int32_t*) c->values)[row] * 86400000000000LL;
         return ora_hash_combine(ora_hash64(ns), total);
      }
      case LDB_T_BOOL8: { /* i1 true sign-extends to -1 (arith.index_cast, :1073-1076) */
         int64_t v = ((const uint8_t*) c->values)[row] ? -1 : 0;
         return ora_hash_combine(ora_hash64(v), total);
      }
      case LDB_T_DECIMAL128:
         if (c->precision >= 19) {
            i128 v = load_int(c, row);
            uint64_t h1 = ora_hash_combine(ora_hash64((int64_t) (v >> 64)), total);
            return ora_hash_combine(ora_hash64((int64_t) (uint64_t) v), h1);
         }
         /* fallthrough: i64 */
      default: return ora_hash_combine(ora_hash64((int64_t) load_int(c, row)), total);
   }
}
static inline uint64_t hash_row(const ora_rel* r, const ldb_colref* keys, int32_t n_keys, int64_t i) {
   uint64_t total = 0; /* combine(h, 0) == h, so "no hash yet" and 0 coincide */
   for (int32_t k = 0; k < n_keys; k++) {
      int64_t row;
      if (!ref_valid(r, keys[k], i, &row)) continue;
      total = hash_part(rel_col(r, keys[k]), row, total);
   }
   return total;
}
void ora_hash_keys(const ora_rel* in, const ldb_colref* keys, int32_t n_keys, uint64_t* out) {
   for (int64_t i = 0; i < in->n_rows; i++) out[i] = hash_row(in, keys, n_keys, i);
}
void ora_partition_ids(const ora_rel* in, const ldb_colref* keys, int32_t n_keys, int32_t nparts, int32_t* out) {
   for (int64_t i = 0; i < in->n_rows; i++) out[i] = (int32_t) ((hash_row(in, keys, n_keys, i) >> 16) % (uint64_t) nparts);
}

/* key equality: db.cmp eq per key pair (createEqFn, RelAlgToSubOp.cpp:1035-1066);
 * nulls_equal selects `isa` semantics (group-by: NULL = NULL) vs join semantics. */
static int keys_equal(const ora_rel* ra, const ldb_colref* ka, int64_t ia, const ora_rel* rb, const ldb_colref* kb, int64_t ib,
                      int32_t n_keys, int nulls_equal) {
   for (int32_t k = 0; k < n_keys; k++) {
      int64_t rowa, rowb;
      int va = ref_valid(ra, ka[k], ia, &rowa), vb = ref_valid(rb, kb[k], ib, &rowb);
      if (!va || !vb) {
         if (nulls_equal && !va && !vb) continue;
         return 0;
      }
      const ora_col* ca = rel_col(ra, ka[k]);
      const ora_col* cb = rel_col(rb, kb[k]);
      if (is_string(ca)) {
         uint32_t la, lb;
         const uint8_t* a = load_str(ca, rowa, &la);
         const uint8_t* b = load_str(cb, rowb, &lb);
         if (la != lb || (la && memcmp(a, b, la))) return 0;
      } else if (is_float(ca)) {
         if (load_f64(ca, rowa) != load_f64(cb, rowb)) return 0;
      } else if (load_int(ca, rowa) != load_int(cb, rowb)) {
         return 0;
      }
   }
   return 1;
}

/* ====================================================================== expressions (a16) */
static const i128 POW10[39] = {
   (i128) 1ULL, (i128) 10ULL, (i128) 100ULL, (i128) 1000ULL, (i128) 10000ULL, (i128) 100000ULL, (i128) 1000000ULL,
   (i128) 10000000ULL, (i128) 100000000ULL, (i128) 1000000000ULL, (i128) 10000000000ULL, (i128) 100000000000ULL,
   (i128) 1000000000000ULL, (i128) 10000000000000ULL, (i128) 100000000000000ULL, (i128) 1000000000000000ULL,
   (i128) 10000000000000000ULL, (i128) 100000000000000000ULL, (i128) 1000000000000000000ULL,
   (i128) 10000000000000000000ULL,
   (i128) 10000000000000000000ULL * 10, (i128) 10000000000000000000ULL * 100, (i128) 10000000000000000000ULL * 1000,
   (i128) 10000000000000000000ULL * 10000, (i128) 10000000000000000000ULL * 100000,
   (i128) 10000000000000000000ULL * 1000000, (i128) 10000000000000000000ULL * 10000000,
   (i128) 10000000000000000000ULL * 100000000, (i128) 10000000000000000000ULL * 1000000000,
   (i128) 10000000000000000000ULL * 10000000000ULL, (i128) 10000000000000000000ULL * 100000000000ULL,
   (i128) 10000000000000000000ULL * 1000000000000ULL, (i128) 10000000000000000000ULL * 10000000000000ULL,
   (i128) 10000000000000000000ULL * 100000000000000ULL, (i128) 10000000000000000000ULL * 1000000000000000ULL,
   (i128) 10000000000000000000ULL * 10000000000000000ULL, (i128) 10000000000000000000ULL * 100000000000000000ULL,
   (i128) 10000000000000000000ULL * 1000000000000000000ULL,
   (i128) 10000000000000000000ULL * 1000000000000000000ULL * 10};

static inline i128 wrap_mul(i128 a, i128 b) { return (i128) ((u128) a * (u128) b); }
static inline i128 wrap_add(i128 a, i128 b) { return (i128) ((u128) a + (u128) b); }

/* Sum-of-products over decimals/ints.  DecimalMulOpLowering = arith.muli (+ arith.divsi by
 * 10^(sL+sR-sRes) when the frontend clamped the scale), LowerToStd.cpp:653-677;
 * add/sub = arith.addi/subi after common-scale casts, :680-699.  *null_out is set when any
 * referenced column is NULL (NULL propagates through arithmetic). */
static i128 eval_expr_int(const ora_rel* r, const ldb_expr* e, int64_t i, int* null_out) {
   i128 total = 0;
   *null_out = 0;
   for (int32_t t = 0; t < e->n_terms; t++) {
      const ldb_term* tm = &e->t[t];
      i128 prod = 1;
      for (int32_t f = 0; f < tm->n_factors; f++) {
         const ldb_factor* fa = &tm->f[f];
         i128 v = fa->a;
         if (fa->has_col) {
            int64_t row;
            if (!ref_valid(r, fa->col, i, &row)) {
               *null_out = 1;
               return 0;
            }
            v = wrap_add(v, wrap_mul((i128) fa->b, load_int(rel_col(r, fa->col), row)));
         }
         prod = wrap_mul(prod, v);
      }
      if (tm->div_pow10 > 0) prod = prod / POW10[tm->div_pow10]; /* arith.divsi: truncating */
      total = tm->negate ? (i128) ((u128) total - (u128) prod) : wrap_add(total, prod);
   }
   return total;
}
static double eval_expr_f64(const ora_rel* r, const ldb_expr* e, int64_t i, int* null_out) {
   double total = 0;
   *null_out = 0;
   for (int32_t t = 0; t < e->n_terms; t++) {
      const ldb_term* tm = &e->t[t];
      double prod = 1;
      for (int32_t f = 0; f < tm->n_factors; f++) {
         const ldb_factor* fa = &tm->f[f];
         double v = (double) fa->a;
         if (fa->has_col) {
            int64_t row;
            if (!ref_valid(r, fa->col, i, &row)) {
               *null_out = 1;
               return 0;
            }
            v += (double) fa->b * load_f64(rel_col(r, fa->col), row);
         }
         prod *= v;
      }
      total = tm->negate ? total - prod : total + prod;
   }
   return total;
}
void ora_eval_expr(const ora_rel* in, const ldb_expr* e, int64_t* out_lohi) {
   for (int64_t i = 0; i < in->n_rows; i++) {
      int nul;
      i128 v = eval_expr_int(in, e, i, &nul);
      out_lohi[2 * i] = (int64_t) (uint64_t) v;
      out_lohi[2 * i + 1] = (int64_t) (v >> 64);
   }
}

/* ====================================================================== group-by (a9, a10, a14) */
/* Aggregate state per (group, aggregate); init/aggregate/combine rules of
 * RelAlgToSubOp.cpp:1786-2026: SUM/MIN/MAX start NULL and become valid on the first
 * non-NULL input; COUNT(x) skips NULL, COUNT(*) counts rows; ANY keeps the first value;
 * SUM accumulates in the argument type (i64 for decimal p<19 — `wide`=0 — else i128). */
typedef struct {
   i128 v;
   double f;
   int64_t cnt; /* AVG divisor / COUNT */
   uint8_t valid;
} agg_state;

typedef struct entry {
   struct entry* next;
   uint64_t hash;
   uint32_t rep; /* logical row holding the key values */
   agg_state st[];
} entry;

static void agg_init(agg_state* s, int32_t n) { memset(s, 0, sizeof(agg_state) * (size_t) n); }

static void agg_update(const ora_rel* r, const ldb_agg_spec* aggs, int32_t n_aggs, int64_t i, agg_state* st) {
   for (int32_t a = 0; a < n_aggs; a++) {
      const ldb_agg_spec* sp = &aggs[a];
      agg_state* s = &st[a];
      int pass = 1;
      for (int32_t p = 0; p < sp->n_preds && pass; p++) pass = eval_pred(r, &sp->preds[p], i);
      if (sp->fn == LDB_AGG_COUNT_STAR) {
         if (pass) s->cnt++;
         s->valid = 1;
         continue;
      }
      int nul = 0;
      i128 v = 0;
      double fv = 0;
      if (sp->arg.is_float)
         fv = eval_expr_f64(r, &sp->arg, i, &nul);
      else
         v = eval_expr_int(r, &sp->arg, i, &nul);
      if (sp->fn == LDB_AGG_COUNT) {
         s->valid = 1;
         if (pass && !nul) s->cnt++;
         continue;
      }
      if (!pass) { /* sum(case when p then x else 0 end): contributes 0, state becomes valid */
         if (sp->n_preds && (sp->fn == LDB_AGG_SUM)) s->valid = 1;
         continue;
      }
      if (nul) continue;
      switch (sp->fn) {
         case LDB_AGG_SUM:
         case LDB_AGG_AVG:
            if (sp->arg.is_float)
               s->f += fv;
            else if (sp->wide)
               s->v = wrap_add(s->v, v);
            else
               s->v = (int64_t) ((uint64_t) (int64_t) s->v + (uint64_t) (int64_t) v);
            if (sp->fn == LDB_AGG_AVG && sp->has_count_expr) { /* combining partial (sum, count) states */
               int cnul;
               s->cnt += (int64_t) eval_expr_int(r, &sp->count_expr, i, &cnul);
            } else {
               s->cnt++;
            }
            s->valid = 1;
            break;
         case LDB_AGG_MIN:
            if (sp->arg.is_float) {
               if (!s->valid || fv < s->f) s->f = fv;
            } else if (!s->valid || v < s->v)
               s->v = v;
            s->valid = 1;
            break;
         case LDB_AGG_MAX:
            if (sp->arg.is_float) {
               if (!s->valid || fv > s->f) s->f = fv;
            } else if (!s->valid || v > s->v)
               s->v = v;
            s->valid = 1;
            break;
         case LDB_AGG_ANY:
            if (!s->valid) {
               s->v = v;
               s->f = fv;
               s->valid = 1;
            }
            break;
         default: break;
      }
   }
}
/* combine(dst, src) of the merge step (MergePreAggrHashMap, SubOpToControlFlow.cpp:1861-1938) */
static void agg_combine(const ldb_agg_spec* aggs, int32_t n_aggs, agg_state* d, const agg_state* s) {
   for (int32_t a = 0; a < n_aggs; a++) {
      const ldb_agg_spec* sp = &aggs[a];
      if (sp->fn == LDB_AGG_COUNT || sp->fn == LDB_AGG_COUNT_STAR) {
         d[a].cnt += s[a].cnt;
         d[a].valid = 1;
         continue;
      }
      if (!s[a].valid) continue;
      if (!d[a].valid) {
         d[a] = s[a];
         continue;
      }
      switch (sp->fn) {
         case LDB_AGG_SUM:
         case LDB_AGG_AVG:
            if (sp->arg.is_float)
               d[a].f += s[a].f;
            else if (sp->wide)
               d[a].v = wrap_add(d[a].v, s[a].v);
            else
               d[a].v = (int64_t) ((uint64_t) (int64_t) d[a].v + (uint64_t) (int64_t) s[a].v);
            d[a].cnt += s[a].cnt;
            break;
         case LDB_AGG_MIN:
            if (sp->arg.is_float) {
               if (s[a].f < d[a].f) d[a].f = s[a].f;
            } else if (s[a].v < d[a].v)
               d[a].v = s[a].v;
            break;
         case LDB_AGG_MAX:
            if (sp->arg.is_float) {
               if (s[a].f > d[a].f) d[a].f = s[a].f;
            } else if (s[a].v > d[a].v)
               d[a].v = s[a].v;
            break;
         default: break; /* ANY keeps dst */
      }
   }
}

/* growable pointer vector = FlexibleBuffer of entries (Buffer.h:43-105) */
typedef struct {
   entry** p;
   int64_t n, cap;
} evec;
static void evec_push(evec* v, entry* e) {
   if (v->n == v->cap) {
      v->cap = v->cap ? v->cap + v->cap / 5 + 1 : 256; /* grows x1.2 like FlexibleBuffer */
      v->p = (entry**) realloc(v->p, sizeof(entry*) * (size_t) v->cap);
   }
   v->p[v->n++] = e;
}

#define FRAG_HT 1024 /* PreAggregationHashtableFragment::hashtableSize, PreAggregationHashtable.h:17 */
#define FRAG_OUT 64 /* numOutputs, :16 */
typedef struct {
   entry* ht[FRAG_HT];
   evec outputs[FRAG_OUT];
   /* arena for entries */
   uint8_t** blocks;
   int32_t n_blocks;
   size_t used, blk_size, entry_size;
} fragment;

static entry* frag_alloc(fragment* f) {
   if (!f->n_blocks || f->used + f->entry_size > f->blk_size) {
      f->blocks = (uint8_t**) realloc(f->blocks, sizeof(uint8_t*) * (size_t) (f->n_blocks + 1));
      f->blk_size = 1 << 20;
      if (f->blk_size < f->entry_size) f->blk_size = f->entry_size;
      f->blocks[f->n_blocks++] = (uint8_t*) malloc(f->blk_size);
      f->used = 0;
   }
   entry* e = (entry*) (f->blocks[f->n_blocks - 1] + f->used);
   f->used += f->entry_size;
   return e;
}

typedef struct {
   const ora_rel* in;
   const ldb_filter_desc* preds;
   int32_t n_preds;
   const ldb_colref* keys;
   int32_t n_keys;
   const ldb_agg_spec* aggs;
   int32_t n_aggs;
   fragment* frags; /* one per worker */
} gb_job;

/* The generated pipeline body for one morsel: scan loop over the selection vector
 * (ScanRefsTableLowering, SubOpToControlFlow.cpp:1178-1191), db.hash of the keys, fragment
 * lookup `ht[(hash >> 6) & 1023]`, hit = hash equal && keys equal → reduce in place; miss →
 * PreAggregationHashtableFragment::insert (PreAggregationHashtable.cpp:46-60: append to
 * outputs[hash & 63], OVERWRITE the cache slot) then init + reduce
 * (LookupPreAggrHtFragment, SubOpToControlFlow.cpp:3065-3157; ReduceOpLowering :3719-3768). */
static void gb_unit(void* a, int worker, int64_t u) {
   gb_job* j = (gb_job*) a;
   fragment* fr = &j->frags[worker];
   int64_t base = u * MORSEL;
   int32_t len = (int32_t) (j->in->n_rows - base < MORSEL ? j->in->n_rows - base : MORSEL);
   uint16_t sv1[MORSEL], sv2[MORSEL];
   uint16_t* sel;
   int32_t n = morsel_filter(j->in, j->preds, j->n_preds, base, len, sv1, sv2, &sel);
   for (int32_t s = 0; s < n; s++) {
      int64_t i = base + sel[s];
      uint64_t h = hash_row(j->in, j->keys, j->n_keys, i);
      entry** slot = &fr->ht[(h >> 6) & (FRAG_HT - 1)];
      entry* e = *slot;
      if (!(e && e->hash == h && keys_equal(j->in, j->keys, (int64_t) e->rep, j->in, j->keys, i, j->n_keys, 1))) {
         e = frag_alloc(fr);
         e->next = NULL;
         e->hash = h;
         e->rep = (uint32_t) i;
         agg_init(e->st, j->n_aggs);
         evec_push(&fr->outputs[h & (FRAG_OUT - 1)], e);
         *slot = e;
      }
      agg_update(j->in, j->aggs, j->n_aggs, i, e->st);
   }
}

typedef struct {
   gb_job* job;
   int32_t n_frags;
   evec result[FRAG_OUT];
} merge_job;
static uint64_t next_pow2(uint64_t v) {
   if (v == 0) return 0; /* (0-1)|... +1 wraps to 0, callers take max(.,1) */
   v--;
   v |= v >> 1;
   v |= v >> 2;
   v |= v >> 4;
   v |= v >> 8;
   v |= v >> 16;
   v |= v >> 32;
   return v + 1;
}
/* slot word = entry pointer << 16 | accumulated bloom tag (tag/untag/matchesTag, helpers.h:326-346) */
static inline uint64_t slot_tag(entry* e, uint64_t prev, uint64_t hash) {
   return ((uint64_t) (uintptr_t) e << 16) | (uint16_t) (ora_bloom_mask((uint32_t) (hash >> 53)) | (uint16_t) prev);
}
static inline entry* slot_untag(uint64_t s) { return (entry*) (uintptr_t) (s >> 16); }
static inline int slot_matches(uint64_t s, uint64_t hash) {
   uint16_t tag = ora_bloom_mask((uint32_t) (hash >> 53));
   return !(tag & ~(uint16_t) s);
}

/* PreAggregationHashtable::merge, one partition (PreAggregationHashtable.cpp:98-155): chained
 * table of nextPow2(1.25 * total) tagged slots at (hash >> 6) & mask; equal hash && eq → combine
 * else push-front. */
static void merge_unit(void* a, int worker, int64_t part) {
   (void) worker;
   merge_job* m = (merge_job*) a;
   gb_job* j = m->job;
   size_t total = 0;
   for (int32_t f = 0; f < m->n_frags; f++) total += (size_t) j->frags[f].outputs[part].n;
   uint64_t ht_size = next_pow2((uint64_t) ((double) total * 1.25));
   if (ht_size < 1) ht_size = 1;
   uint64_t mask = ht_size - 1;
   uint64_t* ht = (uint64_t*) calloc((size_t) ht_size, sizeof(uint64_t));
   for (int32_t f = 0; f < m->n_frags; f++) {
      evec* o = &j->frags[f].outputs[part];
      for (int64_t k = 0; k < o->n; k++) {
         entry* cur = o->p[k];
         uint64_t pos = (cur->hash >> 6) & mask;
         entry* cand = slot_untag(ht[pos]);
         int merged = 0;
         while (cand) {
            if (cand->hash == cur->hash &&
                keys_equal(j->in, j->keys, (int64_t) cand->rep, j->in, j->keys, (int64_t) cur->rep, j->n_keys, 1)) {
               agg_combine(j->aggs, j->n_aggs, cand->st, cur->st);
               merged = 1;
               break;
            }
            cand = cand->next;
         }
         if (!merged) {
            evec_push(&m->result[part], cur);
            uint64_t prev = ht[pos];
            ht[pos] = slot_tag(cur, prev, cur->hash);
            cur->next = slot_untag(prev);
         }
      }
   }
   free(ht);
}

int64_t ora_groupby(const ora_rel* in, const ldb_filter_desc* preds, int32_t n_preds, const ldb_colref* keys, int32_t n_keys,
                    const ldb_agg_spec* aggs, int32_t n_aggs, int32_t threads, uint32_t* rep_rows, int64_t* vals, uint8_t* valid,
                    int64_t cap) {
   if (threads < 1) threads = 1;
   size_t esz = sizeof(entry) + sizeof(agg_state) * (size_t) n_aggs;
   esz = (esz + 15) & ~(size_t) 15;
   fragment* frags = (fragment*) calloc((size_t) threads, sizeof(fragment));
   for (int t = 0; t < threads; t++) frags[t].entry_size = esz;
   gb_job job = {in, preds, n_preds, keys, n_keys, aggs, n_aggs, frags};
   int64_t n_units = (in->n_rows + MORSEL - 1) / MORSEL;
   run_units(threads, n_units, gb_unit, &job);
   merge_job* mj = (merge_job*) calloc(1, sizeof(merge_job));
   mj->job = &job;
   mj->n_frags = threads;
   run_units(threads, FRAG_OUT, merge_unit, mj);
   int64_t ng = 0;
   if (n_keys == 0) {
      /* keyless aggregation = SimpleState (SimpleState.cpp:8-30): exactly one output row even
       * over empty input (state initialised once, merged across workers). */
      agg_state* acc = (agg_state*) calloc((size_t) (n_aggs ? n_aggs : 1), sizeof(agg_state));
      for (int32_t a = 0; a < n_aggs; a++)
         if (aggs[a].fn == LDB_AGG_COUNT || aggs[a].fn == LDB_AGG_COUNT_STAR) acc[a].valid = 1;
      for (int p = 0; p < FRAG_OUT; p++)
         for (int64_t k = 0; k < mj->result[p].n; k++) agg_combine(aggs, n_aggs, acc, mj->result[p].p[k]->st);
      entry* e = (entry*) calloc(1, esz);
      memcpy(e->st, acc, sizeof(agg_state) * (size_t) n_aggs);
      e->rep = 0;
      for (int p = 0; p < FRAG_OUT; p++) mj->result[p].n = 0;
      evec_push(&mj->result[0], e);
      free(acc);
   }
   for (int p = 0; p < FRAG_OUT; p++) {
      for (int64_t k = 0; k < mj->result[p].n; k++) {
         entry* e = mj->result[p].p[k];
         if (ng < cap) {
            rep_rows[ng] = e->rep;
            for (int32_t a = 0; a < n_aggs; a++) {
               const ldb_agg_spec* sp = &aggs[a];
               agg_state* s = &e->st[a];
               i128 v = s->v;
               uint8_t ok = s->valid;
               if (sp->fn == LDB_AGG_COUNT || sp->fn == LDB_AGG_COUNT_STAR) {
                  v = s->cnt;
                  ok = 1;
               } else if (sp->fn == LDB_AGG_AVG) {
                  /* AVG = SUM / COUNT: (sum * 10^k) sdiv count in i128 (DecimalOpScaledLowering,
                   * LowerToStd.cpp:631-651; type rule sql_analyzer.cpp:2636-2642) */
                  if (sp->arg.is_float) {
                     s->f = s->cnt ? s->f / (double) s->cnt : 0;
                  } else if (s->cnt) {
                     v = wrap_mul(s->v, POW10[sp->avg_pow10]) / (i128) s->cnt;
                  } else {
                     ok = 0;
                  }
               }
               if (sp->arg.is_float && sp->fn != LDB_AGG_COUNT && sp->fn != LDB_AGG_COUNT_STAR) {
                  int64_t bits;
                  memcpy(&bits, &s->f, 8);
                  vals[2 * (ng * n_aggs + a)] = bits;
                  vals[2 * (ng * n_aggs + a) + 1] = 0;
               } else {
                  if (!sp->wide && sp->fn != LDB_AGG_AVG) v = (int64_t) v;
                  vals[2 * (ng * n_aggs + a)] = (int64_t) (uint64_t) v;
                  vals[2 * (ng * n_aggs + a) + 1] = (int64_t) (v >> 64);
               }
               valid[ng * n_aggs + a] = ok;
            }
         }
         ng++;
      }
      free(mj->result[p].p);
   }
   if (n_keys == 0) { /* the synthetic entry */
   }
   for (int t = 0; t < threads; t++) {
      for (int b = 0; b < frags[t].n_blocks; b++) free(frags[t].blocks[b]);
      free(frags[t].blocks);
      for (int p = 0; p < FRAG_OUT; p++) free(frags[t].outputs[p].p);
   }
   free(frags);
   free(mj);
   return ng;
}

/* ====================================================================== hash join (a6, a7, a8) */
typedef struct jentry {
   struct jentry* next;
   uint64_t hash;
   uint32_t row; /* logical build row (the reference copies keys+payload here, SpecializeSubOpPass.cpp:70-84) */
} jentry;

typedef struct {
   const ora_rel* build;
   const ldb_colref* bkeys;
   int32_t n_keys;
   jentry* rows;
   _Atomic uint64_t* ht;
   uint64_t mask;
} jbuild_job;

/* build pipeline: map{hash = db.hash(keys)} + materialize (GrowingBuffer::insert) */
static void jmat_unit(void* a, int worker, int64_t u) {
   (void) worker;
   jbuild_job* j = (jbuild_job*) a;
   int64_t base = u * MORSEL, end = base + MORSEL < j->build->n_rows ? base + MORSEL : j->build->n_rows;
   for (int64_t i = base; i < end; i++) {
      j->rows[i].next = NULL;
      j->rows[i].hash = hash_row(j->build, j->bkeys, j->n_keys, i);
      j->rows[i].row = (uint32_t) i;
   }
}
/* HashIndexedView::build (LazyJoinHashtable.cpp:12-34): parallel CAS push-front with tag */
static void jbuild_unit(void* a, int worker, int64_t u) {
   (void) worker;
   jbuild_job* j = (jbuild_job*) a;
   int64_t base = u * MORSEL, end = base + MORSEL < j->build->n_rows ? base + MORSEL : j->build->n_rows;
   for (int64_t i = base; i < end; i++) {
      jentry* e = &j->rows[i];
      uint64_t pos = e->hash & j->mask;
      uint64_t cur = atomic_load(&j->ht[pos]), nw;
      do {
         e->next = (jentry*) (uintptr_t) (cur >> 16);
         nw = ((uint64_t) (uintptr_t) e << 16) | (uint16_t) (ora_bloom_mask((uint32_t) (e->hash >> 53)) | (uint16_t) cur);
      } while (!atomic_compare_exchange_weak(&j->ht[pos], &cur, nw));
   }
}

typedef struct {
   jbuild_job* b;
   const ora_rel* probe;
   const ldb_colref* pkeys;
   int32_t kind;
   /* per-unit staging: counts then write */
   int64_t* unit_counts;
   int64_t* unit_offsets; /* NULL in the counting pass */
   uint32_t* out_probe;
   uint32_t* out_build;
   uint8_t* out_mark;
   int64_t cap;
   int cmp_hash;
} jprobe_job;

/* LookupHashIndexedViewLowering (SubOpToControlFlow.cpp:2558-2586): slot = ht[hash & mask];
 * tag check; ScanListLowering (:2254-2313): walk chain, optional hash compare
 * (compareHashForLookup is dropped for a single integer key, SpecializeSubOpPass.cpp:110-118),
 * key re-check, emit. */
static void jprobe_unit(void* a, int worker, int64_t u) {
   (void) worker;
   jprobe_job* j = (jprobe_job*) a;
   jbuild_job* b = j->b;
   int64_t base = u * MORSEL, end = base + MORSEL < j->probe->n_rows ? base + MORSEL : j->probe->n_rows;
   int64_t n = 0;
   int64_t off = j->unit_offsets ? j->unit_offsets[u] : 0;
   int writing = j->unit_offsets != NULL;
   for (int64_t i = base; i < end; i++) {
      uint64_t h = hash_row(j->probe, j->pkeys, b->n_keys, i);
      uint64_t slot = atomic_load(&b->ht[h & b->mask]);
      int64_t matches = 0;
      if (slot_matches(slot, h)) {
         for (jentry* e = (jentry*) (uintptr_t) (slot >> 16); e; e = e->next) {
            if (j->cmp_hash && e->hash != h) continue;
            if (!keys_equal(b->build, b->bkeys, (int64_t) e->row, j->probe, j->pkeys, i, b->n_keys, 0)) continue;
            matches++;
            if (j->kind == LDB_JOIN_INNER || j->kind == LDB_JOIN_LEFT_OUTER || (j->kind == LDB_JOIN_SINGLE && matches == 1)) {
               if (writing && off + n < j->cap) {
                  j->out_probe[off + n] = (uint32_t) i;
                  j->out_build[off + n] = e->row;
               }
               n++;
            } else if (j->kind != LDB_JOIN_SINGLE) {
               break; /* semi / anti / mark need only existence */
            }
         }
      }
      switch (j->kind) {
         case LDB_JOIN_SEMI:
            if (matches) {
               if (writing && off + n < j->cap) j->out_probe[off + n] = (uint32_t) i;
               n++;
            }
            break;
         case LDB_JOIN_ANTI:
            if (!matches) {
               if (writing && off + n < j->cap) j->out_probe[off + n] = (uint32_t) i;
               n++;
            }
            break;
         case LDB_JOIN_MARK:
            if (writing && i < j->cap) {
               j->out_probe[i] = (uint32_t) i;
               j->out_mark[i] = matches ? 1 : 0;
            }
            n++;
            break;
         case LDB_JOIN_LEFT_OUTER:
         case LDB_JOIN_SINGLE:
            if (!matches) {
               if (writing && off + n < j->cap) {
                  j->out_probe[off + n] = (uint32_t) i;
                  j->out_build[off + n] = LDB_NULL_ROW;
               }
               n++;
            }
            break;
         default: break;
      }
   }
   if (!writing) j->unit_counts[u] = n;
}

/* reverseSides semi / anti join (translateHJWithMarker, RelAlgToSubOp.cpp:1248-1287): probe
 * tuples set a boolean flag in the matching build rows (atomic OR when parallel,
 * SubOpToControlFlow.cpp:3593), then the build buffer is scanned on the flag. */
typedef struct {
   jbuild_job* b;
   const ora_rel* probe;
   const ldb_colref* pkeys;
   _Atomic uint8_t* flags;
   int cmp_hash;
} jmark_job;
static void jmark_unit(void* a, int worker, int64_t u) {
   (void) worker;
   jmark_job* j = (jmark_job*) a;
   jbuild_job* b = j->b;
   int64_t base = u * MORSEL, end = base + MORSEL < j->probe->n_rows ? base + MORSEL : j->probe->n_rows;
   for (int64_t i = base; i < end; i++) {
      uint64_t h = hash_row(j->probe, j->pkeys, b->n_keys, i);
      uint64_t slot = atomic_load(&b->ht[h & b->mask]);
      if (!slot_matches(slot, h)) continue;
      for (jentry* e = (jentry*) (uintptr_t) (slot >> 16); e; e = e->next) {
         if (j->cmp_hash && e->hash != h) continue;
         if (!keys_equal(b->build, b->bkeys, (int64_t) e->row, j->probe, j->pkeys, i, b->n_keys, 0)) continue;
         atomic_store(&j->flags[e->row], 1);
      }
   }
}

int64_t ora_join(const ora_rel* build, const ldb_colref* bkeys, const ora_rel* probe, const ldb_colref* pkeys, int32_t n_keys,
                 int32_t kind, int32_t threads, uint32_t* out_probe, uint32_t* out_build, uint8_t* out_mark, int64_t cap) {
   if (threads < 1) threads = 1;
   if (kind == LDB_JOIN_SEMI_BUILD || kind == LDB_JOIN_ANTI_BUILD) {
      jbuild_job b;
      b.build = build;
      b.bkeys = bkeys;
      b.n_keys = n_keys;
      b.rows = (jentry*) malloc(sizeof(jentry) * (size_t) (build->n_rows ? build->n_rows : 1));
      uint64_t hs = next_pow2((uint64_t) ((double) build->n_rows * 1.25));
      if (hs < 1) hs = 1;
      b.mask = hs - 1;
      b.ht = (_Atomic uint64_t*) calloc((size_t) hs, sizeof(uint64_t));
      int64_t bu = (build->n_rows + MORSEL - 1) / MORSEL;
      run_units(threads, bu, jmat_unit, &b);
      run_units(threads, bu, jbuild_unit, &b);
      jmark_job mj;
      mj.b = &b;
      mj.probe = probe;
      mj.pkeys = pkeys;
      mj.flags = (_Atomic uint8_t*) calloc((size_t) (build->n_rows ? build->n_rows : 1), 1);
      const ora_col* k0 = rel_col(build, bkeys[0]);
      mj.cmp_hash = !(n_keys == 1 && !is_string(k0) && !is_float(k0));
      run_units(threads, (probe->n_rows + MORSEL - 1) / MORSEL, jmark_unit, &mj);
      int64_t n = 0;
      for (int64_t r = 0; r < build->n_rows; r++) {
         int keep = (mj.flags[r] != 0) == (kind == LDB_JOIN_SEMI_BUILD);
         if (keep) {
            if (out_probe && n < cap) out_probe[n] = (uint32_t) r; /* build row numbers, ascending */
            n++;
         }
      }
      free((void*) mj.flags);
      free(b.rows);
      free((void*) b.ht);
      return n;
   }
   jbuild_job b;
   b.build = build;
   b.bkeys = bkeys;
   b.n_keys = n_keys;
   b.rows = (jentry*) malloc(sizeof(jentry) * (size_t) (build->n_rows ? build->n_rows : 1));
   uint64_t ht_size = next_pow2((uint64_t) ((double) build->n_rows * 1.25)); /* LazyJoinHashtable.cpp:16 */
   if (ht_size < 1) ht_size = 1;
   b.mask = ht_size - 1;
   b.ht = (_Atomic uint64_t*) calloc((size_t) ht_size, sizeof(uint64_t));
   int64_t bu = (build->n_rows + MORSEL - 1) / MORSEL;
   run_units(threads, bu, jmat_unit, &b);
   run_units(threads, bu, jbuild_unit, &b);

   int64_t pu = (probe->n_rows + MORSEL - 1) / MORSEL;
   jprobe_job pj;
   memset(&pj, 0, sizeof(pj));
   pj.b = &b;
   pj.probe = probe;
   pj.pkeys = pkeys;
   pj.kind = kind;
   pj.cap = cap;
   pj.out_probe = out_probe;
   pj.out_build = out_build;
   pj.out_mark = out_mark;
   /* hash compare skipped for one integer key */
   const ora_col* k0 = rel_col(build, bkeys[0]);
   pj.cmp_hash = !(n_keys == 1 && !is_string(k0) && !is_float(k0));
   pj.unit_counts = (int64_t*) calloc((size_t) (pu ? pu : 1), sizeof(int64_t));
   run_units(threads, pu, jprobe_unit, &pj);
   int64_t* offs = (int64_t*) calloc((size_t) (pu ? pu : 1), sizeof(int64_t));
   int64_t total = 0;
   for (int64_t u = 0; u < pu; u++) {
      offs[u] = total;
      total += pj.unit_counts[u];
   }
   if (out_probe) {
      pj.unit_offsets = offs;
      run_units(threads, pu, jprobe_unit, &pj);
   }
   free(offs);
   free(pj.unit_counts);
   free(b.rows);
   free((void*) b.ht);
   return total;
}

/* ====================================================================== sort / top-k (a12, a13) */
/* db.sort_compare three-way per key, DESC by operand swap (LowerToStd.cpp:1046-1064,
 * RelAlgToSubOp.cpp:1620-1666).  Ties keep input order (a stable instance of the
 * reference's unspecified tie order). */
typedef struct {
   const ora_rel* in;
   const ldb_sort_spec* specs;
   int32_t n_specs;
} sort_ctx;
static int sort_cmp_rows(const sort_ctx* c, int64_t ia, int64_t ib) {
   for (int32_t k = 0; k < c->n_specs; k++) {
      const ora_col* col = rel_col(c->in, c->specs[k].col);
      int64_t ra = rel_row(c->in, c->specs[k].col.side, ia), rb = rel_row(c->in, c->specs[k].col.side, ib);
      int r;
      if (is_string(col)) {
         uint32_t la, lb;
         const uint8_t* a = load_str(col, ra, &la);
         const uint8_t* b = load_str(col, rb, &lb);
         r = str_cmp(a, la, b, lb);
      } else if (is_float(col)) {
         double a = load_f64(col, ra), b = load_f64(col, rb);
         r = a < b ? -1 : (a > b ? 1 : 0);
      } else {
         i128 a = load_int(col, ra), b = load_int(col, rb);
         r = a < b ? -1 : (a > b ? 1 : 0);
      }
      if (c->specs[k].descending) r = -r;
      if (r) return r;
   }
   return 0;
}
static void merge_sort(const sort_ctx* c, uint32_t* a, uint32_t* tmp, int64_t n) {
   if (n < 2) return;
   int64_t m = n / 2;
   merge_sort(c, a, tmp, m);
   merge_sort(c, a + m, tmp, n - m);
   int64_t i = 0, j = m, k = 0;
   while (i < m && j < n) tmp[k++] = sort_cmp_rows(c, a[j], a[i]) < 0 ? a[j++] : a[i++];
   while (i < m) tmp[k++] = a[i++];
   while (j < n) tmp[k++] = a[j++];
   memcpy(a, tmp, sizeof(uint32_t) * (size_t) n);
}
void ora_sort(const ora_rel* in, const ldb_sort_spec* specs, int32_t n_specs, uint32_t* out_perm) {
   sort_ctx c = {in, specs, n_specs};
   for (int64_t i = 0; i < in->n_rows; i++) out_perm[i] = (uint32_t) i;
   uint32_t* tmp = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) (in->n_rows ? in->n_rows : 1));
   merge_sort(&c, out_perm, tmp, in->n_rows);
   free(tmp);
}
/* Heap (Heap.cpp:8-72) keeps the k smallest rows under the comparator and emits them sorted;
 * equal to the first k rows of the full sort up to tie order. */
int64_t ora_topk(const ora_rel* in, const ldb_sort_spec* specs, int32_t n_specs, int64_t k, uint32_t* out_perm) {
   uint32_t* full = (uint32_t*) malloc(sizeof(uint32_t) * (size_t) (in->n_rows ? in->n_rows : 1));
   ora_sort(in, specs, n_specs, full);
   int64_t n = in->n_rows < k ? in->n_rows : k;
   memcpy(out_perm, full, sizeof(uint32_t) * (size_t) n);
   free(full);
   return n;
}

/* ================================================================== SURVEY §8(f).4: SegmentTreeView, windows, set operations */
typedef struct ora_st_node {
   int64_t left_min, right_max;
   struct ora_st_node *left, *right;
   __int128 v;
   int ok;
} ora_st_node;
/* the generated combine functions: a NULL state is the identity; COUNT and SUM add, MIN / MAX select */
static void ora_st_combine(int fn, __int128* v, int* ok, __int128 v2, int ok2) {
   if (!ok2) return;
   if (!*ok) {
      *v = v2;
      *ok = 1;
      return;
   }
   if (fn == LDB_WIN_MIN) *v = v2 < *v ? v2 : *v;
   else if (fn == LDB_WIN_MAX) *v = v2 > *v ? v2 : *v;
   else *v = (__int128) ((unsigned __int128) *v + (unsigned __int128) v2);
}
/* SegmentTreeView::buildRecursively (src/runtime/SegmentTreeView.cpp:22-47): from / to inclusive, split at from + (to - from) / 2 */
static ora_st_node* ora_st_build(ora_st_node* pool, int64_t* used, const int64_t* vals, const uint8_t* valid, int fn, int64_t from, int64_t to) {
   ora_st_node* t = &pool[(*used)++];
   t->left_min = from;
   t->right_max = to;
   t->left = t->right = NULL;
   if (from == to) { /* createInitialStateFn(entry) */
      const int ok = valid ? valid[from] != 0 : 1;
      if (fn == LDB_WIN_COUNT) {
         t->v = ok ? 1 : 0;
         t->ok = 1;
      } else {
         t->v = ok ? (__int128) (((unsigned __int128) (uint64_t) vals[2 * from + 1] << 64) | (uint64_t) vals[2 * from]) : 0;
         t->ok = ok;
      }
      return t;
   }
   const int64_t mid = from + (to - from) / 2;
   t->left = ora_st_build(pool, used, vals, valid, fn, from, mid);
   t->right = ora_st_build(pool, used, vals, valid, fn, mid + 1, to);
   t->v = t->left->v;
   t->ok = t->left->ok;
   ora_st_combine(fn, &t->v, &t->ok, t->right->v, t->right->ok); /* combineStatesFn(inner, left, right) */
   return t;
}
/* SegmentTreeView::lookupRecursively (:67-79) */
static void ora_st_lookup(const ora_st_node* t, int fn, int64_t from, int64_t to, __int128* v, int* ok, int* first) {
   if (from <= t->left_min && to >= t->right_max) {
      if (*first) {
         *v = t->v;
         *ok = t->ok;
         *first = 0;
      } else {
         ora_st_combine(fn, v, ok, t->v, t->ok);
      }
   } else if (from <= t->right_max && to >= t->left_min) {
      ora_st_lookup(t->left, fn, from, to, v, ok, first);
      ora_st_lookup(t->right, fn, from, to, v, ok, first);
   }
}
int32_t ora_segment_tree(const int64_t* vals_lohi, const uint8_t* valid, int64_t n, int32_t fn, const int64_t* from, const int64_t* to, int64_t nq, int64_t* out_lohi, uint8_t* out_valid) {
   if (n <= 0 || fn < LDB_WIN_SUM || fn > LDB_WIN_COUNT) return -1;
   ora_st_node* pool = (ora_st_node*) malloc(sizeof(ora_st_node) * (size_t) (2 * n));
   int64_t used = 0;
   ora_st_node* root = ora_st_build(pool, &used, vals_lohi, valid, fn, 0, n - 1);
   for (int64_t q = 0; q < nq; q++) {
      if (from[q] > to[q] || from[q] < 0 || to[q] >= n) { /* lookup throws "from must be <= to" */
         free(pool);
         return -2;
      }
      __int128 v = 0;
      int ok = 0, first = 1;
      ora_st_lookup(root, fn, from[q], to[q], &v, &ok, &first);
      out_lohi[2 * q] = (int64_t) (uint64_t) v;
      out_lohi[2 * q + 1] = (int64_t) (v >> 64);
      out_valid[q] = (uint8_t) (fn == LDB_WIN_COUNT ? 1 : ok);
   }
   free(pool);
   return 0;
}
int32_t ora_window(const int64_t* vals_lohi, const uint8_t* valid, const int64_t* part_start, const int64_t* part_end, int64_t n, int32_t fn, int64_t frame_from, int64_t frame_to,
                   int64_t* out_lohi, uint8_t* out_valid) {
   /* per partition a continuous view (its sorted buffer) with its own segment tree, as WindowLowering builds them */
   for (int64_t p0 = 0; p0 < n;) {
      const int64_t st = part_start[p0], en = part_end[p0], len = en - st;
      ora_st_node* pool = NULL;
      ora_st_node* root = NULL;
      if (fn >= LDB_WIN_SUM && fn <= LDB_WIN_COUNT) {
         pool = (ora_st_node*) malloc(sizeof(ora_st_node) * (size_t) (2 * len));
         int64_t used = 0;
         root = ora_st_build(pool, &used, vals_lohi + 2 * st, valid ? valid + st : NULL, fn, 0, len - 1);
      }
      for (int64_t i = st; i < en; i++) {
         const int64_t cur = i - st, last = len - 1;
         /* GetBeginReference = 0, GetEndReference = len - 1, OffsetReferenceBy = min(len - 1, max(0, cur + off)) */
         int64_t lo = frame_from == INT64_MIN ? 0 : (frame_from == 0 ? cur : cur + frame_from);
         int64_t hi = frame_to == INT64_MAX ? last : (frame_to == 0 ? cur : cur + frame_to);
         lo = lo < 0 ? 0 : (lo > last ? last : lo);
         hi = hi < 0 ? 0 : (hi > last ? last : hi);
         __int128 v = 0;
         int ok = 1;
         if (fn == LDB_WIN_RANK) v = cur - lo + 1; /* EntriesBetween(frame begin, current) + 1 */
         else if (fn == LDB_WIN_COUNT_STAR) v = hi >= lo ? hi - lo + 1 : 0;
         else if (hi >= lo) {
            int first = 1;
            ok = 0;
            ora_st_lookup(root, fn, lo, hi, &v, &ok, &first);
            if (fn == LDB_WIN_COUNT) ok = 1;
         } else {
            ok = fn == LDB_WIN_COUNT;
         }
         out_lohi[2 * i] = (int64_t) (uint64_t) v;
         out_lohi[2 * i + 1] = (int64_t) (v >> 64);
         out_valid[i] = (uint8_t) ok;
      }
      free(pool);
      p0 = en;
   }
   return 0;
}
int64_t ora_setop_multiplicity(int32_t op, int64_t l, int64_t r) {
   switch (op) {
      case LDB_SET_UNION: return 1;
      case LDB_SET_INTERSECT: return (l > 0 && r > 0) ? 1 : 0; /* leftNonZero && rightNonZero (:878-881) */
      case LDB_SET_EXCEPT: return (l > 0 && r == 0) ? 1 : 0; /* leftNonZero && rightZero (:875-877) */
      case LDB_SET_INTERSECT_ALL: return l > r ? r : l; /* select(l > r, r, l) (:898-900) */
      case LDB_SET_EXCEPT_ALL: return l - r < 0 ? 0 : l - r; /* max(l - r, 0) (:893-896) */
      default: return l + r; /* UNION ALL: every row of both inputs */
   }
}
