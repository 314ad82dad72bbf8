// ldb_host.cpp — host mirror of the reference's filter interface + TPC-H plan layer (see ldb_host.hpp).
#include "ldb_host.hpp"
#include <cstring>
#include <regex>

namespace lingodb::runtime::gpu {

static thread_local std::string g_plan_err;

void check(int32_t status, const char* what) {
   if (status != LDB_OK) throw CtxError(std::string(what) + ": " + ldb_gpu_last_error());
}

// days from civil (proleptic Gregorian), Howard Hinnant's algorithm
static int32_t daysFromCivil(int y, unsigned m, unsigned d) {
   y -= m <= 2;
   const int era = (y >= 0 ? y : y - 399) / 400;
   const unsigned yoe = (unsigned) (y - era * 400);
   const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
   const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
   return era * 146097 + (int) doe - 719468;
}
// Restrictions.cpp:17-25: normalise "YYYY-M-DD" then parse as date32
int32_t parseDate32(std::string str) {
   static std::regex r("(\\d\\d\\d\\d)-(\\d)-(\\d\\d)");
   str = std::regex_replace(str, r, "$1-0$2-$3");
   int y, m, d;
   if (str.size() != 10 || sscanf(str.c_str(), "%4d-%2d-%2d", &y, &m, &d) != 3 || m < 1 || m > 12 || d < 1 || d > 31) throw std::runtime_error("could not parse date");
   return daysFromCivil(y, (unsigned) m, (unsigned) d);
}

// arrow::Decimal128::FromString + Rescale(scale_in, scale_out) (Restrictions.cpp:455-468):
// rescaling up multiplies by 10^k; rescaling down must be exact (Arrow returns an error
// otherwise, which the reference turns into ValueOrDie()).
__int128 parseDecimal(const std::string& s, int32_t scale) {
   size_t i = 0;
   bool neg = false;
   if (i < s.size() && (s[i] == '-' || s[i] == '+')) neg = s[i++] == '-';
   __int128 v = 0;
   int32_t frac = 0;
   bool dot = false, any = false;
   for (; i < s.size(); i++) {
      if (s[i] == '.' && !dot) {
         dot = true;
         continue;
      }
      if (s[i] < '0' || s[i] > '9') throw std::runtime_error("could not parse decimal const");
      v = v * 10 + (s[i] - '0');
      if (dot) frac++;
      any = true;
   }
   if (!any) throw std::runtime_error("could not parse decimal const");
   for (; frac < scale; frac++) v *= 10;
   for (; frac > scale; frac--) {
      if (v % 10 != 0) throw std::runtime_error("decimal const loses precision when rescaled");
      v /= 10;
   }
   return neg ? -v : v;
}

static void setInt(ldb_filter_desc& d, __int128 v) {
   d.rhs_kind = LDB_RHS_INT;
   d.value_lo = (uint64_t) v;
   d.value_hi = (int64_t) (v >> 64);
}

std::unique_ptr<Restrictions> Restrictions::create(const std::vector<FilterDescription>& filterDescs, const ldb_table* table, int32_t side) {
   auto res = std::make_unique<Restrictions>();
   for (const auto& fd : filterDescs) {
      int32_t colId = ldb_gpu_table_col_index(table, fd.columnName.c_str());
      if (colId < 0) throw std::runtime_error("unknown column in filter");
      ldb_filter_desc d;
      memset(&d, 0, sizeof(d));
      d.col = {side, colId};
      d.op = (int32_t) fd.op;
      if (fd.op == FilterOp::NOTNULL) {
         res->descs.push_back(d);
         continue;
      }
      ldb_coltype type;
      ldb_gpu_table_coltype(table, colId, &type);
      auto intOf = [&](const std::variant<std::string, int64_t, double>& v) -> int64_t {
         if (!std::holds_alternative<int64_t>(v)) throw std::runtime_error("integer constant expected in filter");
         return std::get<int64_t>(v);
      };
      switch (type.type) {
         case LDB_T_CHAR4: { // char(1): memcpy(<=4 bytes) into an int32 (:409-417)
            const std::string& strVal = std::get<std::string>(fd.value);
            if (strVal.size() > 4) throw std::runtime_error("char(1) filter constant longer than 4 bytes");
            int32_t intVal = 0;
            std::memcpy(&intVal, strVal.data(), strVal.size());
            setInt(d, intVal);
            break;
         }
         case LDB_T_BOOL8: // (device-side computed columns only)
         case LDB_T_INT8:
         case LDB_T_INT16:
         case LDB_T_INT32:
         case LDB_T_INT64: {
            auto cast = [&](int64_t v) -> int64_t { // static_cast<T>(int64) of createSimpleTypeFilter (:342-347)
               switch (type.type) {
                  case LDB_T_INT8: return (int8_t) v;
                  case LDB_T_INT16: return (int16_t) v;
                  case LDB_T_INT32: return (int32_t) v;
                  default: return v;
               }
            };
            if (fd.op == FilterOp::IN) {
               auto vals = std::make_unique<std::vector<int64_t>>();
               for (auto v : std::get<std::vector<int64_t>>(fd.values)) {
                  int64_t c = cast(v);
                  vals->push_back(c);
                  vals->push_back(c >> 63);
               }
               d.rhs_kind = LDB_RHS_INT;
               d.n_in = (int32_t) (vals->size() / 2);
               d.in_values = vals->data();
               res->inInts.push_back(std::move(vals));
            } else {
               setInt(d, cast(intOf(fd.value)));
            }
            break;
         }
         case LDB_T_DATE32: {
            if (fd.op == FilterOp::IN) {
               auto vals = std::make_unique<std::vector<int64_t>>();
               for (const auto& s : std::get<std::vector<std::string>>(fd.values)) {
                  vals->push_back(parseDate32(s));
                  vals->push_back(0);
               }
               d.rhs_kind = LDB_RHS_INT;
               d.n_in = (int32_t) (vals->size() / 2);
               d.in_values = vals->data();
               res->inInts.push_back(std::move(vals));
            } else {
               setInt(d, parseDate32(std::get<std::string>(fd.value)));
            }
            break;
         }
         case LDB_T_DECIMAL128: {
            __int128 v;
            if (std::holds_alternative<std::string>(fd.value)) {
               v = parseDecimal(std::get<std::string>(fd.value), type.scale);
            } else if (std::holds_alternative<int64_t>(fd.value)) {
               v = std::get<int64_t>(fd.value);
               for (int32_t s = type.scale; s > 0; s--) v *= 10; // :470-475
            } else {
               throw std::runtime_error("unsupported decimal constant type");
            }
            if (fd.op == FilterOp::IN) throw std::runtime_error("unsupported filter op");
            setInt(d, v);
            break;
         }
         case LDB_T_UTF8: {
            d.rhs_kind = LDB_RHS_STRING;
            if (fd.op == FilterOp::IN) {
               auto ptrs = std::make_unique<std::vector<const char*>>();
               auto lens = std::make_unique<std::vector<int32_t>>();
               for (const auto& s : std::get<std::vector<std::string>>(fd.values)) {
                  res->strings.push_back(std::make_unique<std::string>(s));
                  ptrs->push_back(res->strings.back()->data());
                  lens->push_back((int32_t) s.size());
               }
               d.n_in = (int32_t) ptrs->size();
               d.in_strs = ptrs->data();
               d.in_str_lens = lens->data();
               res->inStrPtrs.push_back(std::move(ptrs));
               res->inStrLens.push_back(std::move(lens));
            } else {
               res->strings.push_back(std::make_unique<std::string>(std::get<std::string>(fd.value)));
               d.str = res->strings.back()->data();
               d.str_len = (int32_t) res->strings.back()->size();
            }
            break;
         }
         case LDB_T_FLOAT64:
         case LDB_T_FLOAT32: {
            d.rhs_kind = LDB_RHS_FLOAT;
            d.value_f64 = std::holds_alternative<double>(fd.value) ? std::get<double>(fd.value) : (double) intOf(fd.value);
            break;
         }
         default: throw std::runtime_error("unsupported type in filter");
      }
      res->descs.push_back(d);
   }
   return res;
}

// ---------------------------------------------------------------- decimal typing
DecimalType adaptedAfterMulDiv(int64_t p, int64_t s) {
   int64_t beforeComma = p - s;
   if (beforeComma > 32 && s > 6) {
      p = 38;
      s = 6;
   } else if (beforeComma > 32 && s <= 6) {
      p = 38;
   } else {
      p = std::min<int64_t>(p, 38);
      s = std::min<int64_t>(s, 38 - beforeComma);
   }
   return {(int32_t) p, (int32_t) s};
}
DecimalType typeAfterMul(DecimalType a, DecimalType b) { return adaptedAfterMulDiv(a.p + b.p, a.s + b.s); }
DecimalType typeAfterDiv(DecimalType a, DecimalType b) {
   int64_t s = std::max<int64_t>(6, a.s + b.p);
   return adaptedAfterMulDiv(a.p - a.s + b.s + s, s);
}
DecimalType higherDecimalType(DecimalType a, DecimalType b) {
   int32_t hidig = std::max(a.p - a.s, b.p - b.s);
   int32_t maxs = std::max(a.s, b.s);
   return {hidig + maxs, maxs};
}
DecimalType avgType(DecimalType arg) { return typeAfterDiv(arg, {19, 0}); }
int64_t pow10i(int k) {
   int64_t r = 1;
   while (k-- > 0) r *= 10;
   return r;
}

} // namespace lingodb::runtime::gpu

// ================================================================== plans
using namespace lingodb::runtime::gpu;

namespace {

struct Rel {
   ldb_ctx* ctx;
   ldb_rel* r = nullptr;
   Rel(ldb_ctx* c) : ctx(c) {}
   ~Rel() {
      if (r) ldb_gpu_rel_release(ctx, r);
   }
   Rel(const Rel&) = delete;
};
struct Table {
   ldb_ctx* ctx;
   ldb_table* t = nullptr;
   Table(ldb_ctx* c) : ctx(c) {}
   ~Table() {
      if (t) ldb_gpu_table_release(ctx, t);
   }
   ldb_table* release() {
      ldb_table* x = t;
      t = nullptr;
      return x;
   }
};
struct Ht {
   ldb_ctx* ctx;
   ldb_hashtable* h = nullptr;
   Ht(ldb_ctx* c) : ctx(c) {}
   ~Ht() {
      if (h) ldb_gpu_hashtable_release(ctx, h);
   }
};

int32_t colOf(const ldb_table* t, const char* name) {
   int32_t c = ldb_gpu_table_col_index(t, name);
   if (c < 0) throw std::runtime_error(std::string("column not found: ") + name);
   return c;
}
DecimalType decOf(const ldb_table* t, int32_t col) {
   ldb_coltype ct;
   ldb_gpu_table_coltype(t, col, &ct);
   if (ct.type != LDB_T_DECIMAL128) throw std::runtime_error("decimal column expected");
   return {ct.precision, ct.scale};
}

ldb_factor colFactor(ldb_colref c) { return {1, c, 0, 1}; }
// (k - col) or (k + col) with k an integer literal: int → decimal(19,0) → common scale of the
// column (sql_analyzer.cpp:3125-3141, getHigherDecimalType): a = k * 10^scale
ldb_factor constPlusCol(int64_t k, int sign, ldb_colref c, DecimalType colType, DecimalType* outType) {
   *outType = higherDecimalType({19, 0}, colType);
   return {1, c, k * pow10i(colType.s), sign};
}
ldb_expr product(std::initializer_list<ldb_factor> fs) {
   ldb_expr e;
   memset(&e, 0, sizeof(e));
   e.n_terms = 1;
   e.t[0].n_factors = (int32_t) fs.size();
   int i = 0;
   for (auto& f : fs) e.t[0].f[i++] = f;
   return e;
}
ldb_agg_spec sumDec(ldb_expr e, DecimalType t) {
   ldb_agg_spec a;
   memset(&a, 0, sizeof(a));
   a.fn = LDB_AGG_SUM;
   a.arg = e;
   a.wide = t.wide();
   a.out_type = LDB_T_DECIMAL128;
   a.out_precision = t.p;
   a.out_scale = t.s;
   return a;
}
ldb_agg_spec avgDec(ldb_expr e, DecimalType t) {
   ldb_agg_spec a = sumDec(e, t);
   a.fn = LDB_AGG_AVG;
   DecimalType r = avgType(t);
   // (sum * 10^(sRes + s2 - s1)) sdiv count with the divisor typed decimal(19,0) (LowerToStd.cpp:631-651)
   a.avg_pow10 = r.s + 0 - t.s;
   a.out_precision = r.p;
   a.out_scale = r.s;
   return a;
}
ldb_agg_spec countStar() {
   ldb_agg_spec a;
   memset(&a, 0, sizeof(a));
   a.fn = LDB_AGG_COUNT_STAR;
   a.out_type = LDB_T_INT64;
   return a;
}

template <typename F>
int32_t guarded(F&& f) {
   try {
      f();
      return LDB_OK;
   } catch (const std::exception& e) {
      g_plan_err = e.what();
      return LDB_ERR_INVALID;
   }
}

} // namespace

extern "C" const char* ldb_plan_last_error(void) { return g_plan_err.c_str(); }

namespace {
// residual column-vs-column conjunct (evaluated by generated db.compare in the reference, not by
// Restrictions: only column-vs-constant filters are pushed into the scan)
ldb_filter_desc colCompare(ldb_colref a, FilterOp op, ldb_colref b) {
   ldb_filter_desc d;
   memset(&d, 0, sizeof(d));
   d.col = a;
   d.op = (int32_t) op;
   d.rhs_kind = LDB_RHS_COLUMN;
   d.rhs_col = b;
   return d;
}
} // namespace

namespace {
// string predicate evaluated in generated code (StringRuntime::like), not a pushed-down restriction
struct LikePred {
   std::string pattern;
   ldb_filter_desc d;
   LikePred(ldb_colref c, std::string pat, bool negate = false) : pattern(std::move(pat)) {
      memset(&d, 0, sizeof(d));
      d.col = c;
      d.op = negate ? LDB_F_NOT_LIKE : LDB_F_LIKE;
      d.rhs_kind = LDB_RHS_STRING;
      d.str = pattern.data();
      d.str_len = (int32_t) pattern.size();
   }
};
} // namespace

// The single-GPU TPC-H plans are data (lingo-db_amd/plans/tpch/qN.json, interpreted by ldb_plan.cpp).
// What follows are the PIECES of the multi-GPU plans: the shard-local parts and the merges that run
// between the exchanges of tpch_dist.py.
// ---------------------------------------------------------------- multi-GPU plan pieces
// Row-range sharded fact tables: every rank runs the *_partial plan on its shard, the tiny partial
// tables are exchanged over RCCL, and *_final merges them exactly as the reference merges
// thread-local aggregate states (combine = add sums, add counts; AVG = SUM / COUNT afterwards).
namespace {
ldb_agg_spec avgMerge(ldb_colref sumCol, ldb_colref cntCol, DecimalType argType) {
   ldb_agg_spec a = avgDec(product({colFactor(sumCol)}), argType);
   a.has_count_expr = 1;
   a.count_expr = product({colFactor(cntCol)});
   return a;
}
ldb_agg_spec sumInt64(ldb_colref c) {
   ldb_agg_spec a;
   memset(&a, 0, sizeof(a));
   a.fn = LDB_AGG_SUM;
   a.arg = product({colFactor(c)});
   a.out_type = LDB_T_INT64;
   return a;
}
} // namespace

// keys, sum_qty, sum_base_price, sum_disc_price, sum_charge, sum_disc, count — no AVG division yet
extern "C" int32_t ldb_plan_tpch_q1_partial(ldb_ctx* ctx, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel scan(ctx);
      check(ldb_gpu_rel_from_table(ctx, li, &scan.r), "q1 scan");
      auto restr = Restrictions::create({{"l_shipdate", FilterOp::LTE, std::string("1998-09-02"), {}}}, li);
      ldb_colref qty{0, colOf(li, "l_quantity")}, ext{0, colOf(li, "l_extendedprice")}, disc{0, colOf(li, "l_discount")}, tax{0, colOf(li, "l_tax")};
      ldb_colref keys[2] = {{0, colOf(li, "l_returnflag")}, {0, colOf(li, "l_linestatus")}};
      DecimalType tq = decOf(li, qty.col), te = decOf(li, ext.col), td = decOf(li, disc.col), tt = decOf(li, tax.col);
      DecimalType t1md, t1pt;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, td, &t1md);
      ldb_factor onePlusTax = constPlusCol(1, +1, tax, tt, &t1pt);
      DecimalType tDiscPrice = typeAfterMul(te, t1md), tCharge = typeAfterMul(tDiscPrice, t1pt);
      ldb_agg_spec aggs[6] = {sumDec(product({colFactor(qty)}), tq),
                              sumDec(product({colFactor(ext)}), te),
                              sumDec(product({colFactor(ext), oneMinusDisc}), tDiscPrice),
                              sumDec(product({colFactor(ext), oneMinusDisc, onePlusTax}), tCharge),
                              sumDec(product({colFactor(disc)}), td),
                              countStar()};
      check(ldb_gpu_groupby(ctx, scan.r, restr->data(), restr->size(), keys, 2, aggs, 6, 6, result), "q1 partial groupby");
   });
}
// partials: the concatenation of every rank's q1_partial table (same column order)
extern "C" int32_t ldb_plan_tpch_q1_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q1 final");
      ldb_colref keys[2] = {{0, 0}, {0, 1}};
      DecimalType tq = decOf(partials, 2), te = decOf(partials, 3), tdp = decOf(partials, 4), tch = decOf(partials, 5), td = decOf(partials, 6);
      ldb_colref cnt{0, 7};
      ldb_agg_spec aggs[8] = {sumDec(product({colFactor({0, 2})}), tq), sumDec(product({colFactor({0, 3})}), te), sumDec(product({colFactor({0, 4})}), tdp),
                              sumDec(product({colFactor({0, 5})}), tch), avgMerge({0, 2}, cnt, tq), avgMerge({0, 3}, cnt, te), avgMerge({0, 6}, cnt, td), sumInt64(cnt)};
      Table grouped(ctx);
      check(ldb_gpu_groupby(ctx, in.r, nullptr, 0, keys, 2, aggs, 8, 6, &grouped.t), "q1 final groupby");
      Rel g(ctx);
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q1 final rel");
      ldb_sort_spec specs[2] = {{{0, 0}, 0, 0}, {{0, 1}, 0, 0}};
      check(ldb_gpu_sort(ctx, g.r, specs, 2, &sorted.r), "q1 final sort");
      ldb_colref outc[10];
      for (int c = 0; c < 10; c++) outc[c] = {0, c};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 10, result), "q1 final materialize");
   });
}
// partials: one row per rank (column 0 = that rank's Q6 revenue, NULL if nothing passed there)
extern "C" int32_t ldb_plan_tpch_q6_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q6 final");
      ldb_agg_spec agg = sumDec(product({colFactor({0, 0})}), decOf(partials, 0));
      check(ldb_gpu_groupby(ctx, in.r, nullptr, 0, nullptr, 0, &agg, 1, 1, result), "q6 final aggregate");
   });
}
// Q3 step 1: c_custkey of the customers of this shard with c_mktsegment = 'BUILDING' (to be replicated)
extern "C" int32_t ldb_plan_tpch_q3_customers(ldb_ctx* ctx, const ldb_table* cust, ldb_table** result) {
   return guarded([&] {
      Rel c0(ctx), c1(ctx);
      check(ldb_gpu_rel_from_table(ctx, cust, &c0.r), "q3 customer");
      auto rc = Restrictions::create({{"c_mktsegment", FilterOp::EQ, std::string("BUILDING"), {}}}, cust);
      check(ldb_gpu_scan_filter(ctx, c0.r, rc->data(), rc->size(), &c1.r), "q3 filter customer");
      ldb_colref ck{0, colOf(cust, "c_custkey")};
      check(ldb_gpu_materialize(ctx, c1.r, &ck, 1, result), "q3 customer keys");
   });
}
// Q3 step 2 on one shard: `custkeys` = the replicated filtered customer keys (column 0); orders and
// lineitem are co-partitioned by order range, so both joins and the group-by are shard-local.
// Result: the shard's top-10 (l_orderkey, revenue, o_orderdate, o_shippriority).
extern "C" int32_t ldb_plan_tpch_q3_local(ldb_ctx* ctx, const ldb_table* custkeys, const ldb_table* ord, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel c1(ctx), o0(ctx), o1(ctx), l0(ctx), l1(ctx), co(ctx), lco(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, custkeys, &c1.r), "q3 customer keys");
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q3 orders");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q3 lineitem");
      auto ro = Restrictions::create({{"o_orderdate", FilterOp::LT, std::string("1995-03-15"), {}}}, ord);
      auto rl = Restrictions::create({{"l_shipdate", FilterOp::GT, std::string("1995-03-15"), {}}}, li);
      check(ldb_gpu_scan_filter(ctx, o0.r, ro->data(), ro->size(), &o1.r), "q3 filter orders");
      check(ldb_gpu_scan_filter(ctx, l0.r, rl->data(), rl->size(), &l1.r), "q3 filter lineitem");
      Ht hc(ctx), ho(ctx);
      ldb_colref ck{0, 0}, ock{0, colOf(ord, "o_custkey")};
      check(ldb_gpu_join_build(ctx, c1.r, &ck, 1, 1, &hc.h), "q3 build customer");
      check(ldb_gpu_join_probe(ctx, hc.h, o1.r, &ock, 1, LDB_JOIN_INNER, &co.r, nullptr), "q3 probe orders");
      ldb_colref ook{0, colOf(ord, "o_orderkey")}, lok{0, colOf(li, "l_orderkey")};
      check(ldb_gpu_join_build(ctx, co.r, &ook, 1, 1, &ho.h), "q3 build orders");
      check(ldb_gpu_join_probe(ctx, ho.h, l1.r, &lok, 1, LDB_JOIN_INNER, &lco.r, nullptr), "q3 probe lineitem");
      ldb_colref ext{0, colOf(li, "l_extendedprice")}, disc{0, colOf(li, "l_discount")};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(li, disc.col), &t1md);
      DecimalType tRev = typeAfterMul(decOf(li, ext.col), t1md);
      ldb_agg_spec agg = sumDec(product({colFactor(ext), oneMinusDisc}), tRev);
      ldb_colref keys[3] = {lok, {1, colOf(ord, "o_orderdate")}, {1, colOf(ord, "o_shippriority")}};
      Table grouped(ctx);
      int64_t est = ldb_gpu_rel_rows(ctx, lco.r);
      check(ldb_gpu_groupby(ctx, lco.r, nullptr, 0, keys, 3, &agg, 1, est > 0 ? est : 1, &grouped.t), "q3 groupby");
      Rel g(ctx);
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q3 rel");
      ldb_sort_spec specs[2] = {{{0, 3}, 1, 0}, {{0, 1}, 0, 0}};
      check(ldb_gpu_topk(ctx, g.r, specs, 2, 10, &top.r), "q3 topk");
      ldb_colref outc[4] = {{0, 0}, {0, 3}, {0, 1}, {0, 2}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 4, result), "q3 materialize");
   });
}
// Q3 step 3: global top-10 of the gathered shard top-10s (order keys are disjoint across shards)
extern "C" int32_t ldb_plan_tpch_q3_final(ldb_ctx* ctx, const ldb_table* tops, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, tops, &in.r), "q3 final");
      ldb_sort_spec specs[2] = {{{0, 1}, 1, 0}, {{0, 2}, 0, 0}};
      check(ldb_gpu_topk(ctx, in.r, specs, 2, 10, &top.r), "q3 final topk");
      ldb_colref outc[4] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 4, result), "q3 final materialize");
   });
}

// Q4 / Q12 multi-GPU: orders and their lineitems live on the same rank, so every rank runs the
// single-GPU plan on its shard; the gathered per-rank rows are merged by summing the counts
// (the reference's combine step) and re-sorted.
namespace {
ldb_agg_spec sumIntCol(ldb_colref c, int32_t out_type) {
   ldb_agg_spec a = sumInt64(c);
   a.out_type = out_type;
   return a;
}
void mergeCounts(ldb_ctx* ctx, const ldb_table* partials, int n_counts, int32_t out_type, ldb_table** result, const char* what) {
   Rel in(ctx), sorted(ctx), g(ctx);
   check(ldb_gpu_rel_from_table(ctx, partials, &in.r), what);
   ldb_colref key{0, 0};
   ldb_agg_spec aggs[4];
   for (int a = 0; a < n_counts; a++) aggs[a] = sumIntCol({0, 1 + a}, out_type);
   Table grouped(ctx);
   check(ldb_gpu_groupby(ctx, in.r, nullptr, 0, &key, 1, aggs, n_counts, 8, &grouped.t), what);
   check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), what);
   ldb_sort_spec spec{{0, 0}, 0, 0};
   check(ldb_gpu_sort(ctx, g.r, &spec, 1, &sorted.r), what);
   ldb_colref outc[5];
   for (int c = 0; c < 1 + n_counts; c++) outc[c] = {0, c};
   check(ldb_gpu_materialize(ctx, sorted.r, outc, 1 + n_counts, result), what);
}
} // namespace
extern "C" int32_t ldb_plan_tpch_q4_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result) {
   return guarded([&] { mergeCounts(ctx, partials, 1, LDB_T_INT64, result, "q4 final"); });
}
extern "C" int32_t ldb_plan_tpch_q12_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result) {
   return guarded([&] { mergeCounts(ctx, partials, 2, LDB_T_INT32, result, "q12 final"); });
}

// Q18 multi-GPU.  Step 1 (per shard; orders and lineitem are co-partitioned): the big group-by,
// HAVING, the join back to orders and lineitem, and the shard's top-100 — customer columns are
// not needed yet: (o_custkey, o_orderkey, o_orderdate, o_totalprice, sum_qty).
extern "C" int32_t ldb_plan_tpch_q18_local(ldb_ctx* ctx, const ldb_table* ord, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel l0(ctx), o0(ctx), g0(ctx), g1(ctx), o1(ctx), lo(ctx), top(ctx), g(ctx);
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q18 lineitem");
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q18 orders");
      ldb_colref lok{0, colOf(li, "l_orderkey")}, qty{0, colOf(li, "l_quantity")};
      ldb_agg_spec sumq = sumDec(product({colFactor(qty)}), decOf(li, qty.col));
      Table perOrder(ctx), grouped(ctx);
      check(ldb_gpu_groupby(ctx, l0.r, nullptr, 0, &lok, 1, &sumq, 1, std::max<int64_t>(1, ldb_gpu_table_rows(ord)), &perOrder.t), "q18 group by l_orderkey");
      check(ldb_gpu_rel_from_table(ctx, perOrder.t, &g0.r), "q18 rel");
      auto having = Restrictions::create({{"agg0", FilterOp::GT, (int64_t) 300, {}}}, perOrder.t);
      check(ldb_gpu_scan_filter(ctx, g0.r, having->data(), having->size(), &g1.r), "q18 having");
      Ht hk(ctx), ho(ctx);
      ldb_colref gk{0, 0}, ook{0, colOf(ord, "o_orderkey")};
      check(ldb_gpu_join_build(ctx, g1.r, &gk, 1, 1, &hk.h), "q18 build keys");
      check(ldb_gpu_join_probe(ctx, hk.h, o0.r, &ook, 1, LDB_JOIN_SEMI, &o1.r, nullptr), "q18 semi join orders");
      check(ldb_gpu_join_build(ctx, o1.r, &ook, 1, 1, &ho.h), "q18 build orders");
      check(ldb_gpu_join_probe(ctx, ho.h, l0.r, &lok, 1, LDB_JOIN_INNER, &lo.r, nullptr), "q18 probe lineitem"); // sides: lineitem, orders
      ldb_colref keys[4] = {{1, colOf(ord, "o_custkey")}, {1, ook.col}, {1, colOf(ord, "o_orderdate")}, {1, colOf(ord, "o_totalprice")}};
      check(ldb_gpu_groupby(ctx, lo.r, nullptr, 0, keys, 4, &sumq, 1, std::max<int64_t>(1, ldb_gpu_rel_rows(ctx, o1.r)), &grouped.t), "q18 groupby");
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q18 rel");
      ldb_sort_spec specs[2] = {{{0, 3}, 1, 0}, {{0, 2}, 0, 0}};
      check(ldb_gpu_topk(ctx, g.r, specs, 2, 100, &top.r), "q18 local topk");
      ldb_colref outc[5] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 5, result), "q18 local materialize");
   });
}
// Step 2 (replicated): the global top-100 of the gathered shard top-100s.
extern "C" int32_t ldb_plan_tpch_q18_mid(ldb_ctx* ctx, const ldb_table* tops, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, tops, &in.r), "q18 mid");
      ldb_sort_spec specs[2] = {{{0, 3}, 1, 0}, {{0, 2}, 0, 0}};
      check(ldb_gpu_topk(ctx, in.r, specs, 2, 100, &top.r), "q18 mid topk");
      ldb_colref outc[5] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 5, result), "q18 mid materialize");
   });
}
// Step 3 (per shard): c_name for the winners whose customer lives in this rank's customer shard.
extern "C" int32_t ldb_plan_tpch_q18_names(ldb_ctx* ctx, const ldb_table* top100, const ldb_table* cust, ldb_table** result) {
   return guarded([&] {
      Rel t0(ctx), c0(ctx), ct(ctx);
      check(ldb_gpu_rel_from_table(ctx, top100, &t0.r), "q18 names");
      check(ldb_gpu_rel_from_table(ctx, cust, &c0.r), "q18 names customer");
      Ht ht(ctx);
      ldb_colref tk{0, 0}, ck{0, colOf(cust, "c_custkey")};
      check(ldb_gpu_join_build(ctx, t0.r, &tk, 1, 0, &ht.h), "q18 names build");
      check(ldb_gpu_join_probe(ctx, ht.h, c0.r, &ck, 1, LDB_JOIN_INNER, &ct.r, nullptr), "q18 names probe"); // sides: customer, top100
      ldb_colref outc[6] = {{0, colOf(cust, "c_name")}, {0, ck.col}, {1, 1}, {1, 2}, {1, 3}, {1, 4}};
      check(ldb_gpu_materialize(ctx, ct.r, outc, 6, result), "q18 names materialize");
   });
}
// Step 4 (replicated): order the gathered rows.
extern "C" int32_t ldb_plan_tpch_q18_final(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, rows, &in.r), "q18 final");
      ldb_sort_spec specs[2] = {{{0, 4}, 1, 0}, {{0, 3}, 0, 0}};
      check(ldb_gpu_topk(ctx, in.r, specs, 2, 100, &top.r), "q18 final topk");
      ldb_colref outc[6] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 6, result), "q18 final materialize");
   });
}

// Q9 multi-GPU (broadcast the filtered part keys, co-partition lineitem and partsupp
// on the part key with one all-to-all each, join locally, merge the tiny partial aggregates).
// lineitem/orders are sharded by order ranges (co-located), part / partsupp / supplier by rows.
// Step 1 (per shard): the keys of this shard's "green" parts → all-gathered by the caller.
extern "C" int32_t ldb_plan_tpch_q9_green(ldb_ctx* ctx, const ldb_table* part, ldb_table** result) {
   return guarded([&] {
      Rel p0(ctx), p1(ctx);
      check(ldb_gpu_rel_from_table(ctx, part, &p0.r), "q9 part");
      LikePred green({0, colOf(part, "p_name")}, "%green%");
      check(ldb_gpu_scan_filter(ctx, p0.r, &green.d, 1, &p1.r), "q9 filter part");
      ldb_colref pk{0, colOf(part, "p_partkey")};
      check(ldb_gpu_materialize(ctx, p1.r, &pk, 1, result), "q9 green keys");
   });
}
// Step 2a (per shard): this shard's lineitems of green parts joined with their (co-located)
// orders, as rows (l_partkey, l_suppkey, o_year, l_extendedprice, l_discount, l_quantity) grouped
// by destination rank = hash-radix of l_partkey; counts[world] rows per destination.
extern "C" int32_t ldb_plan_tpch_q9_lineitem_side(ldb_ctx* ctx, const ldb_table* greenkeys, const ldb_table* li, const ldb_table* ord, int32_t world, ldb_table** result,
                                                  int64_t* counts) {
   return guarded([&] {
      Rel g0(ctx), l0(ctx), lp(ctx), o0(ctx), lo(ctx), loy(ctx);
      check(ldb_gpu_rel_from_table(ctx, greenkeys, &g0.r), "q9 green keys");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q9 lineitem");
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q9 orders");
      Ht hg(ctx), ho(ctx);
      ldb_colref gk{0, 0}, lpk{0, colOf(li, "l_partkey")}, lok{0, colOf(li, "l_orderkey")}, ook{0, colOf(ord, "o_orderkey")};
      check(ldb_gpu_join_build(ctx, g0.r, &gk, 1, 1, &hg.h), "q9 build green");
      check(ldb_gpu_join_probe(ctx, hg.h, l0.r, &lpk, 1, LDB_JOIN_SEMI, &lp.r, nullptr), "q9 lineitem of green parts");
      check(ldb_gpu_join_build(ctx, lp.r, &lok, 1, 0, &ho.h), "q9 build reduced lineitem");
      check(ldb_gpu_join_probe(ctx, ho.h, o0.r, &ook, 1, LDB_JOIN_INNER, &lo.r, nullptr), "q9 probe orders"); // sides: orders, lineitem
      Table years(ctx);
      check(ldb_gpu_map_column(ctx, lo.r, {0, colOf(ord, "o_orderdate")}, LDB_FN_EXTRACT_YEAR, "o_year", &years.t), "q9 extract year");
      check(ldb_gpu_rel_zip(ctx, lo.r, years.t, &loy.r), "q9 zip year");
      ldb_colref key{1, lpk.col};
      ldb_colref cols[6] = {{1, lpk.col}, {1, colOf(li, "l_suppkey")}, {2, 0}, {1, colOf(li, "l_extendedprice")}, {1, colOf(li, "l_discount")}, {1, colOf(li, "l_quantity")}};
      check(ldb_gpu_partition(ctx, loy.r, &key, 1, world, cols, 6, result, counts), "q9 partition lineitem side");
   });
}
// Step 2b (per shard): this shard's partsupp rows of green parts (ps_partkey, ps_suppkey,
// ps_supplycost), partitioned by the same hash-radix of the part key.
extern "C" int32_t ldb_plan_tpch_q9_partsupp_side(ldb_ctx* ctx, const ldb_table* greenkeys, const ldb_table* ps, int32_t world, ldb_table** result, int64_t* counts) {
   return guarded([&] {
      Rel g0(ctx), ps0(ctx), ps1(ctx);
      check(ldb_gpu_rel_from_table(ctx, greenkeys, &g0.r), "q9 green keys");
      check(ldb_gpu_rel_from_table(ctx, ps, &ps0.r), "q9 partsupp");
      Ht hg(ctx);
      ldb_colref gk{0, 0}, pspk{0, colOf(ps, "ps_partkey")};
      check(ldb_gpu_join_build(ctx, g0.r, &gk, 1, 1, &hg.h), "q9 build green");
      check(ldb_gpu_join_probe(ctx, hg.h, ps0.r, &pspk, 1, LDB_JOIN_SEMI, &ps1.r, nullptr), "q9 partsupp of green parts");
      ldb_colref cols[3] = {pspk, {0, colOf(ps, "ps_suppkey")}, {0, colOf(ps, "ps_supplycost")}};
      check(ldb_gpu_partition(ctx, ps1.r, &pspk, 1, world, cols, 3, result, counts), "q9 partition partsupp side");
   });
}
// Step 3 (per rank, after the two all-to-alls): received lineitem rows ⋈ received partsupp rows
// on (partkey, suppkey) ⋈ supplier (replicated) ⋈ nation, partial SUM per (n_name, o_year).
extern "C" int32_t ldb_plan_tpch_q9_join(ldb_ctx* ctx, const ldb_table* lrows, const ldb_table* psrows, const ldb_table* supp, const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel l0(ctx), ps0(ctx), s0(ctx), n0(ctx), lps(ctx), lpss(ctx), all(ctx);
      check(ldb_gpu_rel_from_table(ctx, lrows, &l0.r), "q9 lineitem rows");
      check(ldb_gpu_rel_from_table(ctx, psrows, &ps0.r), "q9 partsupp rows");
      check(ldb_gpu_rel_from_table(ctx, supp, &s0.r), "q9 supplier");
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q9 nation");
      Ht hps(ctx), hs(ctx), hn(ctx);
      ldb_colref psk2[2] = {{0, 0}, {0, 1}}, lk2[2] = {{0, 0}, {0, 1}}, lsk{0, 1};
      check(ldb_gpu_join_build(ctx, ps0.r, psk2, 2, 1, &hps.h), "q9 build partsupp");
      check(ldb_gpu_join_probe(ctx, hps.h, l0.r, lk2, 2, LDB_JOIN_INNER, &lps.r, nullptr), "q9 probe partsupp"); // sides: lrows, psrows
      ldb_colref sk{0, colOf(supp, "s_suppkey")};
      check(ldb_gpu_join_build(ctx, s0.r, &sk, 1, 1, &hs.h), "q9 build supplier");
      check(ldb_gpu_join_probe(ctx, hs.h, lps.r, &lsk, 1, LDB_JOIN_INNER, &lpss.r, nullptr), "q9 probe supplier"); // lrows, psrows, supplier
      (void) nat; // nation is joined after the merge (eager aggregation on the integer key, see ldb_plan_tpch_q9)
      ldb_colref ext{0, 3}, disc{0, 4}, qty{0, 5}, cost{1, 2};
      DecimalType te = decOf(lrows, 3), td = decOf(lrows, 4), tq = decOf(lrows, 5), tc = decOf(psrows, 2), t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, td, &t1md);
      DecimalType tRev = typeAfterMul(te, t1md), tCost = typeAfterMul(tc, tq), tAmount = higherDecimalType(tRev, tCost);
      ldb_expr amount;
      memset(&amount, 0, sizeof(amount));
      amount.n_terms = 2;
      amount.t[0].n_factors = 2;
      amount.t[0].f[0] = colFactor(ext);
      amount.t[0].f[1] = oneMinusDisc;
      amount.t[1].n_factors = 2;
      amount.t[1].negate = 1;
      amount.t[1].f[0] = colFactor(cost);
      amount.t[1].f[1] = colFactor(qty);
      ldb_agg_spec agg = sumDec(amount, tAmount);
      ldb_colref keys[2] = {{2, colOf(supp, "s_nationkey")}, {0, 2}};
      check(ldb_gpu_groupby(ctx, lpss.r, nullptr, 0, keys, 2, &agg, 1, 25 * 8, result), "q9 partial groupby");
   });
}
// Step 4 (replicated): the gathered partial sums (s_nationkey, o_year, sum) ⋈ nation, added up per
// (n_name, o_year), ordered.
extern "C" int32_t ldb_plan_tpch_q9_final(ldb_ctx* ctx, const ldb_table* partials, const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), n0(ctx), pn(ctx), g(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q9 final");
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q9 final nation");
      Ht hn(ctx);
      ldb_colref nk{0, colOf(nat, "n_nationkey")}, pnk{0, 0};
      check(ldb_gpu_join_build(ctx, n0.r, &nk, 1, 1, &hn.h), "q9 build nation");
      check(ldb_gpu_join_probe(ctx, hn.h, in.r, &pnk, 1, LDB_JOIN_INNER, &pn.r, nullptr), "q9 probe nation"); // sides: partials, nation
      ldb_colref keys[2] = {{1, colOf(nat, "n_name")}, {0, 1}};
      ldb_agg_spec agg = sumDec(product({colFactor({0, 2})}), decOf(partials, 2));
      Table grouped(ctx);
      check(ldb_gpu_groupby(ctx, pn.r, nullptr, 0, keys, 2, &agg, 1, 25 * 8, &grouped.t), "q9 final groupby");
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q9 final rel");
      ldb_sort_spec specs[2] = {{{0, 0}, 0, 0}, {{0, 1}, 1, 0}};
      check(ldb_gpu_sort(ctx, g.r, specs, 2, &sorted.r), "q9 final sort");
      ldb_colref outc[3] = {{0, 0}, {0, 1}, {0, 2}};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 3, result), "q9 final materialize");
   });
}

// ---------------------------------------------------------------- TPC-H Q5 (resources/sql/tpch/5.sql)
// Revenue per nation of one region from orders of one year where customer and supplier are of the
// same nation.  The plan is built from four pieces so that the multi-GPU run can exchange the two
// small reduced dimension tables between them (customers are sharded by rows, orders by ranges):
//   q5_customers / q5_suppliers: rows of the region's nations → (key, nationkey)
//   q5_local: orders of the year ⋈ those customers ⋈ lineitem ⋈ those suppliers on
//             (l_suppkey, c_nationkey) = (s_suppkey, s_nationkey) → SUM per nationkey
//   q5_final: (gathered) partial sums ⋈ nation → GROUP BY n_name, ORDER BY revenue DESC
namespace {
// rows of `t` whose `nationCol` is a nation of region 'ASIA', as a table (keyCol, nationCol)
void regionMembers(ldb_ctx* ctx, const ldb_table* t, const char* keyCol, const char* nationCol, const ldb_table* nat, const ldb_table* reg, ldb_table** result,
                   const char* regionName = "ASIA") {
   Rel r0(ctx), r1(ctx), n0(ctx), n1(ctx), t0(ctx), t1(ctx);
   check(ldb_gpu_rel_from_table(ctx, reg, &r0.r), "q5 region");
   check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q5 nation");
   check(ldb_gpu_rel_from_table(ctx, t, &t0.r), "q5 dimension");
   auto rr = Restrictions::create({{"r_name", FilterOp::EQ, std::string(regionName), {}}}, reg);
   check(ldb_gpu_scan_filter(ctx, r0.r, rr->data(), rr->size(), &r1.r), "q5 filter region");
   Ht hr(ctx), hn(ctx);
   ldb_colref rk{0, colOf(reg, "r_regionkey")}, nrk{0, colOf(nat, "n_regionkey")}, nk{0, colOf(nat, "n_nationkey")}, tn{0, colOf(t, nationCol)};
   check(ldb_gpu_join_build(ctx, r1.r, &rk, 1, 1, &hr.h), "q5 build region");
   check(ldb_gpu_join_probe(ctx, hr.h, n0.r, &nrk, 1, LDB_JOIN_SEMI, &n1.r, nullptr), "q5 nations of the region");
   check(ldb_gpu_join_build(ctx, n1.r, &nk, 1, 1, &hn.h), "q5 build nations");
   check(ldb_gpu_join_probe(ctx, hn.h, t0.r, &tn, 1, LDB_JOIN_SEMI, &t1.r, nullptr), "q5 rows of those nations");
   ldb_colref outc[2] = {{0, colOf(t, keyCol)}, tn};
   check(ldb_gpu_materialize(ctx, t1.r, outc, 2, result), "q5 materialize");
}
} // namespace
extern "C" int32_t ldb_plan_tpch_q5_customers(ldb_ctx* ctx, const ldb_table* cust, const ldb_table* nat, const ldb_table* reg, ldb_table** result) {
   return guarded([&] { regionMembers(ctx, cust, "c_custkey", "c_nationkey", nat, reg, result); });
}
extern "C" int32_t ldb_plan_tpch_q5_suppliers(ldb_ctx* ctx, const ldb_table* supp, const ldb_table* nat, const ldb_table* reg, ldb_table** result) {
   return guarded([&] { regionMembers(ctx, supp, "s_suppkey", "s_nationkey", nat, reg, result); });
}
extern "C" int32_t ldb_plan_tpch_q5_local(ldb_ctx* ctx, const ldb_table* custs, const ldb_table* supps, const ldb_table* ord, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel c0(ctx), s0(ctx), o0(ctx), o1(ctx), l0(ctx), oc(ctx), loc(ctx), locs(ctx);
      check(ldb_gpu_rel_from_table(ctx, custs, &c0.r), "q5 customers");
      check(ldb_gpu_rel_from_table(ctx, supps, &s0.r), "q5 suppliers");
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q5 orders");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q5 lineitem");
      auto ro = Restrictions::create({{"o_orderdate", FilterOp::GTE, std::string("1994-01-01"), {}}, {"o_orderdate", FilterOp::LT, std::string("1995-01-01"), {}}}, ord);
      check(ldb_gpu_scan_filter(ctx, o0.r, ro->data(), ro->size(), &o1.r), "q5 filter orders");
      Ht hc(ctx), ho(ctx), hs(ctx);
      ldb_colref ck{0, 0}, ock{0, colOf(ord, "o_custkey")}, ook{0, colOf(ord, "o_orderkey")}, lok{0, colOf(li, "l_orderkey")};
      check(ldb_gpu_join_build(ctx, c0.r, &ck, 1, 1, &hc.h), "q5 build customers");
      check(ldb_gpu_join_probe(ctx, hc.h, o1.r, &ock, 1, LDB_JOIN_INNER, &oc.r, nullptr), "q5 probe orders"); // sides: orders, customers
      check(ldb_gpu_join_build(ctx, oc.r, &ook, 1, 1, &ho.h), "q5 build orders");
      check(ldb_gpu_join_probe(ctx, ho.h, l0.r, &lok, 1, LDB_JOIN_INNER, &loc.r, nullptr), "q5 probe lineitem"); // sides: lineitem, orders, customers
      // supplier of the lineitem must be of the customer's nation: two-column key
      ldb_colref sk2[2] = {{0, 0}, {0, 1}}, lk2[2] = {{0, colOf(li, "l_suppkey")}, {2, 1}};
      check(ldb_gpu_join_build(ctx, s0.r, sk2, 2, 1, &hs.h), "q5 build suppliers");
      check(ldb_gpu_join_probe(ctx, hs.h, loc.r, lk2, 2, LDB_JOIN_SEMI, &locs.r, nullptr), "q5 semi join suppliers"); // (s_suppkey is a key: at most one partner)
      ldb_colref ext{0, colOf(li, "l_extendedprice")}, disc{0, colOf(li, "l_discount")};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(li, disc.col), &t1md);
      DecimalType tRev = typeAfterMul(decOf(li, ext.col), t1md);
      ldb_agg_spec agg = sumDec(product({colFactor(ext), oneMinusDisc}), tRev);
      ldb_colref key{2, 1}; // c_nationkey (= s_nationkey)
      check(ldb_gpu_groupby(ctx, locs.r, nullptr, 0, &key, 1, &agg, 1, 25, result), "q5 partial groupby");
   });
}
extern "C" int32_t ldb_plan_tpch_q5_final(ldb_ctx* ctx, const ldb_table* partials, const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), n0(ctx), pn(ctx), g(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q5 final");
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q5 final nation");
      Ht hn(ctx);
      ldb_colref nk{0, colOf(nat, "n_nationkey")}, pnk{0, 0};
      check(ldb_gpu_join_build(ctx, n0.r, &nk, 1, 1, &hn.h), "q5 build nation");
      check(ldb_gpu_join_probe(ctx, hn.h, in.r, &pnk, 1, LDB_JOIN_INNER, &pn.r, nullptr), "q5 probe nation"); // sides: partials, nation
      ldb_colref key{1, colOf(nat, "n_name")};
      ldb_agg_spec agg = sumDec(product({colFactor({0, 1})}), decOf(partials, 1));
      Table grouped(ctx);
      check(ldb_gpu_groupby(ctx, pn.r, nullptr, 0, &key, 1, &agg, 1, 25, &grouped.t), "q5 final groupby");
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q5 final rel");
      ldb_sort_spec spec{{0, 1}, 1, 0};
      check(ldb_gpu_sort(ctx, g.r, &spec, 1, &sorted.r), "q5 final sort");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 2, result), "q5 final materialize");
   });
}

// ---------------------------------------------------------------- TPC-H Q7 (resources/sql/tpch/7.sql)
// Trade volume between two nations per year.  (n1 = A and n2 = B) or (n1 = B and n2 = A) is
// evaluated as: both nations ∈ {A, B} (pushed into the two dimension tables) and n1 <> n2 (a residual
// column-vs-column conjunct).  Pieces as for Q5: the two reduced dimension tables can be
// all-gathered between `q7_members` and `q7_local`.
namespace {
void nationMembers(ldb_ctx* ctx, const ldb_table* t, const char* keyCol, const char* nationCol, const ldb_table* nat, ldb_table** result) {
   Rel n0(ctx), n1(ctx), t0(ctx), t1(ctx);
   check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q7 nation");
   check(ldb_gpu_rel_from_table(ctx, t, &t0.r), "q7 dimension");
   auto rn = Restrictions::create({{"n_name", FilterOp::IN, {}, std::vector<std::string>{"FRANCE", "GERMANY"}}}, nat);
   check(ldb_gpu_scan_filter(ctx, n0.r, rn->data(), rn->size(), &n1.r), "q7 filter nation");
   Ht hn(ctx);
   ldb_colref nk{0, colOf(nat, "n_nationkey")}, tn{0, colOf(t, nationCol)};
   check(ldb_gpu_join_build(ctx, n1.r, &nk, 1, 1, &hn.h), "q7 build nations");
   check(ldb_gpu_join_probe(ctx, hn.h, t0.r, &tn, 1, LDB_JOIN_SEMI, &t1.r, nullptr), "q7 rows of the two nations");
   ldb_colref outc[2] = {{0, colOf(t, keyCol)}, tn};
   check(ldb_gpu_materialize(ctx, t1.r, outc, 2, result), "q7 materialize");
}
} // namespace
extern "C" int32_t ldb_plan_tpch_q7_customers(ldb_ctx* ctx, const ldb_table* cust, const ldb_table* nat, ldb_table** result) {
   return guarded([&] { nationMembers(ctx, cust, "c_custkey", "c_nationkey", nat, result); });
}
extern "C" int32_t ldb_plan_tpch_q7_suppliers(ldb_ctx* ctx, const ldb_table* supp, const ldb_table* nat, ldb_table** result) {
   return guarded([&] { nationMembers(ctx, supp, "s_suppkey", "s_nationkey", nat, result); });
}
// partial result: (s_nationkey, c_nationkey, l_year, SUM(volume))
extern "C" int32_t ldb_plan_tpch_q7_local(ldb_ctx* ctx, const ldb_table* custs, const ldb_table* supps, const ldb_table* ord, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel c0(ctx), s0(ctx), o0(ctx), l0(ctx), l1(ctx), ls(ctx), m0(ctx), om(ctx), omc(ctx), diff(ctx), withYear(ctx);
      check(ldb_gpu_rel_from_table(ctx, custs, &c0.r), "q7 customers");
      check(ldb_gpu_rel_from_table(ctx, supps, &s0.r), "q7 suppliers");
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q7 orders");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q7 lineitem");
      auto rl = Restrictions::create({{"l_shipdate", FilterOp::GTE, std::string("1995-01-01"), {}}, {"l_shipdate", FilterOp::LTE, std::string("1996-12-31"), {}}}, li);
      check(ldb_gpu_scan_filter(ctx, l0.r, rl->data(), rl->size(), &l1.r), "q7 filter lineitem");
      Ht hs(ctx), hm(ctx), hc(ctx);
      ldb_colref sk{0, 0}, lsk{0, colOf(li, "l_suppkey")};
      check(ldb_gpu_join_build(ctx, s0.r, &sk, 1, 1, &hs.h), "q7 build suppliers");
      check(ldb_gpu_join_probe(ctx, hs.h, l1.r, &lsk, 1, LDB_JOIN_INNER, &ls.r, nullptr), "q7 probe lineitem"); // sides: lineitem, suppliers
      // narrow, then orders probe the reduced lineitem side (it is the smaller one)
      ldb_colref keep[5] = {{0, colOf(li, "l_orderkey")}, {0, colOf(li, "l_shipdate")}, {0, colOf(li, "l_extendedprice")}, {0, colOf(li, "l_discount")}, {1, 1}};
      Table m(ctx), years(ctx);
      check(ldb_gpu_materialize(ctx, ls.r, keep, 5, &m.t), "q7 materialize");
      check(ldb_gpu_rel_from_table(ctx, m.t, &m0.r), "q7 rel");
      ldb_colref mok{0, 0}, ook{0, colOf(ord, "o_orderkey")};
      check(ldb_gpu_join_build(ctx, m0.r, &mok, 1, 0, &hm.h), "q7 build reduced lineitem");
      check(ldb_gpu_join_probe(ctx, hm.h, o0.r, &ook, 1, LDB_JOIN_INNER, &om.r, nullptr), "q7 probe orders"); // sides: orders, m
      ldb_colref ck{0, 0}, ock{0, colOf(ord, "o_custkey")};
      check(ldb_gpu_join_build(ctx, c0.r, &ck, 1, 1, &hc.h), "q7 build customers");
      check(ldb_gpu_join_probe(ctx, hc.h, om.r, &ock, 1, LDB_JOIN_INNER, &omc.r, nullptr), "q7 probe customers"); // sides: orders, m, customers
      ldb_filter_desc differ = colCompare({1, 4}, FilterOp::NEQ, {2, 1}); // s_nationkey <> c_nationkey
      check(ldb_gpu_scan_filter(ctx, omc.r, &differ, 1, &diff.r), "q7 nations differ");
      check(ldb_gpu_map_column(ctx, diff.r, {1, 1}, LDB_FN_EXTRACT_YEAR, "l_year", &years.t), "q7 extract year");
      check(ldb_gpu_rel_zip(ctx, diff.r, years.t, &withYear.r), "q7 zip year");
      ldb_colref ext{1, 2}, disc{1, 3};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(m.t, 3), &t1md);
      DecimalType tVol = typeAfterMul(decOf(m.t, 2), t1md);
      ldb_agg_spec agg = sumDec(product({colFactor(ext), oneMinusDisc}), tVol);
      ldb_colref keys[3] = {{1, 4}, {2, 1}, {3, 0}};
      check(ldb_gpu_groupby(ctx, withYear.r, nullptr, 0, keys, 3, &agg, 1, 16, result), "q7 partial groupby");
   });
}
// (gathered) partials ⋈ nation (supplier side) ⋈ nation (customer side) → names, re-aggregated, ordered
extern "C" int32_t ldb_plan_tpch_q7_final(ldb_ctx* ctx, const ldb_table* partials, const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), n0(ctx), p1(ctx), p2(ctx), g(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q7 final");
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q7 final nation");
      Ht hn(ctx);
      ldb_colref nk{0, colOf(nat, "n_nationkey")}, snk{0, 0}, cnk{0, 1};
      check(ldb_gpu_join_build(ctx, n0.r, &nk, 1, 1, &hn.h), "q7 build nation");
      check(ldb_gpu_join_probe(ctx, hn.h, in.r, &snk, 1, LDB_JOIN_INNER, &p1.r, nullptr), "q7 supplier nation"); // sides: partials, nation(supp)
      check(ldb_gpu_join_probe(ctx, hn.h, p1.r, &cnk, 1, LDB_JOIN_INNER, &p2.r, nullptr), "q7 customer nation"); // sides: partials, nation(supp), nation(cust)
      const int32_t nn = colOf(nat, "n_name");
      ldb_colref keys[3] = {{1, nn}, {2, nn}, {0, 2}};
      ldb_agg_spec agg = sumDec(product({colFactor({0, 3})}), decOf(partials, 3));
      Table grouped(ctx);
      check(ldb_gpu_groupby(ctx, p2.r, nullptr, 0, keys, 3, &agg, 1, 16, &grouped.t), "q7 final groupby");
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q7 final rel");
      ldb_sort_spec specs[3] = {{{0, 0}, 0, 0}, {{0, 1}, 0, 0}, {{0, 2}, 0, 0}};
      check(ldb_gpu_sort(ctx, g.r, specs, 3, &sorted.r), "q7 final sort");
      ldb_colref outc[4] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 4, result), "q7 final materialize");
   });
}

// ---------------------------------------------------------------- TPC-H Q11 (resources/sql/tpch/11.sql)
// Stock value per part held by one nation's suppliers, HAVING value > 0.0001 × the total.
// ps_supplycost decimal(12,2) × ps_availqty (int32 → decimal(19,0), sql_analyzer.cpp:3125-3141);
// the scalar subquery is the SUM over the same groups, `sum × 0.0001` has scale 2+4, so after the
// cast to the common scale the comparison is value·10^4 > total in integers, i.e.
// value > floor(total / 10^4) — the constant of the HAVING filter.
// Pieces (the single-GPU plan is their composition): q11_suppliers → [all-gather] → q11_groups →
// [q11_partition + all-to-all on the hash of ps_partkey → q11_merge: a part's four partsupp rows
// may straddle two row shards] → q11_total → [all-gather] → q11_filter → [all-gather] → q11_sort.
namespace {
ldb_agg_spec sumOfCol(const ldb_table* t, int32_t col) { return sumDec(product({colFactor({0, col})}), decOf(t, col)); }
void groupByFirstCol(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result, const char* what) {
   Rel in(ctx);
   check(ldb_gpu_rel_from_table(ctx, rows, &in.r), what);
   ldb_colref key{0, 0};
   ldb_agg_spec agg = sumOfCol(rows, 1);
   check(ldb_gpu_groupby(ctx, in.r, nullptr, 0, &key, 1, &agg, 1, std::max<int64_t>(ldb_gpu_table_rows(rows), 16), result), what);
}
} // namespace
extern "C" int32_t ldb_plan_tpch_q11_suppliers(ldb_ctx* ctx, const ldb_table* supp, const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel n0(ctx), n1(ctx), s0(ctx), s1(ctx);
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q11 nation");
      check(ldb_gpu_rel_from_table(ctx, supp, &s0.r), "q11 supplier");
      auto rn = Restrictions::create({{"n_name", FilterOp::EQ, std::string("GERMANY"), {}}}, nat);
      check(ldb_gpu_scan_filter(ctx, n0.r, rn->data(), rn->size(), &n1.r), "q11 filter nation");
      Ht hn(ctx);
      ldb_colref nk{0, colOf(nat, "n_nationkey")}, sn{0, colOf(supp, "s_nationkey")}, sk{0, colOf(supp, "s_suppkey")};
      check(ldb_gpu_join_build(ctx, n1.r, &nk, 1, 1, &hn.h), "q11 build nation");
      check(ldb_gpu_join_probe(ctx, hn.h, s0.r, &sn, 1, LDB_JOIN_SEMI, &s1.r, nullptr), "q11 suppliers of the nation");
      check(ldb_gpu_materialize(ctx, s1.r, &sk, 1, result), "q11 materialize suppliers");
   });
}
// (ps_partkey, SUM(ps_supplycost * ps_availqty)) over this shard's partsupp rows of those suppliers
extern "C" int32_t ldb_plan_tpch_q11_groups(ldb_ctx* ctx, const ldb_table* suppkeys, const ldb_table* ps, ldb_table** result) {
   return guarded([&] {
      Rel s0(ctx), ps0(ctx), ps1(ctx);
      check(ldb_gpu_rel_from_table(ctx, suppkeys, &s0.r), "q11 supplier keys");
      check(ldb_gpu_rel_from_table(ctx, ps, &ps0.r), "q11 partsupp");
      Ht hs(ctx);
      ldb_colref sk{0, 0}, pssk{0, colOf(ps, "ps_suppkey")}, pspk{0, colOf(ps, "ps_partkey")};
      check(ldb_gpu_join_build(ctx, s0.r, &sk, 1, 1, &hs.h), "q11 build suppliers");
      check(ldb_gpu_join_probe(ctx, hs.h, ps0.r, &pssk, 1, LDB_JOIN_SEMI, &ps1.r, nullptr), "q11 partsupp of those suppliers");
      ldb_colref cost{0, colOf(ps, "ps_supplycost")}, qty{0, colOf(ps, "ps_availqty")};
      DecimalType tVal = typeAfterMul(decOf(ps, cost.col), {19, 0});
      ldb_agg_spec agg = sumDec(product({colFactor(cost), colFactor(qty)}), tVal);
      const int64_t expected = std::max<int64_t>(ldb_gpu_table_rows(ps) / 16, 1024); // one nation of 25, ≤ 4 rows per part
      check(ldb_gpu_groupby(ctx, ps1.r, nullptr, 0, &pspk, 1, &agg, 1, expected, result), "q11 groupby");
   });
}
extern "C" int32_t ldb_plan_tpch_q11_partition(ldb_ctx* ctx, const ldb_table* groups, int32_t world, ldb_table** result, int64_t* counts) {
   return guarded([&] {
      Rel g0(ctx);
      check(ldb_gpu_rel_from_table(ctx, groups, &g0.r), "q11 partial groups");
      ldb_colref cols[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_partition(ctx, g0.r, &cols[0], 1, world, cols, 2, result, counts), "q11 partition");
   });
}
extern "C" int32_t ldb_plan_tpch_q11_merge(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result) {
   return guarded([&] { groupByFirstCol(ctx, rows, result, "q11 merge"); });
}
extern "C" int32_t ldb_plan_tpch_q11_total(ldb_ctx* ctx, const ldb_table* groups, ldb_table** result) {
   return guarded([&] {
      Rel g0(ctx);
      check(ldb_gpu_rel_from_table(ctx, groups, &g0.r), "q11 total");
      ldb_agg_spec agg = sumOfCol(groups, 1);
      check(ldb_gpu_groupby(ctx, g0.r, nullptr, 0, nullptr, 0, &agg, 1, 1, result), "q11 total");
   });
}
// `totals`: one row per rank (the partial sums of the scalar subquery)
extern "C" int32_t ldb_plan_tpch_q11_filter(ldb_ctx* ctx, const ldb_table* groups, const ldb_table* totals, ldb_table** result) {
   return guarded([&] {
      const int64_t n = ldb_gpu_table_rows(totals);
      std::vector<__int128> parts((size_t) std::max<int64_t>(n, 1), 0);
      if (n) check(ldb_gpu_table_read_fixed(ctx, totals, 0, parts.data(), n * 16), "q11 read totals");
      __int128 total = 0;
      for (int64_t i = 0; i < n; i++) total += parts[(size_t) i];
      Rel g0(ctx), g1(ctx);
      check(ldb_gpu_rel_from_table(ctx, groups, &g0.r), "q11 groups");
      ldb_filter_desc having;
      memset(&having, 0, sizeof(having));
      having.col = {0, 1};
      having.op = (int32_t) FilterOp::GT;
      setInt(having, total / 10000);
      check(ldb_gpu_scan_filter(ctx, g0.r, &having, 1, &g1.r), "q11 having");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, g1.r, outc, 2, result), "q11 materialize");
   });
}
extern "C" int32_t ldb_plan_tpch_q11_sort(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, rows, &in.r), "q11 sort");
      ldb_sort_spec spec{{0, 1}, 1, 0};
      check(ldb_gpu_sort(ctx, in.r, &spec, 1, &sorted.r), "q11 sort");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 2, result), "q11 materialize sorted");
   });
}

// ---------------------------------------------------------------- TPC-H Q10 (resources/sql/tpch/10.sql)
// Returned-item revenue per customer of one quarter, top 20.  Pieces (the single-GPU plan is their
// composition; multi-GPU: orders/lineitem are co-located, a customer's orders are not, so the
// shard-local (o_custkey, revenue) groups are hash-radix partitioned on the key, exchanged and
// merged like Q11's groups before the top-20 is taken).  c_custkey = o_custkey is a foreign key
// and c_nationkey = n_nationkey too, so neither join drops a group: the aggregation runs on
// o_custkey before the customer join and only the 20 winners are joined with customer and nation
// (the other group-by columns are functionally dependent on c_custkey).  The generated customer
// table carries c_custkey, c_name, c_acctbal, c_nationkey; c_address / c_phone / c_comment are not
// generated and not returned.
// Step 1 (per shard): orders of the quarter ⋈ returned lineitems, SUM per o_custkey.
extern "C" int32_t ldb_plan_tpch_q10_local(ldb_ctx* ctx, const ldb_table* ord, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel o0(ctx), o1(ctx), l0(ctx), l1(ctx), lo(ctx);
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q10 orders");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q10 lineitem");
      auto ro = Restrictions::create({{"o_orderdate", FilterOp::GTE, std::string("1993-10-01"), {}}, {"o_orderdate", FilterOp::LT, std::string("1994-01-01"), {}}}, ord);
      auto rl = Restrictions::create({{"l_returnflag", FilterOp::EQ, std::string("R"), {}}}, li);
      check(ldb_gpu_scan_filter(ctx, o0.r, ro->data(), ro->size(), &o1.r), "q10 filter orders");
      check(ldb_gpu_scan_filter(ctx, l0.r, rl->data(), rl->size(), &l1.r), "q10 filter lineitem");
      Ht ho(ctx);
      ldb_colref ook{0, colOf(ord, "o_orderkey")}, lok{0, colOf(li, "l_orderkey")};
      check(ldb_gpu_join_build(ctx, o1.r, &ook, 1, 1, &ho.h), "q10 build orders");
      check(ldb_gpu_join_probe(ctx, ho.h, l1.r, &lok, 1, LDB_JOIN_INNER, &lo.r, nullptr), "q10 probe lineitem"); // sides: lineitem, orders
      ldb_colref ext{0, colOf(li, "l_extendedprice")}, disc{0, colOf(li, "l_discount")};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(li, disc.col), &t1md);
      DecimalType tRev = typeAfterMul(decOf(li, ext.col), t1md);
      ldb_agg_spec agg = sumDec(product({colFactor(ext), oneMinusDisc}), tRev);
      ldb_colref key{1, colOf(ord, "o_custkey")};
      const int64_t est = ldb_gpu_rel_rows(ctx, lo.r);
      check(ldb_gpu_groupby(ctx, lo.r, nullptr, 0, &key, 1, &agg, 1, est > 0 ? est : 1, result), "q10 groupby");
   });
}
// Multi-GPU only: route the shard-local groups to the rank that owns hash(o_custkey), re-aggregate.
extern "C" int32_t ldb_plan_tpch_q10_partition(ldb_ctx* ctx, const ldb_table* groups, int32_t world, ldb_table** result, int64_t* counts) {
   return ldb_plan_tpch_q11_partition(ctx, groups, world, result, counts);
}
extern "C" int32_t ldb_plan_tpch_q10_merge(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result) {
   return guarded([&] { groupByFirstCol(ctx, rows, result, "q10 merge"); });
}
// Step 2: the 20 groups with the highest revenue of a (custkey, revenue) table.
extern "C" int32_t ldb_plan_tpch_q10_top(ldb_ctx* ctx, const ldb_table* groups, ldb_table** result) {
   return guarded([&] {
      Rel g(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, groups, &g.r), "q10 top");
      ldb_sort_spec spec{{0, 1}, 1, 0};
      check(ldb_gpu_topk(ctx, g.r, &spec, 1, 20, &top.r), "q10 topk");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 2, result), "q10 top materialize");
   });
}
// Step 3 (per customer shard): c_name, c_acctbal and n_name of the winners whose customer lives here.
extern "C" int32_t ldb_plan_tpch_q10_names(ldb_ctx* ctx, const ldb_table* top20, const ldb_table* cust, const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel t0(ctx), c0(ctx), n0(ctx), ct(ctx), ctn(ctx);
      check(ldb_gpu_rel_from_table(ctx, top20, &t0.r), "q10 names");
      check(ldb_gpu_rel_from_table(ctx, cust, &c0.r), "q10 names customer");
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q10 names nation");
      Ht ht(ctx), hn(ctx);
      ldb_colref tk{0, 0}, ck{0, colOf(cust, "c_custkey")}, cn{0, colOf(cust, "c_nationkey")}, nk{0, colOf(nat, "n_nationkey")};
      check(ldb_gpu_join_build(ctx, t0.r, &tk, 1, 0, &ht.h), "q10 names build");
      check(ldb_gpu_join_probe(ctx, ht.h, c0.r, &ck, 1, LDB_JOIN_INNER, &ct.r, nullptr), "q10 names probe"); // sides: customer, top20
      check(ldb_gpu_join_build(ctx, n0.r, &nk, 1, 1, &hn.h), "q10 build nation");
      check(ldb_gpu_join_probe(ctx, hn.h, ct.r, &cn, 1, LDB_JOIN_INNER, &ctn.r, nullptr), "q10 probe nation"); // sides: customer, top20, nation
      ldb_colref outc[5] = {ck, {0, colOf(cust, "c_name")}, {1, 1}, {0, colOf(cust, "c_acctbal")}, {2, colOf(nat, "n_name")}};
      check(ldb_gpu_materialize(ctx, ctn.r, outc, 5, result), "q10 names materialize");
   });
}
// Step 4: ORDER BY revenue DESC LIMIT 20 over the named rows.
extern "C" int32_t ldb_plan_tpch_q10_final(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, rows, &in.r), "q10 final");
      ldb_sort_spec spec{{0, 2}, 1, 0};
      check(ldb_gpu_topk(ctx, in.r, &spec, 1, 20, &top.r), "q10 final topk");
      ldb_colref outc[5] = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 5, result), "q10 final materialize");
   });
}

// ---------------------------------------------------------------- TPC-H Q15 (resources/sql/tpch/15.sql)
// The revenue view (SUM per l_suppkey over one quarter's lineitems), its maximum (scalar subquery)
// and the suppliers that reach it.  Pieces: q15_local (shard-local groups; multi-GPU re-partitions
// and merges them on the key like Q10), q15_max (the best group: top-1 by revenue — MAX over a
// 128-bit decimal as an ordered select), q15_winners (groups whose revenue equals the maximum: the
// scalar is read back and becomes the constant of an EQ filter, as the reference materialises
// the subquery first), q15_final (⋈ supplier on the key, ORDER BY s_suppkey).  The generated
// supplier table carries s_suppkey, s_nationkey, s_acctbal; s_name / s_address / s_phone are not
// generated and not returned.
extern "C" int32_t ldb_plan_tpch_q15_local(ldb_ctx* ctx, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel l0(ctx), l1(ctx);
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q15 lineitem");
      auto rl = Restrictions::create({{"l_shipdate", FilterOp::GTE, std::string("1996-01-01"), {}}, {"l_shipdate", FilterOp::LT, std::string("1996-04-01"), {}}}, li);
      check(ldb_gpu_scan_filter(ctx, l0.r, rl->data(), rl->size(), &l1.r), "q15 filter lineitem");
      ldb_colref ext{0, colOf(li, "l_extendedprice")}, disc{0, colOf(li, "l_discount")}, key{0, colOf(li, "l_suppkey")};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(li, disc.col), &t1md);
      DecimalType tRev = typeAfterMul(decOf(li, ext.col), t1md);
      ldb_agg_spec agg = sumDec(product({colFactor(ext), oneMinusDisc}), tRev);
      const int64_t est = std::max<int64_t>(ldb_gpu_table_rows(li) / 512, 1024); // ≈ 600 lineitems per supplier
      check(ldb_gpu_groupby(ctx, l1.r, nullptr, 0, &key, 1, &agg, 1, est, result), "q15 groupby");
   });
}
extern "C" int32_t ldb_plan_tpch_q15_partition(ldb_ctx* ctx, const ldb_table* groups, int32_t world, ldb_table** result, int64_t* counts) {
   return ldb_plan_tpch_q11_partition(ctx, groups, world, result, counts);
}
extern "C" int32_t ldb_plan_tpch_q15_merge(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result) {
   return guarded([&] { groupByFirstCol(ctx, rows, result, "q15 merge"); });
}
extern "C" int32_t ldb_plan_tpch_q15_max(ldb_ctx* ctx, const ldb_table* groups, ldb_table** result) {
   return guarded([&] {
      Rel g(ctx), top(ctx);
      check(ldb_gpu_rel_from_table(ctx, groups, &g.r), "q15 max");
      ldb_sort_spec spec{{0, 1}, 1, 0};
      check(ldb_gpu_topk(ctx, g.r, &spec, 1, 1, &top.r), "q15 max topk");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, top.r, outc, 2, result), "q15 max materialize");
   });
}
extern "C" int32_t ldb_plan_tpch_q15_winners(ldb_ctx* ctx, const ldb_table* groups, const ldb_table* best, ldb_table** result) {
   return guarded([&] {
      Rel g0(ctx), g1(ctx);
      check(ldb_gpu_rel_from_table(ctx, groups, &g0.r), "q15 groups");
      ldb_filter_desc eq;
      memset(&eq, 0, sizeof(eq));
      eq.col = {0, 1};
      if (ldb_gpu_table_rows(best) > 0) {
         __int128 mx = 0;
         check(ldb_gpu_table_read_fixed(ctx, best, 1, &mx, 16), "q15 read max");
         eq.op = (int32_t) FilterOp::EQ;
         setInt(eq, mx);
      } else { // no lineitem in the quarter: MAX is NULL and `= NULL` keeps nothing
         eq.op = (int32_t) FilterOp::LT;
         eq.rhs_kind = LDB_RHS_COLUMN;
         eq.rhs_col = {0, 1};
      }
      check(ldb_gpu_scan_filter(ctx, g0.r, &eq, 1, &g1.r), "q15 filter");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, g1.r, outc, 2, result), "q15 materialize");
   });
}
extern "C" int32_t ldb_plan_tpch_q15_final(ldb_ctx* ctx, const ldb_table* winners, const ldb_table* supp, ldb_table** result) {
   return guarded([&] {
      Rel w0(ctx), s0(ctx), sw(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, winners, &w0.r), "q15 winners");
      check(ldb_gpu_rel_from_table(ctx, supp, &s0.r), "q15 supplier");
      Ht hw(ctx);
      ldb_colref wk{0, 0}, sk{0, colOf(supp, "s_suppkey")};
      check(ldb_gpu_join_build(ctx, w0.r, &wk, 1, 1, &hw.h), "q15 build winners");
      check(ldb_gpu_join_probe(ctx, hw.h, s0.r, &sk, 1, LDB_JOIN_INNER, &sw.r, nullptr), "q15 probe supplier"); // sides: supplier, winners
      Table joined(ctx);
      ldb_colref jc[2] = {sk, {1, 1}};
      check(ldb_gpu_materialize(ctx, sw.r, jc, 2, &joined.t), "q15 materialize");
      Rel j(ctx);
      check(ldb_gpu_rel_from_table(ctx, joined.t, &j.r), "q15 rel");
      ldb_sort_spec spec{{0, 0}, 0, 0};
      check(ldb_gpu_sort(ctx, j.r, &spec, 1, &sorted.r), "q15 sort");
      ldb_colref outc[2] = {{0, 0}, {0, 1}};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 2, result), "q15 final materialize");
   });
}

// ---------------------------------------------------------------- TPC-H Q14 (resources/sql/tpch/14.sql)
// 100.00 * sum(case when p_type like 'PROMO%' then rev else 0 end) / sum(rev) over one month of
// lineitem ⋈ part.  Pieces: q14_promo (keys of the PROMO parts: LIKE runs once per part, not per
// lineitem) → [all-gather] → q14_local (lineitem of the month ⋈ part keys; the CASE is a left outer
// join with the promo keys + a NOT NULL condition on its key) → [all-gather of the two partial
// sums] → q14_final (add, then the literal·sum/sum arithmetic of ldb_gpu_map_muldiv).
// Types: rev decimal(33,4) (as Q3); 100.00 is decimal(5,2); product decimal(38,6); quotient
// typeAfterDiv → decimal(38,6), scaled by 10^(6 + 4 − 6).
extern "C" int32_t ldb_plan_tpch_q14_promo(ldb_ctx* ctx, const ldb_table* part, ldb_table** result) {
   return guarded([&] {
      Rel p0(ctx), p1(ctx);
      check(ldb_gpu_rel_from_table(ctx, part, &p0.r), "q14 part");
      LikePred promo({0, colOf(part, "p_type")}, "PROMO%");
      check(ldb_gpu_scan_filter(ctx, p0.r, &promo.d, 1, &p1.r), "q14 filter part");
      ldb_colref key{0, colOf(part, "p_partkey")};
      check(ldb_gpu_materialize(ctx, p1.r, &key, 1, result), "q14 materialize promo keys");
   });
}
// `partkeys`: any table with a p_partkey column holding every part key (the part table itself, or
// its replicated key column); result: one row (SUM(promo rev), SUM(rev))
extern "C" int32_t ldb_plan_tpch_q14_local(ldb_ctx* ctx, const ldb_table* promokeys, const ldb_table* partkeys, const ldb_table* li, ldb_table** result) {
   return guarded([&] {
      Rel pr0(ctx), pk0(ctx), l0(ctx), l1(ctx), lp(ctx), lpp(ctx);
      check(ldb_gpu_rel_from_table(ctx, promokeys, &pr0.r), "q14 promo keys");
      check(ldb_gpu_rel_from_table(ctx, partkeys, &pk0.r), "q14 part keys");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q14 lineitem");
      auto rl = Restrictions::create({{"l_shipdate", FilterOp::GTE, std::string("1995-09-01"), {}}, {"l_shipdate", FilterOp::LT, std::string("1995-10-01"), {}}}, li);
      check(ldb_gpu_scan_filter(ctx, l0.r, rl->data(), rl->size(), &l1.r), "q14 filter lineitem");
      Ht hp(ctx), hpromo(ctx);
      ldb_colref pk{0, colOf(partkeys, "p_partkey")}, prk{0, 0}, lpk{0, colOf(li, "l_partkey")};
      check(ldb_gpu_join_build(ctx, pk0.r, &pk, 1, 1, &hp.h), "q14 build part");
      check(ldb_gpu_join_probe(ctx, hp.h, l1.r, &lpk, 1, LDB_JOIN_SEMI, &lp.r, nullptr), "q14 lineitem with a part"); // p_partkey is a key: the join adds no rows
      check(ldb_gpu_join_build(ctx, pr0.r, &prk, 1, 1, &hpromo.h), "q14 build promo keys");
      check(ldb_gpu_join_probe(ctx, hpromo.h, lp.r, &lpk, 1, LDB_JOIN_LEFT_OUTER, &lpp.r, nullptr), "q14 mark promo lines"); // sides: lineitem, promo keys
      ldb_colref ext{0, colOf(li, "l_extendedprice")}, disc{0, colOf(li, "l_discount")};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(li, disc.col), &t1md);
      DecimalType tRev = typeAfterMul(decOf(li, ext.col), t1md);
      ldb_agg_spec aggs[2] = {sumDec(product({colFactor(ext), oneMinusDisc}), tRev), sumDec(product({colFactor(ext), oneMinusDisc}), tRev)};
      aggs[0].n_preds = 1;
      memset(&aggs[0].preds[0], 0, sizeof(ldb_filter_desc));
      aggs[0].preds[0].col = {1, 0};
      aggs[0].preds[0].op = LDB_F_NOTNULL;
      check(ldb_gpu_groupby(ctx, lpp.r, nullptr, 0, nullptr, 0, aggs, 2, 1, result), "q14 partial sums");
   });
}
extern "C" int32_t ldb_plan_tpch_q14_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), s0(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q14 final");
      ldb_agg_spec aggs[2] = {sumOfCol(partials, 0), sumOfCol(partials, 1)};
      Table sums(ctx);
      check(ldb_gpu_groupby(ctx, in.r, nullptr, 0, nullptr, 0, aggs, 2, 1, &sums.t), "q14 add partials");
      check(ldb_gpu_rel_from_table(ctx, sums.t, &s0.r), "q14 sums");
      const DecimalType tSum = decOf(partials, 0), lit{5, 2}; // 100.00
      const DecimalType tMul = typeAfterMul(lit, tSum), tDiv = typeAfterDiv(tMul, tSum);
      check(ldb_gpu_map_muldiv(ctx, s0.r, {0, 0}, 10000, 0, lit.s + tSum.s - tMul.s, tDiv.s + tSum.s - tMul.s, {0, 1}, tDiv.p, tDiv.s, "promo_revenue", result), "q14 ratio");
   });
}

// ---------------------------------------------------------------- TPC-H Q8 (resources/sql/tpch/8.sql)
// Market share of one nation's suppliers within a region, per order year, for one part type:
// sum(case when n2.n_name = 'BRAZIL' then volume else 0 end) / sum(volume).  The part-type filter
// keeps 1/150 of part, so lineitem is reduced by it first; orders (two years) probe the reduced
// lineitem side; customers of the region are a semi join.  Pieces: q8_parts, q8_customers →
// [all-gather] → q8_local (partial sums per year) → [all-gather] → q8_final (add, divide, order).
extern "C" int32_t ldb_plan_tpch_q8_parts(ldb_ctx* ctx, const ldb_table* part, ldb_table** result) {
   return guarded([&] {
      Rel p0(ctx), p1(ctx);
      check(ldb_gpu_rel_from_table(ctx, part, &p0.r), "q8 part");
      auto rp = Restrictions::create({{"p_type", FilterOp::EQ, std::string("ECONOMY ANODIZED STEEL"), {}}}, part);
      check(ldb_gpu_scan_filter(ctx, p0.r, rp->data(), rp->size(), &p1.r), "q8 filter part");
      ldb_colref key{0, colOf(part, "p_partkey")};
      check(ldb_gpu_materialize(ctx, p1.r, &key, 1, result), "q8 materialize part keys");
   });
}
extern "C" int32_t ldb_plan_tpch_q8_customers(ldb_ctx* ctx, const ldb_table* cust, const ldb_table* nat, const ldb_table* reg, ldb_table** result) {
   return guarded([&] { regionMembers(ctx, cust, "c_custkey", "c_nationkey", nat, reg, result, "AMERICA"); });
}
// partial result: (o_year, SUM(volume of the nation's suppliers), SUM(volume))
extern "C" int32_t ldb_plan_tpch_q8_local(ldb_ctx* ctx, const ldb_table* partkeys, const ldb_table* custs, const ldb_table* supp, const ldb_table* ord, const ldb_table* li,
                                          const ldb_table* nat, ldb_table** result) {
   return guarded([&] {
      Rel pk0(ctx), c0(ctx), s0(ctx), o0(ctx), o1(ctx), l0(ctx), n0(ctx), lp(ctx), ls(ctx), m0(ctx), om(ctx), omc(ctx), omn(ctx), withYear(ctx);
      check(ldb_gpu_rel_from_table(ctx, partkeys, &pk0.r), "q8 part keys");
      check(ldb_gpu_rel_from_table(ctx, custs, &c0.r), "q8 customers");
      check(ldb_gpu_rel_from_table(ctx, supp, &s0.r), "q8 supplier");
      check(ldb_gpu_rel_from_table(ctx, ord, &o0.r), "q8 orders");
      check(ldb_gpu_rel_from_table(ctx, li, &l0.r), "q8 lineitem");
      check(ldb_gpu_rel_from_table(ctx, nat, &n0.r), "q8 nation");
      Ht hp(ctx), hs(ctx), hm(ctx), hc(ctx), hn(ctx);
      ldb_colref k0{0, 0}, lpk{0, colOf(li, "l_partkey")}, lsk{0, colOf(li, "l_suppkey")}, sk{0, colOf(supp, "s_suppkey")};
      check(ldb_gpu_join_build(ctx, pk0.r, &k0, 1, 1, &hp.h), "q8 build part keys");
      check(ldb_gpu_join_probe(ctx, hp.h, l0.r, &lpk, 1, LDB_JOIN_SEMI, &lp.r, nullptr), "q8 lineitem of the part type");
      check(ldb_gpu_join_build(ctx, s0.r, &sk, 1, 1, &hs.h), "q8 build supplier");
      check(ldb_gpu_join_probe(ctx, hs.h, lp.r, &lsk, 1, LDB_JOIN_INNER, &ls.r, nullptr), "q8 probe supplier"); // sides: lineitem, supplier
      ldb_colref keep[4] = {{0, colOf(li, "l_orderkey")}, {0, colOf(li, "l_extendedprice")}, {0, colOf(li, "l_discount")}, {1, colOf(supp, "s_nationkey")}};
      Table m(ctx), years(ctx);
      check(ldb_gpu_materialize(ctx, ls.r, keep, 4, &m.t), "q8 materialize");
      check(ldb_gpu_rel_from_table(ctx, m.t, &m0.r), "q8 rel");
      auto ro = Restrictions::create({{"o_orderdate", FilterOp::GTE, std::string("1995-01-01"), {}}, {"o_orderdate", FilterOp::LTE, std::string("1996-12-31"), {}}}, ord);
      check(ldb_gpu_scan_filter(ctx, o0.r, ro->data(), ro->size(), &o1.r), "q8 filter orders");
      ldb_colref ook{0, colOf(ord, "o_orderkey")}, ock{0, colOf(ord, "o_custkey")}, msn{1, 3}, nk{0, colOf(nat, "n_nationkey")};
      check(ldb_gpu_join_build(ctx, m0.r, &k0, 1, 0, &hm.h), "q8 build reduced lineitem");
      check(ldb_gpu_join_probe(ctx, hm.h, o1.r, &ook, 1, LDB_JOIN_INNER, &om.r, nullptr), "q8 probe orders"); // sides: orders, m
      check(ldb_gpu_join_build(ctx, c0.r, &k0, 1, 1, &hc.h), "q8 build customers");
      check(ldb_gpu_join_probe(ctx, hc.h, om.r, &ock, 1, LDB_JOIN_SEMI, &omc.r, nullptr), "q8 customers of the region");
      check(ldb_gpu_join_build(ctx, n0.r, &nk, 1, 1, &hn.h), "q8 build nation");
      check(ldb_gpu_join_probe(ctx, hn.h, omc.r, &msn, 1, LDB_JOIN_INNER, &omn.r, nullptr), "q8 supplier nation"); // sides: orders, m, nation
      check(ldb_gpu_map_column(ctx, omn.r, {0, colOf(ord, "o_orderdate")}, LDB_FN_EXTRACT_YEAR, "o_year", &years.t), "q8 extract year");
      check(ldb_gpu_rel_zip(ctx, omn.r, years.t, &withYear.r), "q8 zip year"); // sides: orders, m, nation, year
      ldb_colref ext{1, 1}, disc{1, 2};
      DecimalType t1md;
      ldb_factor oneMinusDisc = constPlusCol(1, -1, disc, decOf(m.t, 2), &t1md);
      DecimalType tVol = typeAfterMul(decOf(m.t, 1), t1md);
      ldb_agg_spec aggs[2] = {sumDec(product({colFactor(ext), oneMinusDisc}), tVol), sumDec(product({colFactor(ext), oneMinusDisc}), tVol)};
      auto brazil = Restrictions::create({{"n_name", FilterOp::EQ, std::string("BRAZIL"), {}}}, nat, 2);
      aggs[0].n_preds = 1;
      aggs[0].preds[0] = brazil->data()[0];
      ldb_colref key{3, 0};
      check(ldb_gpu_groupby(ctx, withYear.r, nullptr, 0, &key, 1, aggs, 2, 8, result), "q8 partial groupby");
   });
}
extern "C" int32_t ldb_plan_tpch_q8_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result) {
   return guarded([&] {
      Rel in(ctx), g(ctx), gz(ctx), sorted(ctx);
      check(ldb_gpu_rel_from_table(ctx, partials, &in.r), "q8 final");
      ldb_colref key{0, 0};
      ldb_agg_spec aggs[2] = {sumOfCol(partials, 1), sumOfCol(partials, 2)};
      Table grouped(ctx), share(ctx);
      check(ldb_gpu_groupby(ctx, in.r, nullptr, 0, &key, 1, aggs, 2, 8, &grouped.t), "q8 add partials");
      check(ldb_gpu_rel_from_table(ctx, grouped.t, &g.r), "q8 sums");
      const DecimalType tSum = decOf(partials, 1), tDiv = typeAfterDiv(tSum, tSum);
      check(ldb_gpu_map_muldiv(ctx, g.r, {0, 1}, 1, 0, 0, tDiv.s + tSum.s - tSum.s, {0, 2}, tDiv.p, tDiv.s, "mkt_share", &share.t), "q8 ratio");
      check(ldb_gpu_rel_zip(ctx, g.r, share.t, &gz.r), "q8 zip ratio");
      ldb_sort_spec spec{{0, 0}, 0, 0};
      check(ldb_gpu_sort(ctx, gz.r, &spec, 1, &sorted.r), "q8 sort");
      ldb_colref outc[2] = {{0, 0}, {1, 0}};
      check(ldb_gpu_materialize(ctx, sorted.r, outc, 2, result), "q8 materialize");
   });
}

// ---------------------------------------------------------------- C hooks for the host-logic tests
extern "C" int32_t ldb_host_parse_date32(const char* s, int32_t* out) {
   return guarded([&] { *out = parseDate32(s); });
}
extern "C" int32_t ldb_host_parse_decimal(const char* s, int32_t scale, int64_t* lo, int64_t* hi) {
   return guarded([&] {
      __int128 v = parseDecimal(s, scale);
      *lo = (int64_t) (uint64_t) v;
      *hi = (int64_t) (v >> 64);
   });
}
// op: 0 = mul, 1 = div, 2 = add/sub/compare common type, 3 = avg
extern "C" void ldb_host_decimal_type(int32_t op, int32_t p1, int32_t s1, int32_t p2, int32_t s2, int32_t* p, int32_t* s) {
   DecimalType r = op == 0 ? typeAfterMul({p1, s1}, {p2, s2}) : op == 1 ? typeAfterDiv({p1, s1}, {p2, s2}) : op == 2 ? higherDecimalType({p1, s1}, {p2, s2}) : avgType({p1, s1});
   *p = r.p;
   *s = r.s;
}
