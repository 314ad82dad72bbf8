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
// Every plan — single-GPU and sharded — is DATA: lingo-db_amd/plans/tpch/*.json and plans/tpch/dist/*.json,
// interpreted by ldb_plan.cpp (ldb_plan_run_json / ldb_plan_run_json_comm).  No query-specific C++ is left here.
using namespace lingodb::runtime::gpu;

namespace {
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
