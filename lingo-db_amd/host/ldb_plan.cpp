// ldb_plan.cpp — plans as DATA: a JSON list of execution steps interpreted over the C-ABI.
//
// LingoDB hands a query to its backend as `subop.execution_step`s (one per pipeline,
// src/compiler/Dialect/SubOperator/Transforms/OrganizeExecutionStepsPass.cpp; walked by
// handleExecutionStepCPU, SubOpToControlFlow.cpp:4363-4395) and can dump them as JSON
// (tools/ct/mlir-subop-to-json.cpp: {"type":"execution_step","subops":[{"subop":"scan", …}]}).
// This file is the receiving end a GPU `ExecutionBackend` needs: a step list with the same
// vocabulary, at the granularity of the C-ABI (scan + filter, lookup + scan_list = join_probe,
// reduce + merge = groupby, …), each step's callbacks replaced by declarative descriptors.  The
// plan text is data — the TPC-H plans under lingo-db_amd/plans/ are JSON files, not C++ functions;
// INTEGRATION.md shows the LingoDB-side emitter that would produce the same documents.
//
// Column references are NAMES, resolved against the tables a relation was built from; constants
// are SQL literals typed against the column by the mirror of Restrictions::create; scalar
// expressions are trees typed by the reference's decimal rules (sql_analyzer.cpp:3058-3159) and
// compiled either into the aggregate normal form (ldb_expr) or into a postfix program (ldb_xinstr).
#include "ldb_host.hpp"
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>

using namespace lingodb::runtime::gpu;

namespace {

} // namespace
#include "ldb_json.hpp"
using ldbjson::J;
using ldbjson::JParser;
namespace {

// ================================================================== environment
struct Value {
   enum Kind { TABLE, REL, HT } kind = TABLE;
   const ldb_table* table = nullptr;
   ldb_rel* rel = nullptr;
   ldb_hashtable* ht = nullptr;
   bool owned = false;
   std::vector<const ldb_table*> sides; // REL: the table behind each side; HT: the sides of its build relation
   ldb_rel* lazyRel = nullptr; // TABLE used as a relation
};

struct Scalar { // expression type: integer (p = 0) or decimal(p, s), date, bool
   enum Kind { INT, DEC, DATE, BOOL } kind = INT;
   int32_t p = 0, s = 0;
};

struct Interp {
   ldb_ctx* ctx;
   ldb_comm* comm = nullptr; // exchange steps: NULL = one rank (allgather is a copy, shuffle a materialize)
   std::map<std::string, Value> env;
   std::vector<ldb_table*> hidden; // computed-column tables that live as long as the plan run
   std::vector<std::unique_ptr<Restrictions>> keepRestr; // constant storage the descriptors point into
   std::vector<std::unique_ptr<std::string>> keepStr;
   std::string result;
   // a prepared plan remembers how many groups each group-by WITHOUT an estimate produced (by its `out` name): the next execution sizes the
   // table from that instead of paying an overflowing attempt again (plans translated from the reference's dumps carry no cardinalities)
   std::map<std::string, int64_t>* groupsSeen = nullptr;
   explicit Interp(ldb_ctx* c) : ctx(c) {}
   ~Interp() {
      // hash tables first (they reference relations), then relations, then tables
      for (auto& kv : env)
         if (kv.second.kind == Value::HT && kv.second.owned && kv.second.ht) ldb_gpu_hashtable_release(ctx, kv.second.ht);
      for (auto& kv : env) {
         if (kv.second.lazyRel) ldb_gpu_rel_release(ctx, kv.second.lazyRel);
         if (kv.second.kind == Value::REL && kv.second.owned && kv.second.rel) ldb_gpu_rel_release(ctx, kv.second.rel);
      }
      for (auto& kv : env)
         if (kv.second.kind == Value::TABLE && kv.second.owned && kv.second.table && kv.first != result) ldb_gpu_table_release(ctx, const_cast<ldb_table*>(kv.second.table));
      for (auto* t : hidden) ldb_gpu_table_release(ctx, t);
   }

   Value& val(const std::string& name) {
      auto it = env.find(name);
      if (it == env.end()) throw std::runtime_error("plan: unknown value '" + name + "'");
      return it->second;
   }
   // a value as a relation (tables get an identity relation on first use)
   ldb_rel* relOf(const std::string& name, std::vector<const ldb_table*>** sides = nullptr) {
      Value& v = val(name);
      if (v.kind == Value::HT) throw std::runtime_error("plan: '" + name + "' is a hash table, a relation is expected");
      if (v.kind == Value::TABLE) {
         if (!v.lazyRel) check(ldb_gpu_rel_from_table(ctx, v.table, &v.lazyRel), "scan");
         if (v.sides.empty()) v.sides = {v.table};
         if (sides) *sides = &v.sides;
         return v.lazyRel;
      }
      if (sides) *sides = &v.sides;
      return v.rel;
   }
   int64_t rowsOf(const std::string& name) {
      Value& v = val(name);
      if (v.kind == Value::TABLE) return ldb_gpu_table_rows(v.table);
      if (v.kind == Value::REL) return ldb_gpu_rel_rows(ctx, v.rel);
      throw std::runtime_error("plan: '" + name + "' has no row count");
   }
   void put(const std::string& name, Value v) {
      if (env.count(name)) throw std::runtime_error("plan: value '" + name + "' defined twice");
      env.emplace(name, std::move(v));
   }
   void putRel(const std::string& name, ldb_rel* r, std::vector<const ldb_table*> sides) {
      Value v;
      v.kind = Value::REL;
      v.rel = r;
      v.owned = true;
      v.sides = std::move(sides);
      put(name, std::move(v));
   }
   void putTable(const std::string& name, ldb_table* t) {
      Value v;
      v.kind = Value::TABLE;
      v.table = t;
      v.owned = true;
      put(name, std::move(v));
   }

   // "name" or "side:name" → column reference inside a relation with the given side tables
   static ldb_colref resolve(const std::vector<const ldb_table*>& sides, const std::string& ref, const char* what) {
      const auto colon = ref.find(':');
      if (colon != std::string::npos && colon > 0 && isdigit((unsigned char) ref[0])) {
         const int side = std::stoi(ref.substr(0, colon));
         if (side < 0 || (size_t) side >= sides.size()) throw std::runtime_error(std::string(what) + ": side out of range in '" + ref + "'");
         const int32_t c = ldb_gpu_table_col_index(sides[(size_t) side], ref.substr(colon + 1).c_str());
         if (c < 0) throw std::runtime_error(std::string(what) + ": no column '" + ref + "'");
         return {side, c};
      }
      for (size_t s = 0; s < sides.size(); s++) {
         const int32_t c = ldb_gpu_table_col_index(sides[s], ref.c_str());
         if (c >= 0) return {(int32_t) s, c};
      }
      throw std::runtime_error(std::string(what) + ": no column '" + ref + "' in the relation");
   }
   static ldb_coltype typeOf(const std::vector<const ldb_table*>& sides, ldb_colref c) {
      ldb_coltype t;
      ldb_gpu_table_coltype(sides[(size_t) c.side], c.col, &t);
      return t;
   }

   // ------------------------------------------------------------ predicates
   static int32_t opOf(const std::string& op) {
      static const std::map<std::string, int32_t> ops = {{"EQ", LDB_F_EQ},   {"NEQ", LDB_F_NEQ}, {"LT", LDB_F_LT}, {"LTE", LDB_F_LTE},   {"GT", LDB_F_GT},
                                                         {"GTE", LDB_F_GTE}, {"NOTNULL", LDB_F_NOTNULL}, {"IN", LDB_F_IN}, {"LIKE", LDB_F_LIKE}, {"NOT LIKE", LDB_F_NOT_LIKE}};
      auto it = ops.find(op);
      if (it == ops.end()) throw std::runtime_error("plan: unknown comparison '" + op + "'");
      return it->second;
   }
   // value of row 0 of a 1-row table's column as a 128-bit integer (scalar subquery results)
   __int128 readScalar(const std::string& table, const std::string& col) {
      Value& v = val(table);
      if (v.kind != Value::TABLE) throw std::runtime_error("plan: scalar source '" + table + "' must be a table");
      if (ldb_gpu_table_rows(v.table) < 1) throw std::runtime_error("plan: scalar source '" + table + "' is empty");
      const int32_t c = ldb_gpu_table_col_index(v.table, col.c_str());
      if (c < 0) throw std::runtime_error("plan: scalar source has no column '" + col + "'");
      const int32_t w = ldb_gpu_table_col_width(v.table, c);
      const int64_t rows = ldb_gpu_table_rows(v.table);
      std::vector<uint8_t> buf((size_t) (rows * w));
      check(ldb_gpu_table_read_fixed(ctx, v.table, c, buf.data(), (int64_t) buf.size()), "scalar read");
      __int128 x = 0;
      if (w == 16) memcpy(&x, buf.data(), 16);
      else if (w == 8) {
         int64_t t;
         memcpy(&t, buf.data(), 8);
         x = t;
      } else if (w == 4) {
         int32_t t;
         memcpy(&t, buf.data(), 4);
         x = t;
      } else if (w == 2) {
         int16_t t;
         memcpy(&t, buf.data(), 2);
         x = t;
      } else if (w == 1) { // int8 / bool
         x = (int8_t) buf[0];
      } else {
         throw std::runtime_error("plan: scalar source column of width " + std::to_string(w));
      }
      return x;
   }
   // row 0 of the scalar source is NULL (SimpleState over an empty input yields one NULL row)
   bool scalarIsNull(const std::string& table, const std::string& col) {
      Value& v = val(table);
      if (v.kind != Value::TABLE) throw std::runtime_error("plan: scalar source '" + table + "' must be a table");
      const int32_t c = ldb_gpu_table_col_index(v.table, col.c_str());
      if (c < 0) throw std::runtime_error("plan: scalar source has no column '" + col + "'");
      int32_t valid = 1;
      check(ldb_gpu_table_row_valid(ctx, v.table, c, 0, &valid), "scalar validity");
      return valid == 0;
   }
   static __int128 floorDiv(__int128 a, __int128 b) {
      __int128 q = a / b;
      if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
      return q;
   }
   ldb_filter_desc pred(const std::vector<const ldb_table*>& sides, const J& jp) {
      const std::string& opName = jp.s("op");
      const int32_t op = opOf(opName);
      const ldb_colref c = resolve(sides, jp.s("col"), "filter");
      const ldb_table* tab = sides[(size_t) c.side];
      const std::string colName = ldb_gpu_table_col_name(tab, c.col);
      if (const J* rc = jp.get("rhs_col")) { // residual column-vs-column conjunct (generated db.compare in the reference)
         ldb_filter_desc d;
         memset(&d, 0, sizeof(d));
         d.col = c;
         d.op = op;
         d.rhs_kind = LDB_RHS_COLUMN;
         d.rhs_col = resolve(sides, rc->str, "filter rhs");
         return d;
      }
      if (op == LDB_F_LIKE || op == LDB_F_NOT_LIKE) { // StringRuntime::like, evaluated in generated code
         keepStr.push_back(std::make_unique<std::string>(jp.s("value")));
         ldb_filter_desc d;
         memset(&d, 0, sizeof(d));
         d.col = c;
         d.op = op;
         d.rhs_kind = LDB_RHS_STRING;
         d.str = keepStr.back()->data();
         d.str_len = (int32_t) keepStr.back()->size();
         return d;
      }
      if (const J* sc = jp.get("scalar")) { // column OP (scalar subquery result): the constant is read back from the device
         if (rowsOf(sc->s("from")) < 1 || scalarIsNull(sc->s("from"), sc->s("col"))) { // no row, or a NULL (a key-less aggregate over no rows): a comparison with NULL keeps nothing
            ldb_filter_desc d;
            memset(&d, 0, sizeof(d));
            d.col = c;
            d.op = LDB_F_LT;
            d.rhs_kind = LDB_RHS_COLUMN;
            d.rhs_col = c;
            return d;
         }
         __int128 x = readScalar(sc->s("from"), sc->s("col"));
         // the two sides of the comparison are cast to a common decimal type first; with
         // x at scale sx and the column at scale sc <= sx:  col * 10^k OP x  ⇔  col OP' floor-ish(x / 10^k)
         const int64_t k = sc->iOr("div_pow10", 0);
         if (k < 0 || k > 38) throw std::runtime_error("plan: div_pow10 must be in [0, 38]");
         if (k > 0) {
            __int128 m = 1;
            for (int64_t i = 0; i < k; i++) m *= 10;
            const __int128 fl = floorDiv(x, m);
            const bool exact = fl * m == x;
            // col*m > x ⇔ col > floor(x/m);  col*m >= x ⇔ col > floor(x/m) unless exact;  col*m < x ⇔ col <= floor unless exact …
            ldb_filter_desc d;
            memset(&d, 0, sizeof(d));
            d.col = c;
            d.rhs_kind = LDB_RHS_INT;
            int32_t o = op;
            __int128 v = fl;
            switch (op) {
               case LDB_F_GT: o = LDB_F_GT; break;
               case LDB_F_GTE: o = exact ? LDB_F_GTE : LDB_F_GT; break;
               case LDB_F_LT: o = exact ? LDB_F_LT : LDB_F_LTE; break;
               case LDB_F_LTE: o = LDB_F_LTE; break;
               case LDB_F_EQ:
                  if (!exact) { // never true: compare with an impossible pair
                     o = LDB_F_LT;
                     v = ((__int128) 1) << 126;
                     v = -v;
                  }
                  break;
               default: throw std::runtime_error("plan: scalar comparison operator not supported");
            }
            d.op = o;
            d.value_lo = (uint64_t) v;
            d.value_hi = (int64_t) (v >> 64);
            return d;
         }
         ldb_filter_desc d;
         memset(&d, 0, sizeof(d));
         d.col = c;
         d.op = op;
         d.rhs_kind = LDB_RHS_INT;
         d.value_lo = (uint64_t) x;
         d.value_hi = (int64_t) (x >> 64);
         return d;
      }
      // column vs constant: typed by the mirror of Restrictions::create (Restrictions.cpp:392-521)
      FilterDescription fd;
      fd.columnName = colName;
      fd.op = (FilterOp) op;
      if (op == LDB_F_IN) {
         const J& vals = jp.at("values");
         if (vals.kind != J::ARR || vals.arr.empty()) throw std::runtime_error("plan: IN needs a non-empty 'values' array");
         if (vals.arr[0].kind == J::STR) {
            std::vector<std::string> v;
            for (auto& e : vals.arr) v.push_back(e.str);
            fd.values = v;
         } else {
            std::vector<int64_t> v;
            for (auto& e : vals.arr) v.push_back(e.inum);
            fd.values = v;
         }
      } else if (op != LDB_F_NOTNULL) {
         const J& v = jp.at("value");
         if (v.kind == J::STR) fd.value = v.str;
         else if (v.kind == J::NUM && v.isInt) fd.value = v.inum;
         else if (v.kind == J::NUM) fd.value = v.num;
         else throw std::runtime_error("plan: filter constant must be a string or a number");
      }
      keepRestr.push_back(Restrictions::create({fd}, tab, c.side));
      return keepRestr.back()->data()[0];
   }
   static constexpr int kMaxInConstants = 8, kMaxConjuncts = 8, kMaxResidual = 2; // what one scan_filter call takes (LDB_MAX_IN / LDB_MAX_PREDS of the kernels' descriptors)
   // rows of `cur` whose column d.col is one of d's constants: cur ⋉ (table of the constants).  The table, its relation and the hash table live as hidden
   // values until the plan run ends
   ldb_rel* semiJoinConstants(ldb_rel* cur, const ldb_filter_desc& d, const std::vector<const ldb_table*>& sides, const std::string& stem) {
      ldb_coltype ct;
      check(ldb_gpu_table_coltype(sides[(size_t) d.col.side], d.col.col, &ct), "IN list (column type)");
      const size_t w = ct.type == LDB_T_INT32 || ct.type == LDB_T_DATE32 || ct.type == LDB_T_CHAR4 ? 4 : ct.type == LDB_T_INT64 ? 8 : 0;
      if (!w) throw std::runtime_error("filter: an IN list of more than " + std::to_string(kMaxInConstants) + " constants over a column that is not a 4- or 8-byte integer / date / char(1)");
      std::vector<uint8_t> buf(w * (size_t) d.n_in);
      for (int32_t k = 0; k < d.n_in; k++) memcpy(buf.data() + w * (size_t) k, &d.in_values[2 * k], w); // (little-endian low word of the 128-bit constant)
      ct.nullable = 0;
      const char* colName = "in_value";
      const std::string tag = "$in_list:" + stem + ":" + std::to_string(env.size());
      ldb_table* t = nullptr;
      check(ldb_gpu_table_alloc(ctx, "in_list", 1, &ct, &colName, d.n_in, nullptr, 0, &t), "IN list (table)");
      putTable(tag, t);
      check(ldb_gpu_table_write_fixed(ctx, t, 0, buf.data(), (int64_t) buf.size()), "IN list (constants)");
      ldb_rel* br = nullptr;
      check(ldb_gpu_rel_from_table(ctx, t, &br), "IN list (relation)");
      putRel(tag + ":rel", br, {t});
      const ldb_colref key{0, 0};
      Value hv;
      hv.kind = Value::HT;
      hv.owned = true;
      hv.sides = {t};
      check(ldb_gpu_join_build(ctx, br, &key, 1, 0, &hv.ht), "IN list (hash table)");
      ldb_hashtable* ht = hv.ht;
      put(tag + ":ht", std::move(hv));
      ldb_rel* out = nullptr;
      check(ldb_gpu_join_probe(ctx, ht, cur, &d.col, 1, LDB_JOIN_SEMI, &out, nullptr), "IN list (semi join)");
      return out;
   }
   std::vector<ldb_filter_desc> preds(const std::vector<const ldb_table*>& sides, const J* list) {
      std::vector<ldb_filter_desc> out;
      if (!list) return out;
      for (auto& jp : list->arr) out.push_back(pred(sides, jp));
      return out;
   }

   // ------------------------------------------------------------ expressions
   static Scalar scalarOfCol(const ldb_coltype& t) {
      Scalar s;
      switch (t.type) {
         case LDB_T_DECIMAL128: s.kind = Scalar::DEC; s.p = t.precision; s.s = t.scale; break;
         case LDB_T_DATE32: s.kind = Scalar::DATE; break;
         case LDB_T_BOOL8: s.kind = Scalar::BOOL; break;
         case LDB_T_INT8:
         case LDB_T_INT16:
         case LDB_T_INT32:
         case LDB_T_INT64:
         case LDB_T_CHAR4: s.kind = Scalar::INT; break;
         default: throw std::runtime_error("plan: expression over a non-numeric column");
      }
      return s;
   }
   static DecimalType asDec(const Scalar& s) { return s.kind == Scalar::DEC ? DecimalType{s.p, s.s} : DecimalType{19, 0}; } // int → decimal(19,0), sql_analyzer.cpp:3125-3141
   static Scalar decScalar(DecimalType t) {
      Scalar s;
      s.kind = Scalar::DEC;
      s.p = t.p;
      s.s = t.s;
      return s;
   }
   // decimal literal "0.2" → decimal(2,1) value 2 (sql_analyzer.cpp:2103-2113: p = digits, s = digits after the point)
   static bool decimalLiteral(const std::string& txt, __int128* v, Scalar* t) {
      const auto dot = txt.find('.');
      if (dot == std::string::npos) return false;
      for (size_t i = 0; i < txt.size(); i++)
         if (!(isdigit((unsigned char) txt[i]) || txt[i] == '.' || (i == 0 && txt[i] == '-'))) return false;
      const int32_t s = (int32_t) (txt.size() - dot - 1);
      const bool neg = txt[0] == '-';
      const int32_t p = (int32_t) txt.size() - 1 - (neg ? 1 : 0);
      *v = parseDecimal(txt, s);
      *t = decScalar({p, s});
      return true;
   }

   struct XB { // postfix program under construction
      std::vector<ldb_xinstr> ins;
      void push(int32_t op, int32_t arg = 0, ldb_colref c = {0, 0}, __int128 k = 0) {
         ldb_xinstr x;
         memset(&x, 0, sizeof(x));
         x.op = op;
         x.arg = arg;
         x.col = c;
         x.lo = (int64_t) (uint64_t) k;
         x.hi = (int64_t) (k >> 64);
         ins.push_back(x);
      }
   };
   static void castTo(XB& b, const Scalar& from, const DecimalType& to) { // db.cast to a decimal of larger-or-equal scale
      const int32_t fs = from.kind == Scalar::DEC ? from.s : 0;
      if (to.s > fs) b.push(LDB_X_MUL_POW10, to.s - fs);
      else if (to.s < fs) b.push(LDB_X_SDIV_POW10, fs - to.s);
   }
   // compile an expression tree into postfix code; returns its type
   Scalar compileX(const std::vector<const ldb_table*>& sides, const J& e, XB& b) {
      if (e.kind == J::NUM) {
         if (!e.isInt) throw std::runtime_error("plan: write decimal literals as strings (\"0.2\")");
         b.push(LDB_X_CONST, 0, {0, 0}, (__int128) e.inum);
         return Scalar{};
      }
      if (e.kind == J::STR) {
         __int128 v;
         Scalar t;
         if (decimalLiteral(e.str, &v, &t)) {
            b.push(LDB_X_CONST, 0, {0, 0}, v);
            return t;
         }
         const ldb_colref c = resolve(sides, e.str, "expression");
         b.push(LDB_X_COL, 0, c);
         return scalarOfCol(typeOf(sides, c));
      }
      if (e.kind != J::OBJ || e.obj.size() != 1) throw std::runtime_error("plan: expression node must be {\"op\": [args]}");
      const std::string& op = e.obj[0].first;
      const J& args = e.obj[0].second;
      auto arity = [&](size_t n) {
         if (args.kind != J::ARR || args.arr.size() != n) throw std::runtime_error("plan: '" + op + "' takes " + std::to_string(n) + " arguments");
      };
      if (op == "add" || op == "sub") {
         arity(2);
         XB lb, rb;
         const Scalar l = compileX(sides, args.arr[0], lb), r = compileX(sides, args.arr[1], rb);
         if (l.kind == Scalar::INT && r.kind == Scalar::INT) {
            b.ins.insert(b.ins.end(), lb.ins.begin(), lb.ins.end());
            b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
            b.push(op == "add" ? LDB_X_ADD : LDB_X_SUB);
            return Scalar{};
         }
         const DecimalType t = higherDecimalType(asDec(l), asDec(r)); // DecimalBinOpLowering after the frontend's casts (LowerToStd.cpp:680-699)
         b.ins.insert(b.ins.end(), lb.ins.begin(), lb.ins.end());
         castTo(b, l, t);
         b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
         castTo(b, r, t);
         b.push(op == "add" ? LDB_X_ADD : LDB_X_SUB);
         return decScalar(t);
      }
      if (op == "mul") {
         arity(2);
         const Scalar l = compileX(sides, args.arr[0], b);
         const Scalar r = compileX(sides, args.arr[1], b);
         b.push(LDB_X_MUL);
         if (l.kind == Scalar::INT && r.kind == Scalar::INT) return Scalar{};
         const DecimalType dl = asDec(l), dr = asDec(r), t = typeAfterMul(dl, dr); // DecimalMulOpLowering (LowerToStd.cpp:653-677)
         if (dl.s + dr.s > t.s) b.push(LDB_X_SDIV_POW10, dl.s + dr.s - t.s);
         return decScalar(t);
      }
      if (op == "div") {
         arity(2);
         const Scalar l = compileX(sides, args.arr[0], b);
         XB rb;
         const Scalar r = compileX(sides, args.arr[1], rb);
         const DecimalType dl = asDec(l), dr = asDec(r), t = typeAfterDiv(dl, dr); // DecimalOpScaledLowering (LowerToStd.cpp:631-651)
         const int32_t k = t.s + dr.s - dl.s;
         if (k < 0) throw std::runtime_error("plan: decimal division with a negative scale adjustment");
         b.push(LDB_X_MUL_POW10, k);
         b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
         b.push(LDB_X_SDIV);
         return decScalar(t);
      }
      if (op == "idiv") { // integer division of two integers (arith.divsi: truncating) — the result stays an integer
         arity(2);
         const Scalar l = compileX(sides, args.arr[0], b);
         const Scalar r = compileX(sides, args.arr[1], b);
         if (l.kind != Scalar::INT || r.kind != Scalar::INT) throw std::runtime_error("plan: idiv takes two integers (decimals divide with 'div')");
         b.push(LDB_X_SDIV);
         return Scalar{};
      }
      if (op == "row_number") { // the logical row number of the input relation, 0 … n-1 (a tuple's identity inside nested_map)
         arity(0);
         b.push(LDB_X_ROW);
         return Scalar{};
      }
      if (op == "cmp") { // ["GT", a, b]
         arity(3);
         const int32_t cmp = opOf(args.arr[0].str);
         if (cmp > LDB_F_GTE) throw std::runtime_error("plan: cmp needs EQ/NEQ/LT/LTE/GT/GTE");
         XB lb, rb;
         const Scalar l = compileX(sides, args.arr[1], lb), r = compileX(sides, args.arr[2], rb);
         b.ins.insert(b.ins.end(), lb.ins.begin(), lb.ins.end());
         if (l.kind == Scalar::DEC || r.kind == Scalar::DEC) {
            const DecimalType t = higherDecimalType(asDec(l), asDec(r));
            castTo(b, l, t);
            b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
            castTo(b, r, t);
         } else {
            b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
         }
         b.push(LDB_X_CMP, cmp);
         Scalar s;
         s.kind = Scalar::BOOL;
         return s;
      }
      if (op == "and" || op == "or") {
         arity(2);
         compileX(sides, args.arr[0], b);
         compileX(sides, args.arr[1], b);
         b.push(op == "and" ? LDB_X_AND : LDB_X_OR);
         Scalar s;
         s.kind = Scalar::BOOL;
         return s;
      }
      if (op == "not" || op == "isnull") {
         arity(1);
         compileX(sides, args.arr[0], b);
         b.push(op == "not" ? LDB_X_NOT : LDB_X_ISNULL);
         Scalar s;
         s.kind = Scalar::BOOL;
         return s;
      }
      if (op == "coalesce" || op == "case") { // coalesce(a, b) / case when c then a else b
         const bool isCase = op == "case";
         arity(isCase ? 3 : 2);
         if (isCase) compileX(sides, args.arr[0], b);
         XB lb, rb;
         const Scalar l = compileX(sides, args.arr[isCase ? 1 : 0], lb), r = compileX(sides, args.arr[isCase ? 2 : 1], rb);
         Scalar out = l;
         b.ins.insert(b.ins.end(), lb.ins.begin(), lb.ins.end());
         if (l.kind == Scalar::DEC || r.kind == Scalar::DEC) {
            const DecimalType t = higherDecimalType(asDec(l), asDec(r));
            castTo(b, l, t);
            b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
            castTo(b, r, t);
            out = decScalar(t);
         } else {
            b.ins.insert(b.ins.end(), rb.ins.begin(), rb.ins.end());
         }
         b.push(isCase ? LDB_X_SELECT : LDB_X_COALESCE);
         return out;
      }
      if (op == "neg") {
         arity(1);
         const Scalar t = compileX(sides, args.arr[0], b);
         b.push(LDB_X_NEG);
         return t;
      }
      throw std::runtime_error("plan: unknown expression operator '" + op + "'");
   }

   // aggregate argument → the sum-of-products normal form of ldb_expr (what the decimal lowerings
   // reduce to once the scales are fixed); returns the result type
   struct Factor {
      bool isCol = false, isConst = false;
      ldb_colref col{0, 0};
      __int128 k = 0; // constant (unscaled)
      int sign = 1; // (k + sign * col)
      bool constPlus = false;
      Scalar type;
   };
   Factor factorOf(const std::vector<const ldb_table*>& sides, const J& e) {
      Factor f;
      if (e.kind == J::NUM) {
         f.isConst = true;
         f.k = e.inum;
         return f;
      }
      if (e.kind == J::STR) {
         __int128 v;
         Scalar t;
         if (decimalLiteral(e.str, &v, &t)) {
            f.isConst = true;
            f.k = v;
            f.type = t;
            return f;
         }
         f.isCol = true;
         f.col = resolve(sides, e.str, "aggregate");
         f.type = scalarOfCol(typeOf(sides, f.col));
         return f;
      }
      if (e.kind == J::OBJ && e.obj.size() == 1 && (e.obj[0].first == "add" || e.obj[0].first == "sub")) { // (k ± col)
         const J& a = e.obj[0].second;
         if (a.kind == J::ARR && a.arr.size() == 2 && a.arr[0].kind == J::NUM && a.arr[1].kind == J::STR) {
            f.constPlus = true;
            f.col = resolve(sides, a.arr[1].str, "aggregate");
            const Scalar ct = scalarOfCol(typeOf(sides, f.col));
            f.sign = e.obj[0].first == "add" ? 1 : -1;
            if (ct.kind == Scalar::DEC) { // int literal → decimal(19,0) → common scale of the column (sql_analyzer.cpp:3125-3141)
               f.type = decScalar(higherDecimalType({19, 0}, {ct.p, ct.s}));
               f.k = (__int128) a.arr[0].inum * pow10i(ct.s);
            } else {
               f.type = ct;
               f.k = a.arr[0].inum;
            }
            return f;
         }
      }
      throw std::runtime_error("plan: aggregate factors must be a column, a literal or (integer ± column)");
   }
   Scalar termOf(const std::vector<const ldb_table*>& sides, const J& e, ldb_term* t) {
      memset(t, 0, sizeof(*t));
      std::vector<const J*> fs;
      if (e.kind == J::OBJ && e.obj.size() == 1 && e.obj[0].first == "mul") {
         for (auto& a : e.obj[0].second.arr) fs.push_back(&a);
      } else {
         fs.push_back(&e);
      }
      if (fs.size() > LDB_MAX_FACTORS) throw std::runtime_error("plan: more than 3 factors in an aggregate product");
      Scalar type;
      bool first = true;
      for (auto* fe : fs) {
         const Factor f = factorOf(sides, *fe);
         ldb_factor& o = t->f[t->n_factors++];
         if (f.isConst) {
            o = {0, {0, 0}, (int64_t) f.k, 0};
         } else if (f.constPlus) {
            o = {1, f.col, (int64_t) f.k, f.sign};
         } else {
            o = {1, f.col, 0, 1};
         }
         if (first) {
            type = f.type;
            first = false;
         } else if (type.kind == Scalar::DEC || f.type.kind == Scalar::DEC) {
            const DecimalType a = asDec(type), b2 = asDec(f.type), r = typeAfterMul(a, b2);
            if (a.s + b2.s != r.s) t->div_pow10 += a.s + b2.s - r.s; // clamped scale: truncating divide (LowerToStd.cpp:653-677)
            type = decScalar(r);
         }
      }
      return type;
   }
   Scalar aggExpr(const std::vector<const ldb_table*>& sides, const J& e, ldb_expr* out) {
      memset(out, 0, sizeof(*out));
      if (e.kind == J::OBJ && e.obj.size() == 1 && (e.obj[0].first == "add" || e.obj[0].first == "sub")) {
         const J& a = e.obj[0].second;
         const bool constPlusCol = a.kind == J::ARR && a.arr.size() == 2 && a.arr[0].kind == J::NUM && a.arr[1].kind == J::STR;
         if (!constPlusCol) { // difference / sum of two products (Q9's amount)
            if (a.kind != J::ARR || a.arr.size() != 2) throw std::runtime_error("plan: add/sub take two arguments");
            out->n_terms = 2;
            const Scalar l = termOf(sides, a.arr[0], &out->t[0]), r = termOf(sides, a.arr[1], &out->t[1]);
            out->t[1].negate = e.obj[0].first == "sub";
            if (l.kind != Scalar::DEC && r.kind != Scalar::DEC) return Scalar{};
            const DecimalType dl = asDec(l), dr = asDec(r), t = higherDecimalType(dl, dr);
            // bring both terms to the common scale with one more constant factor
            ldb_term* ts[2] = {&out->t[0], &out->t[1]};
            const DecimalType ds[2] = {dl, dr};
            for (int i = 0; i < 2; i++) {
               if (ds[i].s < t.s) {
                  if (ts[i]->n_factors >= LDB_MAX_FACTORS) throw std::runtime_error("plan: no room for the scale factor in an aggregate term");
                  ts[i]->f[ts[i]->n_factors++] = {0, {0, 0}, pow10i(t.s - ds[i].s), 0};
               }
            }
            return decScalar(t);
         }
      }
      out->n_terms = 1;
      return termOf(sides, e, &out->t[0]);
   }

   // ------------------------------------------------------------ steps
   void run(const J& plan, const char* const* names, const ldb_table* const* tables, int32_t n) {
      for (int32_t i = 0; i < n; i++) {
         Value v;
         v.kind = Value::TABLE;
         v.table = tables[i];
         put(names[i], v);
      }
      const J& steps = plan.at("steps");
      result = plan.sOr("result", "result");
      static const bool stepTrace = getenv("LDB_PLAN_STEP_TRACE") != nullptr; // diagnostics: host time of every step on stderr
      for (auto& st : steps.arr) {
         try {
            const auto t0 = std::chrono::steady_clock::now();
            step(st);
            if (stepTrace)
               fprintf(stderr, "[ldb plan] %s: %s -> %s: %.3f ms host\n", plan.sOr("name", "plan").c_str(), st.sOr("op", "?").c_str(), st.sOr("out", "").c_str(),
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
         } catch (const std::exception& e) {
            throw std::runtime_error(plan.sOr("name", "plan") + ": step '" + st.sOr("op", "?") + "' → '" + st.sOr("out", "") + "': " + e.what());
         }
      }
   }

   std::vector<ldb_colref> cols(const std::vector<const ldb_table*>& sides, const J& list, const char* what) {
      std::vector<ldb_colref> out;
      for (auto& c : list.arr) out.push_back(resolve(sides, c.kind == J::STR ? c.str : c.s("col"), what));
      return out;
   }

   void step(const J& st) {
      const std::string& op = st.s("op");
      if (op == "scan") { // subop.scan_refs over a table (get_external)
         Value& t = val(st.s("table"));
         if (t.kind != Value::TABLE) throw std::runtime_error("scan: not a table");
         ldb_rel* r;
         check(ldb_gpu_rel_from_table(ctx, t.table, &r), "scan");
         putRel(st.s("out"), r, {t.table});
      } else if (op == "filter") {
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         auto ps = preds(*sides, &st.at("preds"));
         // Limits of ONE scan_filter call that the reference's Restrictions do not have (round 6): a conjunction of more than eight conjuncts is applied
         // eight at a time (a conjunction may be evaluated in any order); an integer IN list of more than eight constants — Restrictions.cpp:481-515 keeps
         // a hash set of any size — becomes what it is relationally: a semi join against the table of its constants
         std::vector<ldb_filter_desc> plain, longIn;
         for (auto& d : ps) (d.op == LDB_F_IN && d.n_in > kMaxInConstants && d.in_values ? longIn : plain).push_back(d);
         ldb_rel* cur = in;
         auto advance = [&](ldb_rel* next) {
            if (cur != in) check(ldb_gpu_rel_release(ctx, cur), "filter (intermediate)");
            cur = next;
         };
         for (size_t at = 0; at < plain.size() || (at == 0 && longIn.empty()); at += kMaxConjuncts) {
            const size_t n = std::min<size_t>(kMaxConjuncts, plain.size() - at);
            ldb_rel* r;
            check(ldb_gpu_scan_filter(ctx, cur, plain.data() + at, (int32_t) n, &r), "filter");
            advance(r);
         }
         for (auto& d : longIn) advance(semiJoinConstants(cur, d, *sides, st.s("out")));
         putRel(st.s("out"), cur, *sides);
      } else if (op == "filter_dnf") {
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         std::vector<ldb_filter_desc> all;
         std::vector<int32_t> sizes;
         for (auto& cl : st.at("clauses").arr) {
            auto ps = preds(*sides, &cl);
            sizes.push_back((int32_t) ps.size());
            all.insert(all.end(), ps.begin(), ps.end());
         }
         ldb_rel* r;
         check(ldb_gpu_scan_filter_dnf(ctx, in, all.data(), sizes.data(), (int32_t) sizes.size(), &r), "filter_dnf");
         putRel(st.s("out"), r, *sides);
      } else if (op == "join_build") {
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         auto keys = cols(*sides, st.at("keys"), "join_build");
         Value v;
         v.kind = Value::HT;
         v.owned = true;
         v.sides = *sides;
         Value& src = val(st.s("in"));
         if (src.kind == Value::TABLE && st.bOr("unique", false) && st.bOr("index", true)) {
            // a bare base table keyed by its primary key: the table's own (persistent) hash index — the reference's index
            // nested-loop join over LingoDBHashIndex (translateINLJ); built once, reused by every later plan run
            std::vector<int32_t> cs;
            for (auto& k : keys) cs.push_back(k.col);
            v.owned = false;
            check(ldb_gpu_table_index(ctx, const_cast<ldb_table*>(src.table), cs.data(), (int32_t) cs.size(), &v.ht), "join_build (table index)");
         } else {
            check(ldb_gpu_join_build(ctx, in, keys.data(), (int32_t) keys.size(), st.bOr("unique", false) ? 1 : 0, &v.ht), "join_build");
         }
         put(st.s("out"), std::move(v));
      } else if (op == "join_probe") {
         Value& ht = val(st.s("ht"));
         if (ht.kind != Value::HT) throw std::runtime_error("join_probe: 'ht' is not a hash table");
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         auto keys = cols(*sides, st.at("keys"), "join_probe");
         static const std::map<std::string, int32_t> kinds = {{"inner", LDB_JOIN_INNER}, {"semi", LDB_JOIN_SEMI}, {"anti", LDB_JOIN_ANTI}, {"left_outer", LDB_JOIN_LEFT_OUTER},
                                                             {"mark", LDB_JOIN_MARK},   {"single", LDB_JOIN_SINGLE}, {"semi_build", LDB_JOIN_SEMI_BUILD}, {"anti_build", LDB_JOIN_ANTI_BUILD},
                                                             {"right_outer", LDB_JOIN_RIGHT_OUTER}, {"full_outer", LDB_JOIN_FULL_OUTER}};
         if (st.sOr("kind", "inner") == "semi_anti_build") {
            // EXISTS (probe rows) AND NOT EXISTS (probe rows that also pass `anti_preds`) for every build row, one pass over the probe side
            // (Q21; the reference lowers the two subqueries to two marker joins, translateHJWithMarker)
            std::vector<ldb_join_residual> resid;
            if (const J* rs = st.get("residual"))
               for (auto& r : rs->arr) resid.push_back({resolve(*sides, r.s("probe"), "residual probe"), resolve(ht.sides, r.s("build"), "residual build"), opOf(r.s("op")), 0});
            auto ap = preds(*sides, &st.at("anti_preds"));
            ldb_rel* r;
            check(ldb_gpu_join_probe_semi_anti_build(ctx, ht.ht, in, keys.data(), (int32_t) keys.size(), resid.data(), (int32_t) resid.size(), ap.data(), (int32_t) ap.size(), &r), "join_probe (semi + anti, build side)");
            putRel(st.s("out"), r, ht.sides);
            return;
         }
         auto kit = kinds.find(st.sOr("kind", "inner"));
         if (kit == kinds.end()) throw std::runtime_error("join_probe: unknown kind");
         const int32_t kind = kit->second;
         std::vector<ldb_join_residual> resid;
         if (const J* rs = st.get("residual"))
            for (auto& r : rs->arr) resid.push_back({resolve(*sides, r.s("probe"), "residual probe"), resolve(ht.sides, r.s("build"), "residual build"), opOf(r.s("op")), 0});
         ldb_rel* r;
         ldb_table* mark = nullptr;
         // an INNER join takes any number of residual conjuncts (round 6): the probe kernel checks the first two on the candidate pair, the rest are
         // column-vs-column comparisons over the joined rows — for an inner join the same rows (the reference evaluates the whole non-equality part of
         // the predicate as the filter behind the lookup, SpecializeSubOpPass.cpp:152-205).  The other kinds decide per PROBE row and keep the limit
         std::vector<ldb_join_residual> later;
         if (kind == LDB_JOIN_INNER && resid.size() > (size_t) kMaxResidual) {
            later.assign(resid.begin() + kMaxResidual, resid.end());
            resid.resize((size_t) kMaxResidual);
         }
         check(ldb_gpu_join_probe_residual(ctx, ht.ht, in, keys.data(), (int32_t) keys.size(), kind, resid.data(), (int32_t) resid.size(), &r, &mark), "join_probe");
         if (!later.empty()) {
            std::vector<ldb_filter_desc> fs;
            for (auto& x : later) {
               ldb_filter_desc d;
               memset(&d, 0, sizeof(d));
               d.col = x.probe_col;
               d.op = x.op;
               d.rhs_kind = LDB_RHS_COLUMN;
               d.rhs_col = {(int32_t) (x.build_col.side + (int32_t) sides->size()), x.build_col.col}; // (the result's sides: probe sides, then build sides)
               fs.push_back(d);
            }
            for (size_t at = 0; at < fs.size(); at += kMaxConjuncts) {
               ldb_rel* f;
               check(ldb_gpu_scan_filter(ctx, r, fs.data() + at, (int32_t) std::min<size_t>(kMaxConjuncts, fs.size() - at), &f), "join_probe (further residual conjuncts)");
               check(ldb_gpu_rel_release(ctx, r), "join_probe (intermediate)");
               r = f;
            }
         }
         std::vector<const ldb_table*> outSides;
         if (kind == LDB_JOIN_SEMI_BUILD || kind == LDB_JOIN_ANTI_BUILD) {
            outSides = ht.sides;
         } else {
            outSides = *sides;
            if (kind == LDB_JOIN_INNER || kind == LDB_JOIN_LEFT_OUTER || kind == LDB_JOIN_SINGLE || kind == LDB_JOIN_RIGHT_OUTER || kind == LDB_JOIN_FULL_OUTER)
               outSides.insert(outSides.end(), ht.sides.begin(), ht.sides.end());
         }
         if (mark && st.get("mark_as")) { // the mark column as a column of the result relation (MarkJoinLowering: the stream carries the mark attribute)
            ldb_gpu_table_rename_col(mark, 0, st.s("mark_as").c_str());
            ldb_rel* z;
            check(ldb_gpu_rel_zip(ctx, r, mark, &z), "join_probe (mark column)");
            outSides.push_back(mark);
            ldb_gpu_rel_release(ctx, r);
            r = z;
         }
         putRel(st.s("out"), r, std::move(outSides));
         if (mark) {
            if (const J* mo = st.get("mark_out")) putTable(mo->str, mark);
            else hidden.push_back(mark);
         }
      } else if (op == "join_nl") { // translateNLJ: no key equality, the predicate is the residual conjuncts (small build sides)
         std::vector<const ldb_table*>*sides, *bsides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         ldb_rel* build = relOf(st.s("build"), &bsides);
         static const std::map<std::string, int32_t> kinds = {{"inner", LDB_JOIN_INNER}, {"semi", LDB_JOIN_SEMI}, {"anti", LDB_JOIN_ANTI}, {"left_outer", LDB_JOIN_LEFT_OUTER},
                                                             {"semi_build", LDB_JOIN_SEMI_BUILD}, {"anti_build", LDB_JOIN_ANTI_BUILD}};
         auto kit = kinds.find(st.sOr("kind", "inner"));
         if (kit == kinds.end()) throw std::runtime_error("join_nl: unknown kind");
         const int32_t kind = kit->second;
         std::vector<ldb_join_residual> resid;
         if (const J* rs = st.get("residual"))
            for (auto& r : rs->arr) resid.push_back({resolve(*sides, r.s("probe"), "join_nl probe column"), resolve(*bsides, r.s("build"), "join_nl build column"), opOf(r.s("op")), 0});
         ldb_rel* r;
         check(ldb_gpu_join_nl(ctx, in, build, kind, resid.data(), (int32_t) resid.size(), &r, nullptr), "join_nl");
         std::vector<const ldb_table*> outSides;
         if (kind == LDB_JOIN_SEMI_BUILD || kind == LDB_JOIN_ANTI_BUILD) {
            outSides = *bsides;
         } else {
            outSides = *sides;
            if (kind == LDB_JOIN_INNER || kind == LDB_JOIN_LEFT_OUTER) outSides.insert(outSides.end(), bsides->begin(), bsides->end());
         }
         putRel(st.s("out"), r, std::move(outSides));
      } else if (op == "groupby") {
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         auto ps = preds(*sides, st.get("preds"));
         std::vector<ldb_colref> keys;
         if (const J* k = st.get("keys")) keys = cols(*sides, *k, "groupby key");
         std::vector<ldb_agg_spec> aggs;
         std::vector<std::string> names;
         for (auto& ja : st.at("aggs").arr) {
            ldb_agg_spec a;
            memset(&a, 0, sizeof(a));
            const std::string fn = ja.s("fn");
            names.push_back(ja.sOr("as", ""));
            if (const J* when = ja.get("when")) { // sum(case when <conjunction> then x else 0 end)
               auto wp = preds(*sides, when);
               if (wp.size() > LDB_MAX_AGG_PREDS) throw std::runtime_error("groupby: more than 3 conjuncts in a conditional aggregate");
               a.n_preds = (int32_t) wp.size();
               for (size_t p = 0; p < wp.size(); p++) a.preds[p] = wp[p];
            }
            if (fn == "count_star") {
               a.fn = LDB_AGG_COUNT_STAR;
               a.out_type = LDB_T_INT64;
            } else {
               const Scalar t = aggExpr(*sides, ja.at("expr"), &a.arg);
               if (const J* cnt = ja.get("count")) { // AVG over exchanged partials: SUM(expr) / SUM(count) — the merge of (sum, count) pairs
                  if (fn != "avg") throw std::runtime_error("groupby: 'count' belongs to fn avg");
                  a.has_count_expr = 1;
                  aggExpr(*sides, *cnt, &a.count_expr);
               }
               if (fn == "count") {
                  a.fn = LDB_AGG_COUNT;
                  a.out_type = LDB_T_INT64;
               } else {
                  static const std::map<std::string, int32_t> fns = {{"sum", LDB_AGG_SUM}, {"min", LDB_AGG_MIN}, {"max", LDB_AGG_MAX}, {"any", LDB_AGG_ANY}, {"avg", LDB_AGG_AVG}};
                  auto fit = fns.find(fn);
                  if (fit == fns.end()) throw std::runtime_error("groupby: unknown aggregate '" + fn + "'");
                  a.fn = fit->second;
                  if (t.kind == Scalar::DEC) { // SUM / MIN / MAX keep the argument type (sql_analyzer.cpp:2631-2632)
                     DecimalType dt{t.p, t.s};
                     a.wide = dt.wide();
                     a.out_type = LDB_T_DECIMAL128;
                     a.out_precision = dt.p;
                     a.out_scale = dt.s;
                     if (a.fn == LDB_AGG_AVG) { // SUM / COUNT with the divisor typed decimal(19,0) (:2636-2642)
                        const DecimalType r = avgType(dt);
                        a.avg_pow10 = r.s - dt.s;
                        a.out_precision = r.p;
                        a.out_scale = r.s;
                     }
                  } else if (t.kind == Scalar::DATE) {
                     a.out_type = LDB_T_DATE32;
                  } else {
                     a.out_type = ja.sOr("type", "int64") == "int32" ? LDB_T_INT32 : LDB_T_INT64;
                     if (a.fn == LDB_AGG_AVG) throw std::runtime_error("groupby: avg over integers is not supported (cast to decimal)");
                  }
               }
            }
            aggs.push_back(a);
         }
         // expected number of groups (sizes the hash table; a low estimate costs a retry, never a wrong result):
         // a number, "est_groups_from": "rows" (the input's row count), or {"rows_of": value, "div": d, "min": m}
         int64_t est = 0;
         if (const J* eg = st.get("est_groups")) {
            if (eg->kind == J::OBJ) {
               est = rowsOf(eg->s("rows_of")) / std::max<int64_t>(1, eg->iOr("div", 1));
               est = std::max<int64_t>(est, eg->iOr("min", 1));
            } else {
               est = eg->inum;
            }
         }
         if (st.sOr("est_groups_from", "") == "rows") est = std::max<int64_t>(1, ldb_gpu_rel_rows(ctx, in));
         const bool learn = est == 0 && groupsSeen && !keys.empty();
         if (learn) {
            auto seen = groupsSeen->find(st.s("out"));
            if (seen != groupsSeen->end()) est = std::max<int64_t>(1, seen->second);
         }
         ldb_table* out;
         check(ldb_gpu_groupby(ctx, in, ps.data(), (int32_t) ps.size(), keys.data(), (int32_t) keys.size(), aggs.data(), (int32_t) aggs.size(), est, &out), "groupby");
         if (learn) (*groupsSeen)[st.s("out")] = ldb_gpu_table_rows(out);
         for (size_t a = 0; a < names.size(); a++)
            if (!names[a].empty()) ldb_gpu_table_rename_col(out, (int32_t) (keys.size() + a), names[a].c_str());
         if (const J* kn = st.get("key_names"))
            for (size_t k = 0; k < kn->arr.size() && k < keys.size(); k++) ldb_gpu_table_rename_col(out, (int32_t) k, kn->arr[k].str.c_str());
         putTable(st.s("out"), out);
      } else if (op == "map") { // subop.map: one computed column, attached as a new last side
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         const std::string as = st.s("as");
         ldb_table* t = nullptr;
         const std::string fn = st.sOr("fn", "");
         if (fn == "extract_year") {
            check(ldb_gpu_map_column(ctx, in, resolve(*sides, st.s("col"), "map"), LDB_FN_EXTRACT_YEAR, as.c_str(), &t), "map extract_year");
         } else if (fn == "substr") {
            check(ldb_gpu_map_substr(ctx, in, resolve(*sides, st.s("col"), "map"), st.iOr("from", 1), st.iOr("for", 1 << 30), as.c_str(), &t), "map substr");
         } else {
            XB b;
            const Scalar ty = compileX(*sides, st.at("expr"), b);
            ldb_coltype ct = {LDB_T_INT64, 0, 0, 1};
            if (ty.kind == Scalar::DEC) ct = {LDB_T_DECIMAL128, ty.p, ty.s, 1};
            else if (ty.kind == Scalar::BOOL) ct = {LDB_T_BOOL8, 0, 0, 1};
            else if (ty.kind == Scalar::DATE) ct = {LDB_T_DATE32, 0, 0, 1};
            check(ldb_gpu_map_expr(ctx, in, b.ins.data(), (int32_t) b.ins.size(), ct, as.c_str(), &t), "map expr");
         }
         hidden.push_back(t);
         ldb_rel* r;
         check(ldb_gpu_rel_zip(ctx, in, t, &r), "map zip");
         auto outSides = *sides;
         outSides.push_back(t);
         putRel(st.s("out"), r, std::move(outSides));
      } else if (op == "sort" || op == "topk") {
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         std::vector<ldb_sort_spec> specs;
         for (auto& b : st.at("by").arr) specs.push_back({resolve(*sides, b.kind == J::STR ? b.str : b.s("col"), "sort"), b.kind == J::OBJ && b.bOr("desc", false) ? 1 : 0, 0});
         ldb_rel* r;
         if (op == "sort") check(ldb_gpu_sort(ctx, in, specs.data(), (int32_t) specs.size(), &r), "sort");
         else check(ldb_gpu_topk(ctx, in, specs.data(), (int32_t) specs.size(), st.at("k").inum, &r), "topk");
         putRel(st.s("out"), r, *sides);
      } else if (op == "materialize") { // MaterializeTableLowering: gather the listed columns
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         auto cs = cols(*sides, st.at("cols"), "materialize");
         ldb_table* t;
         check(ldb_gpu_materialize(ctx, in, cs.data(), (int32_t) cs.size(), &t), "materialize");
         size_t i = 0;
         for (auto& c : st.at("cols").arr) {
            if (c.kind == J::OBJ)
               if (const J* as = c.get("as")) ldb_gpu_table_rename_col(t, (int32_t) i, as->str.c_str());
            i++;
         }
         putTable(st.s("out"), t);
      } else if (op == "set_op") { // UnionAll / UnionDistinct / CountingSetOperation lowerings: two relations, column lists of pairwise equal types
         std::vector<const ldb_table*>*ls, *rs;
         ldb_rel* left = relOf(st.s("left"), &ls);
         ldb_rel* right = relOf(st.s("right"), &rs);
         auto lc = cols(*ls, st.at("left_cols"), "set_op (left)");
         auto rc = cols(*rs, st.at("right_cols"), "set_op (right)");
         if (lc.size() != rc.size() || lc.empty()) throw std::runtime_error("set_op: the two column lists must have the same, non-zero length");
         static const std::map<std::string, int32_t> kinds = {{"union_all", LDB_SET_UNION_ALL}, {"union", LDB_SET_UNION},   {"intersect", LDB_SET_INTERSECT},
                                                             {"intersect_all", LDB_SET_INTERSECT_ALL}, {"except", LDB_SET_EXCEPT}, {"except_all", LDB_SET_EXCEPT_ALL}};
         auto kit = kinds.find(st.s("kind"));
         if (kit == kinds.end()) throw std::runtime_error("set_op: unknown kind '" + st.s("kind") + "'");
         ldb_table* t;
         check(ldb_gpu_set_op(ctx, left, lc.data(), right, rc.data(), (int32_t) lc.size(), kit->second, &t), "set_op");
         if (const J* as = st.get("as"))
            for (size_t k = 0; k < as->arr.size() && k < lc.size(); k++) ldb_gpu_table_rename_col(t, (int32_t) k, as->arr[k].str.c_str());
         putTable(st.s("out"), t);
      } else if (op == "window") { // WindowLowering: partition + order, one frame for all functions; the function results are attached as new columns
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         std::vector<ldb_colref> part;
         if (const J* pk = st.get("partition_by")) part = cols(*sides, *pk, "window partition key");
         std::vector<ldb_sort_spec> order;
         if (const J* ob = st.get("order_by"))
            for (auto& b : ob->arr) order.push_back({resolve(*sides, b.kind == J::STR ? b.str : b.s("col"), "window order"), b.kind == J::OBJ && b.bOr("desc", false) ? 1 : 0, 0});
         auto frameEnd = [&](const char* key, int64_t dflt) -> int64_t {
            const J* f = st.get(key);
            if (!f) return dflt;
            if (f->kind == J::STR) {
               if (f->str == "unbounded_preceding") return LDB_FRAME_UNBOUNDED_PRECEDING;
               if (f->str == "unbounded_following") return LDB_FRAME_UNBOUNDED_FOLLOWING;
               if (f->str == "current_row") return 0;
               throw std::runtime_error("window: frame end '" + f->str + "'");
            }
            return f->inum;
         };
         const int64_t from = frameEnd("frame_from", LDB_FRAME_UNBOUNDED_PRECEDING), to = frameEnd("frame_to", 0);
         static const std::map<std::string, int32_t> fns = {{"rank", LDB_WIN_RANK}, {"sum", LDB_WIN_SUM}, {"min", LDB_WIN_MIN}, {"max", LDB_WIN_MAX}, {"count", LDB_WIN_COUNT}, {"count_star", LDB_WIN_COUNT_STAR}};
         std::vector<ldb_window_fn> wf;
         std::vector<std::string> names;
         for (auto& f : st.at("fns").arr) {
            auto fit = fns.find(f.s("fn"));
            if (fit == fns.end()) throw std::runtime_error("window: unknown function '" + f.s("fn") + "'");
            ldb_window_fn w;
            memset(&w, 0, sizeof(w));
            w.fn = fit->second;
            if (const J* c = f.get("col")) w.col = resolve(*sides, c->str, "window argument");
            wf.push_back(w);
            names.push_back(f.s("as"));
         }
         if (wf.empty()) throw std::runtime_error("window: no functions");
         ldb_rel* r;
         ldb_table* t;
         check(ldb_gpu_window(ctx, in, part.data(), (int32_t) part.size(), order.data(), (int32_t) order.size(), from, to, wf.data(), (int32_t) wf.size(), &r, &t), "window");
         for (size_t k = 0; k < names.size(); k++) ldb_gpu_table_rename_col(t, (int32_t) k, names[k].c_str());
         hidden.push_back(t);
         ldb_rel* z;
         check(ldb_gpu_rel_zip(ctx, r, t, &z), "window zip");
         ldb_gpu_rel_release(ctx, r);
         auto outSides = *sides;
         outSides.push_back(t);
         putRel(st.s("out"), z, std::move(outSides));
      } else if (op == "allgather") { // every rank's rows of a (small) table, concatenated in rank order on every rank
         Value& t = val(st.s("in"));
         if (t.kind != Value::TABLE) throw std::runtime_error("allgather: 'in' must be a table (materialize first)");
         ldb_table* out = nullptr;
         if (comm) {
            check(ldb_gpu_allgather(ctx, comm, t.table, st.s("out").c_str(), &out), "allgather");
         } else { // one rank: a copy
            std::vector<const ldb_table*>* sides;
            ldb_rel* in = relOf(st.s("in"), &sides);
            std::vector<ldb_colref> all;
            for (int32_t c = 0; c < ldb_gpu_table_cols(t.table); c++) all.push_back({0, c});
            check(ldb_gpu_materialize(ctx, in, all.data(), (int32_t) all.size(), &out), "allgather (one rank)");
         }
         putTable(st.s("out"), out);
      } else if (op == "shuffle") { // hash-radix re-partition on db.hash(keys): afterwards equal keys are on one rank
         std::vector<const ldb_table*>* sides;
         ldb_rel* in = relOf(st.s("in"), &sides);
         auto keys = cols(*sides, st.at("keys"), "shuffle key");
         auto cs = cols(*sides, st.at("cols"), "shuffle");
         ldb_table* out = nullptr;
         if (comm) check(ldb_gpu_shuffle(ctx, comm, in, keys.data(), (int32_t) keys.size(), cs.data(), (int32_t) cs.size(), st.s("out").c_str(), &out), "shuffle");
         else check(ldb_gpu_materialize(ctx, in, cs.data(), (int32_t) cs.size(), &out), "shuffle (one rank)");
         size_t i = 0;
         for (auto& c : st.at("cols").arr) {
            if (c.kind == J::OBJ)
               if (const J* as = c.get("as")) ldb_gpu_table_rename_col(out, (int32_t) i, as->str.c_str());
            i++;
         }
         putTable(st.s("out"), out);
      } else if (op == "nested_map") {
         nestedMap(st);
      } else if (op == "loop") {
         loop(st);
      } else {
         throw std::runtime_error("unknown step");
      }
   }

   // ------------------------------------------------------------ nested_map / loop
   static J jStr(const std::string& s) {
      J j;
      j.kind = J::STR;
      j.str = s;
      return j;
   }
   static J jObj(std::vector<std::pair<std::string, J>> fields) {
      J j;
      j.kind = J::OBJ;
      j.obj = std::move(fields);
      return j;
   }
   static J jArr(std::vector<J> items) {
      J j;
      j.kind = J::ARR;
      j.arr = std::move(items);
      return j;
   }
   int tmpCounter = 0;
   // subop.nested_map (NestedMapLowering, SubOpToControlFlow.cpp:3996-4048): for every tuple of the outer stream the nested steps
   // run with the tuple's columns as parameters — in kmeans.mlir / pagerank.mlir: scan a state, map over (outer tuple, scanned
   // tuple), reduce into a per-tuple simple_state, scan that state.  On the device the correlation is removed: the outer tuples get
   // their row number as identity, the nested scan becomes a nested-loop join (outer x scanned state, optional residual conjuncts),
   // the nested maps run over the pairs, and the per-tuple state is a group-by on the identity.  Without "reduce" the pairs
   // themselves are the result (the nested stream returned as it is).
   //   {"op": "nested_map", "in": outer, "scan": inner, "residual": [...], "map": [{"as", "expr"} …],
   //    "reduce": {"aggs": [...], "keep": [outer columns]}, "out": name}
   void nestedMap(const J& st) {
      const std::string base = "__nm" + std::to_string(tmpCounter++) + "_";
      const std::string& out = st.s("out");
      const J* red = st.get("reduce");
      std::string cur = st.s("in");
      if (red) {
         step(jObj({{"op", jStr("map")}, {"in", jStr(cur)}, {"as", jStr("__tuple")}, {"expr", jObj({{"row_number", jArr({})}})}, {"out", jStr(base + "id")}}));
         cur = base + "id";
      }
      {
         std::vector<std::pair<std::string, J>> f = {{"op", jStr("join_nl")}, {"in", jStr(cur)}, {"build", jStr(st.s("scan"))}, {"kind", jStr("inner")}, {"out", jStr(base + "pairs")}};
         if (const J* rs = st.get("residual")) f.push_back({"residual", *rs});
         step(jObj(std::move(f)));
         cur = base + "pairs";
      }
      int k = 0;
      if (const J* maps = st.get("map"))
         for (auto& m : maps->arr) {
            const std::string nm = base + "m" + std::to_string(k++);
            step(jObj({{"op", jStr("map")}, {"in", jStr(cur)}, {"as", jStr(m.s("as"))}, {"expr", m.at("expr")}, {"out", jStr(nm)}}));
            cur = nm;
         }
      if (!red) { // the nested stream itself: the last value under the step's name
         auto it = env.find(cur);
         Value v = std::move(it->second);
         env.erase(it);
         put(out, std::move(v));
         return;
      }
      std::vector<J> aggs = red->at("aggs").arr;
      if (const J* keep = red->get("keep"))
         for (auto& c : keep->arr) aggs.push_back(jObj({{"fn", jStr("any")}, {"expr", jStr(c.str)}, {"as", jStr(c.str)}}));
      step(jObj({{"op", jStr("groupby")}, {"in", jStr(cur)}, {"keys", jArr({jStr("__tuple")})}, {"aggs", jArr(std::move(aggs))}, {"est_groups", jObj({{"rows_of", jStr(st.s("in"))}})}, {"out", jStr(out)}}));
   }

   // subop.loop (LoopLowering, SubOpToControlFlow.cpp:4058-4120: an scf.while over (condition, loop-carried states)): the body's
   // steps run once per iteration with the loop variables bound to the current states; loop_continue names the condition (a member
   // of a simple_state, here: column `col` of the one-row table `from`) and the states of the next iteration.  As in the
   // reference the loop's results are the states handed to the LAST loop_continue (the one whose condition was false).
   //   {"op": "loop", "vars": [{"name", "init"}], "body": [steps], "continue": {"from", "col"}, "next": [{"var", "from"}],
   //    "results": [{"name", "var"}], "max_iterations": 1000}
   // Loop-carried values are tables (materialised states: buffers, simple states, hash-map contents).
   void loop(const J& st) {
      struct Var {
         std::string name, next, result;
         const ldb_table* cur = nullptr;
         bool owned = false;
      };
      std::vector<Var> vars;
      for (auto& jv : st.at("vars").arr) {
         Var v;
         v.name = jv.s("name");
         Value& init = val(jv.s("init"));
         if (init.kind != Value::TABLE) throw std::runtime_error("loop: the initial value of '" + v.name + "' must be a table (materialize first)");
         v.cur = init.table;
         vars.push_back(v);
      }
      for (auto& jn : st.at("next").arr) {
         bool found = false;
         for (auto& v : vars)
            if (v.name == jn.s("var")) v.next = jn.s("from"), found = true;
         if (!found) throw std::runtime_error("loop: next names the unknown variable '" + jn.s("var") + "'");
      }
      for (auto& v : vars)
         if (v.next.empty()) throw std::runtime_error("loop: no next value for '" + v.name + "'");
      for (auto& jr : st.at("results").arr)
         for (auto& v : vars)
            if (v.name == jr.s("var")) v.result = jr.s("name");
      const J& body = st.at("body");
      const J& cont = st.at("continue");
      const int64_t maxIter = st.iOr("max_iterations", 1000);
      auto releaseVar = [&](Var& v) {
         if (v.owned && v.cur) ldb_gpu_table_release(ctx, const_cast<ldb_table*>(v.cur));
         v.cur = nullptr;
         v.owned = false;
      };
      try {
         for (int64_t iter = 0;; iter++) {
            if (iter >= maxIter) throw std::runtime_error("loop: no fixpoint after " + std::to_string(maxIter) + " iterations");
            std::vector<std::string> before;
            for (auto& kv : env) before.push_back(kv.first);
            const size_t hiddenBefore = hidden.size();
            for (auto& v : vars) {
               Value b;
               b.kind = Value::TABLE;
               b.table = v.cur;
               b.owned = false; // the loop owns it
               put(v.name, std::move(b));
            }
            for (auto& s : body.arr) step(s);
            // the condition and the next states, read before the iteration's values go away
            bool again = false;
            if (rowsOf(cont.s("from")) >= 1 && !scalarIsNull(cont.s("from"), cont.s("col"))) again = readScalar(cont.s("from"), cont.s("col")) != 0;
            std::vector<const ldb_table*> nexts;
            for (auto& v : vars) {
               Value& nv = val(v.next);
               if (nv.kind != Value::TABLE || !nv.owned) throw std::runtime_error("loop: the next value '" + v.next + "' must be a table produced inside the body");
               nexts.push_back(nv.table);
               nv.owned = false; // taken over by the loop
            }
            // everything else the body defined dies with the iteration: hash tables, then relations, then tables
            std::vector<std::string> mine;
            for (auto& kv : env)
               if (!std::binary_search(before.begin(), before.end(), kv.first)) mine.push_back(kv.first);
            for (auto& n : mine) {
               Value& v = env[n];
               if (v.kind == Value::HT && v.owned && v.ht) ldb_gpu_hashtable_release(ctx, v.ht);
            }
            for (auto& n : mine) {
               Value& v = env[n];
               if (v.lazyRel) ldb_gpu_rel_release(ctx, v.lazyRel);
               if (v.kind == Value::REL && v.owned && v.rel) ldb_gpu_rel_release(ctx, v.rel);
            }
            for (auto& n : mine) {
               Value& v = env[n];
               if (v.kind == Value::TABLE && v.owned && v.table) ldb_gpu_table_release(ctx, const_cast<ldb_table*>(v.table));
               env.erase(n);
            }
            for (size_t h = hiddenBefore; h < hidden.size(); h++) ldb_gpu_table_release(ctx, hidden[h]);
            hidden.resize(hiddenBefore);
            for (size_t i = 0; i < vars.size(); i++) {
               releaseVar(vars[i]);
               vars[i].cur = nexts[i];
               vars[i].owned = true;
            }
            if (!again) break;
         }
      } catch (...) {
         for (auto& v : vars) releaseVar(v);
         throw;
      }
      for (auto& v : vars) {
         if (v.result.empty()) {
            releaseVar(v);
         } else {
            putTable(v.result, const_cast<ldb_table*>(v.cur));
            v.cur = nullptr;
         }
      }
   }
};

thread_local std::string g_plan_json_err;

} // namespace

// ================================================================== prepared plans
// A plan is static data, yet ldb_plan_run_json parses it, resolves every name and waits for the device after every
// operator whose output size decides the next allocation.  ldb_plan_prepare parses once; ldb_plan_execute brackets the
// interpretation with a read-back trace (lingodb_gpu.h, ldb_gpu_trace_*): the first execution over a set of input tables
// records every count the operators read back, the following ones — same tables (ldb_gpu_table_stamp), same options
// (ldb_gpu_option_epoch) — replay them, so that the whole plan is issued without one wait and checked once at the end.  A
// check that fails (LDB_TRACE_MISSED / LDB_ERR_RETRY: something the key does not cover changed) discards the execution and
// repeats it recording.  The descriptors of a repeated execution are byte-identical to the previous one's and are served
// from the context's descriptor cache (no upload).  Reference shape: the query is compiled once and run as one main()
// (src/execution/LLVMBackends.cpp:856-865), a pipeline never returns to the driver between its operators
// (SubOpToControlFlow.cpp:1123-1202).
struct ldb_plan {
   ldb_ctx* ctx = nullptr;
   J doc;
   ldb_trace* trace = nullptr;
   std::vector<uint64_t> key; // what the trace was recorded under: per input (stamp, rows), option epoch, exchange or not
   std::map<std::string, int64_t> groupsSeen; // Interp::groupsSeen
   int64_t executions = 0, replays = 0, misses = 0, peerMisses = 0; // peerMisses: executions repeated because ANOTHER rank's replay failed
   double issue_ms = 0, wait_ms = 0; // over the replayed executions: host time to issue the whole plan / time spent in the one wait at its end
};

namespace {
int32_t runParsed(ldb_ctx* ctx, ldb_comm* comm, const J& plan, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables, ldb_table** result,
                  std::map<std::string, int64_t>* groupsSeen = nullptr) {
   try {
      Interp in(ctx);
      in.comm = comm;
      in.groupsSeen = groupsSeen;
      in.run(plan, table_names, tables, n_tables);
      Value& r = in.val(in.result);
      if (r.kind != Value::TABLE || !r.owned) throw std::runtime_error("plan: result '" + in.result + "' must be a table produced by the plan");
      *result = const_cast<ldb_table*>(r.table);
      return LDB_OK;
   } catch (const std::exception& e) { // (an operator that met a wrong replayed count fails with LDB_ERR_RETRY: the trace's end reports it)
      g_plan_json_err = e.what();
      return LDB_ERR_INVALID;
   }
}
} // namespace

// Run a JSON plan over the named input tables; *result = the table named by the plan's "result".
extern "C" int32_t ldb_plan_run_json(ldb_ctx* ctx, const char* plan_json, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables, ldb_table** result) {
   return ldb_plan_run_json_comm(ctx, nullptr, plan_json, table_names, tables, n_tables, result);
}
// the same with a communicator: the plan's `allgather` / `shuffle` steps exchange rows with the other ranks
// (every rank runs the same plan text over its shard)
extern "C" int32_t ldb_plan_run_json_comm(ldb_ctx* ctx, ldb_comm* comm, const char* plan_json, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables,
                                          ldb_table** result) {
   if (!ctx || !plan_json || !result || n_tables < 0) {
      g_plan_json_err = "plan_run_json: bad argument";
      return LDB_ERR_INVALID;
   }
   try {
      JParser parser(plan_json);
      const J plan = parser.value();
      // a one-off execution is bracketed like a prepared one (recording only): the context's counter arena restarts at the plan's
      // first operator, so that a text that is run again builds the descriptors the descriptor cache already holds
      ldb_trace* tr = nullptr;
      const bool bracket = ldb_gpu_trace_create(ctx, &tr) == LDB_OK && ldb_gpu_trace_begin(ctx, tr, 0) == LDB_OK;
      const int32_t st = runParsed(ctx, comm, plan, table_names, tables, n_tables, result);
      if (bracket) {
         int32_t status = 0;
         (void) ldb_gpu_trace_end(ctx, &status);
      }
      if (tr) ldb_gpu_trace_destroy(ctx, tr);
      return st;
   } catch (const std::exception& e) {
      g_plan_json_err = e.what();
      return LDB_ERR_INVALID;
   }
}
extern "C" int32_t ldb_plan_prepare(ldb_ctx* ctx, const char* plan_json, ldb_plan** out) {
   if (!ctx || !plan_json || !out) {
      g_plan_json_err = "plan_prepare: bad argument";
      return LDB_ERR_INVALID;
   }
   try {
      auto p = std::make_unique<ldb_plan>();
      p->ctx = ctx;
      JParser parser(plan_json);
      p->doc = parser.value();
      if (p->doc.kind != J::OBJ) throw std::runtime_error("plan: the top level must be an object");
      (void) p->doc.at("steps");
      if (ldb_gpu_trace_create(ctx, &p->trace) != LDB_OK) throw std::runtime_error(ldb_gpu_last_error());
      *out = p.release();
      return LDB_OK;
   } catch (const std::exception& e) {
      g_plan_json_err = e.what();
      return LDB_ERR_INVALID;
   }
}
extern "C" int32_t ldb_plan_release(ldb_plan* p) {
   if (!p) return LDB_OK;
   if (p->trace) ldb_gpu_trace_destroy(p->ctx, p->trace);
   delete p;
   return LDB_OK;
}
extern "C" int32_t ldb_plan_execute(ldb_plan* p, ldb_comm* comm, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables, ldb_table** result) {
   if (!p || !result || n_tables < 0 || (n_tables && (!table_names || !tables))) {
      g_plan_json_err = "plan_execute: bad argument";
      return LDB_ERR_INVALID;
   }
   std::vector<uint64_t> key;
   for (int32_t i = 0; i < n_tables; i++) {
      key.push_back(ldb_gpu_table_stamp(tables[i]));
      key.push_back((uint64_t) ldb_gpu_table_rows(tables[i]));
   }
   key.push_back((uint64_t) ldb_gpu_option_epoch());
   key.push_back(comm ? 1 : 0);
   // replay only over the very inputs the trace was recorded on.  With a communicator the ranks replay TOGETHER or not at all (a replaying rank
   // queues its transfers with the recorded sizes, so its peers must do the same): they agree on the minimum of "I could replay" before the
   // execution and on the minimum of their verdicts after it; a miss on any rank makes every rank repeat the execution recording.  Two small
   // collectives per execution instead of a wait at every operator (ldb_gpu_comm_agree); the plan itself is issued without a host wait.
   bool replay = key == p->key;
   if (comm) {
      int32_t all = 0;
      if (ldb_gpu_comm_agree(p->ctx, comm, replay && ldb_gpu_trace_replayable(p->trace) ? 1 : 0, &all) != LDB_OK) {
         g_plan_json_err = ldb_gpu_last_error();
         return LDB_ERR_HIP;
      }
      replay = all == 1;
   }
   const int32_t collective = comm ? LDB_TRACE_COLLECTIVE : 0;
   for (int attempt = 0; attempt < 2; attempt++) {
      if (ldb_gpu_trace_begin(p->ctx, p->trace, (replay && attempt == 0 ? 1 : 0) | collective) != LDB_OK) {
         g_plan_json_err = ldb_gpu_last_error();
         return LDB_ERR_INVALID;
      }
      ldb_table* out = nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      int32_t st = runParsed(p->ctx, comm, p->doc, table_names, tables, n_tables, &out, &p->groupsSeen);
      const auto t1 = std::chrono::steady_clock::now();
      int32_t status = LDB_TRACE_OFF;
      if (ldb_gpu_trace_end(p->ctx, &status) != LDB_OK) {
         if (out) ldb_gpu_table_release(p->ctx, out);
         g_plan_json_err = ldb_gpu_last_error();
         return LDB_ERR_HIP;
      }
      p->executions++;
      if (comm && replay && attempt == 0) { // did every rank's replay hold?  (only asked where a replay was attempted: all ranks take this branch together)
         int32_t all_ok = 0;
         if (ldb_gpu_comm_agree(p->ctx, comm, status == LDB_TRACE_MISSED ? 0 : 1, &all_ok) != LDB_OK) {
            if (out) ldb_gpu_table_release(p->ctx, out);
            g_plan_json_err = ldb_gpu_last_error();
            return LDB_ERR_HIP;
         }
         if (!all_ok && status != LDB_TRACE_MISSED) { // a peer mis-speculated: what it sent here is void too
            status = LDB_TRACE_MISSED;
            p->peerMisses++;
         }
      }
      if (status == LDB_TRACE_MISSED) { // a replayed count was wrong: everything computed from it is void
         if (out) ldb_gpu_table_release(p->ctx, out);
         p->misses++;
         p->key.clear();
         continue;
      }
      if (st != LDB_OK) {
         p->key.clear();
         return st;
      }
      if (status == LDB_TRACE_REPLAYED) {
         p->replays++;
         p->issue_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
         p->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      }
      p->key = key;
      *result = out;
      return LDB_OK;
   }
   g_plan_json_err = "plan_execute: the execution could not be completed after a failed replay";
   return LDB_ERR_INVALID;
}
extern "C" int32_t ldb_plan_times(const ldb_plan* p, double* issue_ms, double* wait_ms) {
   if (!p) return LDB_ERR_INVALID;
   if (issue_ms) *issue_ms = p->issue_ms;
   if (wait_ms) *wait_ms = p->wait_ms;
   return LDB_OK;
}
extern "C" int32_t ldb_plan_stats(const ldb_plan* p, int64_t* executions, int64_t* replays, int64_t* misses, int64_t* readbacks) {
   if (!p) return LDB_ERR_INVALID;
   if (executions) *executions = p->executions;
   if (replays) *replays = p->replays;
   if (misses) *misses = p->misses;
   if (readbacks) ldb_gpu_trace_stats(p->trace, readbacks, nullptr, nullptr, nullptr);
   return LDB_OK;
}
extern "C" const char* ldb_plan_json_last_error(void) { return g_plan_json_err.c_str(); }

// Device-less check of a plan's structure (what an emitter can verify before shipping a plan): the text
// parses, every step has a known "op" with its required fields, every value a step reads was named as an
// input or produced by an earlier step, no value is produced twice, and the result is produced by a step.
// Column names and types are only known at run time (ldb_plan_run_json).
extern "C" int32_t ldb_plan_json_check(const char* plan_json, const char* const* input_names, int32_t n_inputs) {
   if (!plan_json || n_inputs < 0) {
      g_plan_json_err = "plan_json_check: bad argument";
      return LDB_ERR_INVALID;
   }
   try {
      JParser parser(plan_json);
      const J plan = parser.value();
      if (plan.kind != J::OBJ) throw std::runtime_error("plan: the top level must be an object");
      std::map<std::string, bool> known; // name → produced by a step (false: an input)
      for (int32_t i = 0; i < n_inputs; i++) known[input_names[i]] = false;
      if (const J* in = plan.get("inputs"))
         for (auto& e : in->arr)
            if (!known.count(e.str)) {
               if (n_inputs > 0) throw std::runtime_error("plan: input '" + e.str + "' is not provided");
               known[e.str] = false;
            }
      struct Shape {
         const char* op;
         std::vector<const char*> reads, needs;
      };
      static const std::vector<Shape> shapes = {{"scan", {"table"}, {}},           {"filter", {"in"}, {"preds"}},        {"filter_dnf", {"in"}, {"clauses"}},
                                                {"join_build", {"in"}, {"keys"}},  {"join_probe", {"ht", "in"}, {"keys"}}, {"groupby", {"in"}, {"aggs"}},
                                                {"map", {"in"}, {"as"}},           {"sort", {"in"}, {"by"}},             {"topk", {"in"}, {"by", "k"}},
                                                {"materialize", {"in"}, {"cols"}}, {"join_nl", {"in", "build"}, {}}, {"allgather", {"in"}, {}},           {"shuffle", {"in"}, {"keys", "cols"}},
                                                {"nested_map", {"in", "scan"}, {}}, {"set_op", {"left", "right"}, {"left_cols", "right_cols", "kind"}}, {"window", {"in"}, {"fns"}}};
      const J& steps = plan.at("steps");
      if (steps.kind != J::ARR) throw std::runtime_error("plan: 'steps' must be an array");
      std::function<void(const J&, std::map<std::string, bool>&)> checkSteps = [&](const J& list, std::map<std::string, bool>& known) {
      for (auto& st : list.arr) {
         const std::string& op = st.s("op");
         if (op == "loop") { // variables bound from existing tables, a body with its own scope, results defined afterwards
            std::map<std::string, bool> inner = known;
            std::vector<std::string> vars;
            for (auto& v : st.at("vars").arr) {
               if (!known.count(v.s("init"))) throw std::runtime_error("plan: loop variable '" + v.s("name") + "' starts from '" + v.s("init") + "' before it exists");
               if (inner.count(v.s("name"))) throw std::runtime_error("plan: value '" + v.s("name") + "' defined twice");
               inner[v.s("name")] = true;
               vars.push_back(v.s("name"));
            }
            if (st.at("body").kind != J::ARR) throw std::runtime_error("plan: loop 'body' must be an array of steps");
            checkSteps(st.at("body"), inner);
            if (!inner.count(st.at("continue").s("from"))) throw std::runtime_error("plan: loop condition reads '" + st.at("continue").s("from") + "' before it exists");
            (void) st.at("continue").s("col");
            for (auto& n : st.at("next").arr) {
               if (std::find(vars.begin(), vars.end(), n.s("var")) == vars.end()) throw std::runtime_error("plan: loop 'next' names the unknown variable '" + n.s("var") + "'");
               auto it = inner.find(n.s("from"));
               if (it == inner.end() || known.count(n.s("from"))) throw std::runtime_error("plan: loop 'next' value '" + n.s("from") + "' is not produced by the body");
            }
            for (auto& r : st.at("results").arr) {
               if (std::find(vars.begin(), vars.end(), r.s("var")) == vars.end()) throw std::runtime_error("plan: loop result names the unknown variable '" + r.s("var") + "'");
               if (known.count(r.s("name"))) throw std::runtime_error("plan: value '" + r.s("name") + "' defined twice");
               known[r.s("name")] = true;
            }
            continue;
         }
         const Shape* sh = nullptr;
         for (auto& c : shapes)
            if (op == c.op) sh = &c;
         if (!sh) throw std::runtime_error("plan: unknown step '" + op + "'");
         const std::string& out = st.s("out");
         for (const char* r : sh->reads)
            if (!known.count(st.s(r))) throw std::runtime_error("plan: step '" + op + "' → '" + out + "' reads '" + st.s(r) + "' before it exists");
         for (const char* f : sh->needs) (void) st.at(f);
         if (op == "map" && !st.get("expr") && !st.get("fn")) throw std::runtime_error("plan: map → '" + out + "' needs 'expr' or 'fn'");
         if (known.count(out)) throw std::runtime_error("plan: value '" + out + "' defined twice");
         known[out] = true;
         if (const J* mo = st.get("mark_out")) known[mo->str] = true;
      }
      };
      checkSteps(steps, known);
      const std::string result = plan.sOr("result", "result");
      auto it = known.find(result);
      if (it == known.end() || !it->second) throw std::runtime_error("plan: result '" + result + "' is not produced by a step");
      return LDB_OK;
   } catch (const std::exception& e) {
      g_plan_json_err = e.what();
      return LDB_ERR_INVALID;
   }
}
