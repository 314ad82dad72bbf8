// ldb_subop.cpp — consumer of the reference's OWN sub-operator dump.
//
// LingoDB's CPU backend walks `subop.execution_step`s (handleExecutionStepCPU, SubOpToControlFlow.cpp:4363-4395;
// the GPU stub handleExecutionStepGPU, :4512-4548, is where a device backend plugs in) and the repository ships a
// tool that prints exactly that IR as JSON: tools/ct/mlir-subop-to-json.cpp (`[{"type":"execution_step",
// "subops":[{"subop":"get_external","meta":{tableName,mapping,filters}}, {"subop":"scan","mapping":[…]}, …],
// "inputs","results","outerEdges","innerEdges"}]`, :434-560 and the sub-operator cases :560-850).  This file
// reads that document and pattern-matches the sub-operator sequences the reference's lowerings produce onto
// the C-ABI's operator-level steps (the step list ldb_plan.cpp interprets):
//
//   get_external + scan                         → a base table relation; meta.filters → restrictions
//   map                                         → column bindings (expressions are inlined into their consumers; runtime calls
//                                                 ExtractYearFromDate / Substring → map fn, ConstLike → LIKE restrictions)
//   filter all_true                             → restrictions (conjunctions, IN, LIKE, DNF → filter_dnf, anything else → a computed
//                                                 boolean column + `= true`) / the conjuncts of a hash join (keys, residuals, post-join filters)
//   lookup(SimpleState) | lookup_or_insert(HashMap) + reduce [+ create_thread_local / merge]
//                                               → groupby (sum / count / count(*) / min / max / any, the nullable bodies; sum ÷ count → avg;
//                                                 an empty reduce → distinct)
//   materialize(Buffer) + create_hash_indexed_view + lookup(HashIndexedView) + nested_map{scan_list, gather,
//     combine_tuple, map, filter …}             → join_build + join_probe: inner; anyTuple + marker filter → semi / anti, read as a value →
//                                                 mark; flag member + scatter → semi_build / anti_build; + null / as-nullable maps + union →
//                                                 left_outer, right_outer (reverseSides), full_outer
//   create_simple_state + scatter / lookup + gather → the cross product with a one-row side (constant single join)
//   lookup(HashMap) + unwrap_optional_ref + … + reduce into another input's map → group join
//   two inputs counting into one map / two maps + union → set_op (UNION [ALL], INTERSECT [ALL], EXCEPT [ALL])
//   sorted / continuous / segment-tree views + scan_ref, frame references, entries_between, lookup(SegmentTreeView) → window
//   a buffer scanned inside a nested_map body   → join_nl
//   materialize(Buffer) + create_sorted_view    → sort          materialize(Heap) + scan → topk
//   materialize(ResultTable)                    → materialize (the query result)
// (INTEGRATION.md §1b has the table with the lowering each row comes from.)
//
// Everything else is reported per execution step as "cpu" with the reason (the reference would run such a step on
// its CPU backend; there is none here, so the translation as a whole fails with LDB_ERR_UNSUPPORTED and the report
// says which step).  The dump loses a few facts a backend needs; the consumer relies on small emitter additions, listed in
// INTEGRATION.md §1b and in tools/write_subop_dumps.py / tools/subop_lower.py (E1 " - " for db.sub, E4 sortBy / maxRows on
// create_sorted_view / create_heap, E5 primaryKey, E6 combine_tuple, E7 the arith.* ops of the nullable aggregate bodies / set-operation
// counters / rank, E8 the aggregates of create_segment_tree_view and the keys of lookup, E9 the materialize inside the reduce that
// fills a window partition's buffer).  Group-by keys need no
// extension: performAggregation names key members "keyval$n" and the later scan of the map re-defines the SAME
// columns (RelAlgToSubOp.cpp:2158-2166, test/lit/RelAlg/lowering.mlir:37), so keys are read off that scan.
#include "ldb_host.hpp"
#include "ldb_json.hpp"
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>

namespace {
using ldbjson::J;
using ldbjson::JParser;
using ldbjson::quote;

struct Unsupported : std::runtime_error {
   using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------ expressions (the tool's expression_leaf / expression_inner)
struct Expr;
using ExprP = std::shared_ptr<Expr>;
struct Expr {
   enum Kind { COL, CONST_INT, CONST_STR, CONST_BOOL, MEMBER, REF, HASH, MARKER, FLAG, NULLV, SETCNT, UNKNOWN, OP } kind = UNKNOWN; // SETCNT: one of the two counters of INTERSECT / EXCEPT (i = 1 | 2) // MARKER: "a partner exists" of a probe row; FLAG: of a build row; NULLV: db.null
   std::string name; // COL: column name in the plan language; MEMBER: member; OP: add sub mul div cast cmp and or not isnull select between in call (cmp = the runtime function)
   std::string cmp; // OP cmp: EQ NEQ LT LTE GT GTE; constants: the dump's data_type
   int64_t i = 0;
   bool build = false; // COL gathered from the build side of the hash join being matched
   bool base = false; // COL of a scanned base table (its type is the table's: column-vs-column restrictions compare like with like)
   std::string dtype; // COL: the dump's datatype of the column it was first defined as ("decimal(12,2)", "int32", …; may be empty)
   std::vector<ExprP> args;
};
ExprP mk(Expr::Kind k, const std::string& name = "") {
   auto e = std::make_shared<Expr>();
   e->kind = k;
   e->name = name;
   return e;
}
ExprP stripCast(ExprP e) {
   while (e->kind == Expr::OP && e->name == "cast") e = e->args[0];
   return e;
}
std::string exprJson(const ExprP& e0) { // the plan language of ldb_plan.cpp (compileX)
   const ExprP e = stripCast(e0); // db.cast is implicit there: operands are typed by the reference's decimal rules
   switch (e->kind) {
      case Expr::COL: return quote(e->name);
      case Expr::CONST_INT: return std::to_string(e->i);
      case Expr::CONST_STR: return quote(e->name);
      case Expr::OP: {
         if (e->name == "cmp") return "{\"cmp\": [" + quote(e->cmp) + ", " + exprJson(e->args[0]) + ", " + exprJson(e->args[1]) + "]}";
         if (e->name == "select") return "{\"case\": [" + exprJson(e->args[0]) + ", " + exprJson(e->args[1]) + ", " + exprJson(e->args[2]) + "]}";
         if (e->name == "add" || e->name == "sub" || e->name == "mul" || e->name == "div" || e->name == "and" || e->name == "or" || e->name == "not" || e->name == "isnull" || e->name == "coalesce") {
            std::string o = "{" + quote(e->name) + ": [";
            if ((e->name == "and" || e->name == "or" || e->name == "mul") && e->args.size() > 2) { // n-ary in the dump, flattened for mul, nested for and/or
               if (e->name == "mul") {
                  for (size_t k = 0; k < e->args.size(); k++) o += (k ? ", " : "") + exprJson(e->args[k]);
                  return o + "]}";
               }
               ExprP acc = e->args[0];
               for (size_t k = 1; k < e->args.size(); k++) {
                  ExprP n = mk(Expr::OP, e->name);
                  n->args = {acc, e->args[k]};
                  acc = n;
               }
               return exprJson(acc);
            }
            for (size_t k = 0; k < e->args.size(); k++) o += (k ? ", " : "") + exprJson(e->args[k]);
            return o + "]}";
         }
         throw Unsupported("expression '" + e->name + "' has no device form");
      }
      default: throw Unsupported("expression leaf without a device form");
   }
}
// (a * b) * c → one n-ary product, the form the aggregate normal form of ldb_plan.cpp recognises
ExprP flattenMul(const ExprP& e) {
   if (e->kind != Expr::OP) return e;
   auto r = std::make_shared<Expr>(*e);
   r->args.clear();
   for (auto& a : e->args) {
      ExprP f = flattenMul(a);
      if (e->name == "mul" && f->kind == Expr::OP && f->name == "mul")
         for (auto& g : f->args) r->args.push_back(g);
      else
         r->args.push_back(f);
   }
   return r;
}

// ------------------------------------------------------------------ what the walk keeps
struct Stream {
   std::string rel; // plan value holding the rows
   bool bareTable = false; // `rel` is an input table nothing has been applied to yet (its restrictions are in `preds`)
   std::vector<std::string> preds; // restrictions not yet applied (JSON objects of the plan language)
   std::map<std::string, ExprP> cols; // column displayName ("lineitem::l_quantity") → what it is
   std::set<std::string> names; // every column name the relation physically has (the plan language resolves columns by name)
   std::vector<std::set<std::string>> unique; // column sets known to identify a row
   std::string probeHiv, aggState; // the hash-indexed view being probed / the state being reduced into
   // the equalities of the hash join being matched are known, the join kind is not yet: inner unless the marker idiom of a
   // semi / anti join follows (RelAlgToSubOp.cpp:1296-1375)
   bool pending = false;
   std::vector<std::string> probeKeys, buildKeys;
   std::vector<bool> keyDecimal; // per key pair: a decimal column is involved (kept as a residual equality when an integer key exists)
   std::string flagState; // scan of a build buffer whose flag member a semi / anti join with reversed sides has set
   std::string nlBuild; // nested-loop join (translateNLJ): the buffer being scanned once per tuple of this stream
   bool inJoinBody = false; // between the gather of a hash join and the end of its nested_map: filters are conjuncts of the join predicate
   std::vector<std::string> residual; // non-equality conjuncts relating the two sides ({"probe", "op", "build"})
   std::vector<ExprP> postJoin; // conjuncts of the join predicate that do not relate one probe with one build column: applied to the joined rows (inner joins only)
   bool flagAntiPending = false; // a flagged buffer scanned for its UNflagged rows in a step that also has a union: the unmatched half of a reversed outer join (when a map of NULLs follows) or an ordinary anti join of the build side (anything else)
   bool fullOuter = false; // the matches and the partner-less probe rows of a FULL outer join: united with the unmatched build rows after the probe pipeline
   bool antiBranch = false; // outer / single join: the branch of the probe rows WITHOUT a partner (filter none_true on the marker)
   std::string constState; // constant single join: the one-row state this stream looked up
   std::vector<std::string> lastMapped; // the columns the most recent map defined (UNION ALL maps both inputs to the result columns)
   // a window being evaluated on this stream: the continuous view it scans, the functions collected so far and their common frame
   std::string winView, winLookup, winStatic; // winStatic: a plain scan of the view feeding the whole-partition aggregation of a frame unbounded on both sides
   struct WinFn {
      std::string fn, col, as;
   };
   std::vector<WinFn> winFns;
   bool winFrame = false, winHasTo = false;
   int64_t winFrom = 0, winTo = 0;
   std::string partState; // scan of the map of per-partition buffers: the buffer column stands for this state inside the nested_map
   std::string gjState; // group join: this stream probes the map the other input created (GroupJoinLowering, RelAlgToSubOp.cpp:2682-2950)
   std::string setState; // scan of the counting map of INTERSECT / EXCEPT: the set operation is emitted when its predicate / repeat count is consumed
};
struct AggSpec {
   std::string member, fn;
   ExprP arg; // null for count(*)
};
struct OutAgg {
   std::string fn, expr, as;
   std::string when, type; // conditional aggregate: sum(case when <conjunction> then x else 0 end); result type override
   bool used = false;
};
struct OutStep {
   std::string op, out;
   std::vector<std::pair<std::string, std::string>> fields; // key → raw JSON, in order
   std::vector<OutAgg> aggs; // groupby
};
struct State {
   enum Kind { UNKNOWN, TABLE, AGG, BUFFER, HIV, SORTED, HEAP, RESULT, MARKER, CONST1, CONTVIEW, SEGTREE } kind = UNKNOWN; // CONST1: the scattered one-row state of a constant single join; CONTVIEW / SEGTREE: the views of a window
   std::string table; // TABLE
   std::map<std::string, std::string> memberToIdent;
   std::vector<std::string> filters, pkey;
   Stream in; // AGG: the reduced stream; BUFFER/HEAP/RESULT/SORTED: the materialised one
   std::vector<AggSpec> aggs;
   int groupbyStep = -1; // AGG: index of the emitted groupby (emitted when the state is first scanned)
   std::map<std::string, std::string> aggOut; // AGG: member → output column
   std::map<std::string, ExprP> members; // BUFFER …: member → what was stored
   std::shared_ptr<State> source; // HIV / SORTED: the buffer underneath
   std::string ht; // HIV: the plan value of the built table
   std::vector<std::string> htKeys;
   bool htUnique = false;
   std::vector<std::pair<std::string, bool>> sortBy; // member, descending
   int64_t maxRows = -1;
   bool emitted = false; // SORTED / HEAP: the sort / topk step exists and in.rel names its output
   // BUFFER that is the build side of a semi / anti join with reversed sides: the flag member, the probing stream (kept
   // for the anti form, which is emitted when the flag filter says all_false) and the build rows with a partner
   std::string flagMember, semiRel, antiRel;
   std::shared_ptr<Stream> flagProbe;
   std::shared_ptr<State> flagHiv;
   // AGG reduced by TWO pipelines with nothing but row counters: the map of a UNION (distinct) / INTERSECT / EXCEPT
   // windows (WindowLowering, RelAlgToSubOp.cpp:2193-2553): SORTED remembers its order, a BUFFER filled inside the reduce of a map keyed by the
   // PARTITION BY columns remembers them, a SEGTREE its aggregates (emitter extension E8)
   std::vector<std::pair<std::string, bool>> sortCols; // column, descending
   std::vector<std::string> partCols;
   bool partPending = false; // BUFFER filled per partition: the keys are named by the scan of the map
   struct WinAgg {
      std::string member, fn, source;
   };
   std::vector<WinAgg> winAggs;
   // AGG whose map a second stream looks up and aggregates into: a group join
   std::shared_ptr<Stream> gjRight;
   std::vector<AggSpec> gjAggs;
   std::vector<ExprP> gjFilters;
   std::vector<std::pair<std::string, ExprP>> gjPairs, gjStored; // (left key column, the right column renamed to it); (gjval member, placeholder of the stored left column)
   bool gjInner = false;
   std::shared_ptr<Stream> setRight;
   std::string setLeftCounter, setRightCounter;
   std::vector<std::string> setLeftCols, setRightCols;
};
using StateP = std::shared_ptr<State>;

struct Translator {
   std::vector<OutStep> steps;
   std::map<std::string, StateP> states; // "<step ref>#<resnr>" → state
   std::map<std::string, std::pair<int, int>> aggCol; // output column name → (groupby step, agg index)
   std::set<std::string> inputs;
   std::string result;
   std::string markName; // the column a mark join about to be emitted adds
   std::string report; // JSON array body
   std::set<std::string> ext; // the emitter extensions the dump's manifest declares (integration/mlir-subop-to-json.patch)
   int nval = 0;

   std::string fresh(const char* stem) { return std::string(stem) + std::to_string(++nval); }
   static std::string sanitize(std::string s) {
      for (size_t p; (p = s.find("::")) != std::string::npos;) s.replace(p, 2, ".");
      return s;
   }
   static std::string stripSuffix(const std::string& m) { // "revenue$4" → "revenue"
      const size_t p = m.rfind('$');
      return p == std::string::npos ? m : m.substr(0, p);
   }

   // ---------------------------------------------------------------- expressions
   static bool strs(const J& e, std::initializer_list<const char*> want) {
      const J& s = e.at("strings");
      if (s.kind != J::ARR || s.arr.size() != want.size()) return false;
      size_t k = 0;
      for (const char* w : want)
         if (s.arr[k++].str != w) return false;
      return true;
   }
   ExprP convert(const J& e, const Stream& st) {
      const std::string& type = e.s("type");
      if (type == "expression_leaf") {
         const std::string& leaf = e.s("leaf_type");
         if (leaf == "column" || leaf == "external_column") {
            auto it = st.cols.find(e.s("displayName"));
            if (it == st.cols.end()) throw Unsupported("column '" + e.s("displayName") + "' is not defined on this stream");
            return it->second;
         }
         if (leaf == "constant") {
            const J& v = e.at("value");
            if (v.kind == J::NUM && v.isInt) {
               ExprP c = mk(Expr::CONST_INT);
               c->i = v.inum;
               c->cmp = e.sOr("data_type", "");
               return c;
            }
            if (v.kind == J::STR) return mk(Expr::CONST_STR, v.str);
            if (v.kind == J::BOOL) {
               ExprP c = mk(Expr::CONST_BOOL);
               c->i = v.b;
               return c;
            }
            throw Unsupported("floating-point constant");
         }
         if (leaf == "member") return mk(Expr::MEMBER, e.s("member"));
         if (leaf == "null") return mk(Expr::NULLV);
         return mk(Expr::UNKNOWN); // "unknown" / "null": only legal inside the recognised aggregate bodies
      }
      if (type != "expression_inner") throw Unsupported("expression node of type '" + type + "'");
      const J& subs = e.at("subExpressions");
      std::vector<ExprP> a;
      for (auto& s : subs.arr) a.push_back(convert(s, st));
      auto op = [&](const char* name, size_t n) {
         if (a.size() != n) throw Unsupported(std::string("malformed '") + name + "' expression");
         ExprP r = mk(Expr::OP, name);
         r->args = a;
         return r;
      };
      const J& ss = e.at("strings");
      const std::string first = ss.arr.empty() ? "" : ss.arr[0].str;
      if (strs(e, {"", " * ", ""})) return op("mul", 2);
      if (strs(e, {"", " + ", ""})) {
         // the unpatched tool prints db.sub with the separator " + " too (mlir-subop-to-json.cpp:315-317): a sum is a sum only from an emitter with E1
         if (!ext.count("E1")) throw Unsupported("' + ' is ambiguous without emitter extension E1: the unpatched tool prints db.sub as ' + ' as well (mlir-subop-to-json.cpp:315-317)");
         return op("add", 2);
      }
      if (strs(e, {"", " - ", ""})) return op("sub", 2); // emitter extension E1
      if (strs(e, {"", " / ", ""})) return op("div", 2);
      if (strs(e, {"cast(", ")"})) return op("cast", 1);
      if (strs(e, {"not ", ""})) return op("not", 1);
      if (strs(e, {"", " is null"})) return op("isnull", 1);
      if (strs(e, {"", " ? ", " : ", ""})) return op("select", 3);
      if (strs(e, {"if ", " then ", " else ", ""})) return op("select", 3);
      if (strs(e, {"", " between ", " and ", ""})) {
         // db.between carries lowerInclusive / upperInclusive (DBOps.td:507); `x >= a and x < b` is canonicalised into a HALF-OPEN one (DBOps.cpp:475-483,
         // lowered at LowerToStd.cpp:1026) and the unpatched tool drops both flags (mlir-subop-to-json.cpp:261-263): only an emitter with E10 says which
         if (a.size() != 3) throw Unsupported("malformed between");
         const J *li = e.get("lowerInclusive"), *ui = e.get("upperInclusive");
         if (!ext.count("E10") || !li || !ui || li->kind != J::BOOL || ui->kind != J::BOOL)
            throw Unsupported("db.between without its lowerInclusive / upperInclusive flags (emitter extension E10): the unpatched tool prints a half-open range like a closed one (mlir-subop-to-json.cpp:261-263)");
         ExprP lo = mk(Expr::OP, "cmp"), hi = mk(Expr::OP, "cmp"), r = mk(Expr::OP, "and");
         lo->cmp = li->b ? "GTE" : "GT", lo->args = {a[0], a[1]};
         hi->cmp = ui->b ? "LTE" : "LT", hi->args = {a[0], a[2]};
         r->args = {lo, hi};
         return r;
      }
      if (first == "hash(") { // hash(x) or hash(pack(x, y, …)): only the hashed columns matter
         ExprP h = mk(Expr::HASH);
         h->args = a.size() == 1 && a[0]->kind == Expr::OP && a[0]->name == "pack" ? a[0]->args : a;
         return h;
      }
      if (first == "pack(") {
         ExprP r = mk(Expr::OP, "pack");
         r->args = a;
         return r;
      }
      if (ss.arr.size() == 3 && ss.arr[0].str.empty() && ss.arr[2].str.empty() && a.size() == 2) { // db.compare: convertCmpPredicate, no spaces
         static const std::pair<const char*, const char*> cmps[] = {{"=", "EQ"}, {"<>", "NEQ"}, {"<", "LT"}, {"<=", "LTE"}, {">", "GT"}, {">=", "GTE"}, {"isa", "EQ"}};
         for (auto& c : cmps)
            if (ss.arr[1].str == c.first) {
               ExprP r = mk(Expr::OP, "cmp");
               r->cmp = c.second;
               r->args = a;
               return r;
            }
      }
      if (ss.arr.size() >= 3 && first.empty() && ss.arr[1].str == " and ") {
         ExprP r = mk(Expr::OP, "and");
         r->args = a;
         return r;
      }
      if (ss.arr.size() >= 3 && first == "(" && ss.arr[1].str == " or ") {
         ExprP r = mk(Expr::OP, "or");
         r->args = a;
         return r;
      }
      if (first.empty() && ss.arr.size() >= 3 && ss.arr[1].str == " in [") {
         ExprP r = mk(Expr::OP, "in");
         r->args = a;
         return r;
      }
      if (first.size() > 1 && first.back() == '(' && isalpha((unsigned char) first[0])) { // db.runtime_call: "Fn(" a ", " b ")" (mlir-subop-to-json.cpp:318-326)
         ExprP r = mk(Expr::OP, "call");
         r->cmp = first.substr(0, first.size() - 1);
         r->args = a;
         return r;
      }
      throw Unsupported("expression '" + first + "…' (operator without a device form)");
   }

   // ---------------------------------------------------------------- emission helpers
   void flush(Stream& s) { // apply pending restrictions as a filter step
      if (s.preds.empty()) {
         s.bareTable = false;
         return;
      }
      OutStep f;
      f.op = "filter";
      f.out = fresh("v");
      std::string p = "[";
      for (size_t k = 0; k < s.preds.size(); k++) p += (k ? ", " : "") + s.preds[k];
      f.fields = {{"in", quote(s.rel)}, {"preds", p + "]"}};
      steps.push_back(f);
      s.rel = f.out;
      s.preds.clear();
      s.bareTable = false;
   }
   void use(const std::string& col) {
      auto it = aggCol.find(col);
      if (it != aggCol.end()) steps[(size_t) it->second.first].aggs[(size_t) it->second.second].used = true;
   }
   void useAll(const ExprP& e) {
      if (e->kind == Expr::COL) use(e->name);
      for (auto& a : e->args) useAll(a);
   }
   // runtime calls nested inside an expression become columns first (the expression language is arithmetic / boolean only)
   ExprP materializeCalls(Stream& s, const ExprP& e, const std::string& hint) {
      if (e->kind != Expr::OP) return e;
      if (e->name == "call" && (e->cmp == "ExtractYearFromDate" || e->cmp == "Substring")) return mk(Expr::COL, ensureCol(s, e, hint + "_f" + std::to_string(++nval)));
      bool changed = false;
      std::vector<ExprP> args;
      for (auto& a : e->args) {
         ExprP n = materializeCalls(s, a, hint);
         changed = changed || n != a;
         args.push_back(n);
      }
      if (!changed) return e;
      auto r = std::make_shared<Expr>(*e);
      r->args = args;
      return r;
   }
   // a plain column holding `e` on the stream: a computed expression becomes a `map` step
   std::string ensureCol(Stream& s, const ExprP& e0, const std::string& hint) {
      const ExprP e = stripCast(e0);
      if (e->kind == Expr::COL) {
         use(e->name);
         return e->name;
      }
      OutStep m;
      m.op = "map";
      const std::string as = sanitize(hint);
      if (e->kind == Expr::OP && e->name == "call") { // the runtime functions with a device kernel: extract(year …) and substring
         const ExprP a0 = e->args.empty() ? nullptr : stripCast(e->args[0]);
         if (e->cmp == "ExtractYearFromDate" && e->args.size() == 1) {
            const std::string src = ensureCol(s, a0, hint + "_arg");
            flush(s);
            m.fields = {{"in", quote(s.rel)}, {"fn", "\"extract_year\""}, {"col", quote(src)}, {"as", quote(as)}};
         } else if (e->cmp == "Substring" && e->args.size() == 3 && stripCast(e->args[1])->kind == Expr::CONST_INT && stripCast(e->args[2])->kind == Expr::CONST_INT) {
            const std::string src = ensureCol(s, a0, hint + "_arg"); // StringRuntime::substr(str, from, len)
            flush(s);
            m.fields = {{"in", quote(s.rel)}, {"fn", "\"substr\""}, {"col", quote(src)}, {"from", std::to_string(stripCast(e->args[1])->i)}, {"for", std::to_string(stripCast(e->args[2])->i)}, {"as", quote(as)}};
         } else {
            throw Unsupported("runtime function '" + e->cmp + "' has no device form");
         }
      } else {
         const ExprP x = materializeCalls(s, e, hint);
         flush(s);
         useAll(x);
         m.fields = {{"in", quote(s.rel)}, {"expr", exprJson(flattenMul(x))}, {"as", quote(as)}};
      }
      m.out = fresh("v");
      steps.push_back(m);
      s.rel = m.out;
      s.names.insert(as);
      ExprP c = mk(Expr::COL, as);
      for (auto& kv : s.cols)
         if (kv.second == e0) kv.second = c;
      return as;
   }

   // a condition as a conjunction of the plan language's restrictions: column-vs-constant comparisons, AND of such, and an
   // OR of equalities on ONE column (→ IN); false when it is anything else
   bool condPreds(const ExprP& c0, std::vector<std::string>& out) {
      const ExprP c = stripCast(c0);
      if (c->kind != Expr::OP) return false;
      auto lit = [](const ExprP& k) { return k->kind == Expr::CONST_INT ? std::to_string(k->i) : quote(k->name); };
      auto isConst = [](const ExprP& k) { return k->kind == Expr::CONST_INT || k->kind == Expr::CONST_STR; };
      if (c->name == "and") {
         for (auto& a : c->args)
            if (!condPreds(a, out)) return false;
         return true;
      }
      if (c->name == "cmp") {
         ExprP l = stripCast(c->args[0]), r = stripCast(c->args[1]);
         std::string op = c->cmp;
         if (l->kind != Expr::COL && r->kind == Expr::COL) {
            std::swap(l, r);
            op = op == "LT" ? "GT" : op == "GT" ? "LT" : op == "LTE" ? "GTE" : op == "GTE" ? "LTE" : op;
         }
         if (l->kind != Expr::COL || !isConst(r)) return false;
         use(l->name);
         out.push_back("{\"col\": " + quote(l->name) + ", \"op\": " + quote(op) + ", \"value\": " + lit(r) + "}");
         return true;
      }
      if (c->name == "in") { // db.oneof: column in [constants]
         const ExprP l = stripCast(c->args[0]);
         if (l->kind != Expr::COL || c->args.size() < 2) return false;
         std::string vals;
         for (size_t k = 1; k < c->args.size(); k++) {
            const ExprP v = stripCast(c->args[k]);
            if (!isConst(v)) return false;
            vals += (vals.empty() ? "" : ", ") + lit(v);
         }
         use(l->name);
         out.push_back("{\"col\": " + quote(l->name) + ", \"op\": \"IN\", \"values\": [" + vals + "]}");
         return true;
      }
      if (c->name == "call" || (c->name == "not" && stripCast(c->args[0])->kind == Expr::OP && stripCast(c->args[0])->name == "call")) { // [not] ConstLike(column, pattern)
         const bool neg = c->name == "not";
         const ExprP f = neg ? stripCast(c->args[0]) : c;
         if ((f->cmp != "ConstLike" && f->cmp != "Like") || f->args.size() != 2) return false;
         const ExprP l = stripCast(f->args[0]), pat = stripCast(f->args[1]);
         if (l->kind != Expr::COL || pat->kind != Expr::CONST_STR) return false;
         use(l->name);
         out.push_back("{\"col\": " + quote(l->name) + ", \"op\": " + (neg ? "\"NOT LIKE\"" : "\"LIKE\"") + ", \"value\": " + quote(pat->name) + "}");
         return true;
      }
      if (c->name == "not" && stripCast(c->args[0])->kind == Expr::OP && stripCast(c->args[0])->name == "isnull") { // not (x is null)
         const ExprP l = stripCast(stripCast(c->args[0])->args[0]);
         if (l->kind != Expr::COL) return false;
         use(l->name);
         out.push_back("{\"col\": " + quote(l->name) + ", \"op\": \"NOTNULL\"}");
         return true;
      }
      if (c->name == "or") {
         std::string col, vals;
         for (auto& a0 : c->args) {
            const ExprP a = stripCast(a0);
            if (!(a->kind == Expr::OP && a->name == "cmp" && a->cmp == "EQ")) return false;
            ExprP l = stripCast(a->args[0]), r = stripCast(a->args[1]);
            if (l->kind != Expr::COL) std::swap(l, r);
            if (l->kind != Expr::COL || !isConst(r) || (!col.empty() && col != l->name)) return false;
            col = l->name;
            vals += (vals.empty() ? "" : ", ") + lit(r);
         }
         if (col.empty()) return false;
         use(col);
         out.push_back("{\"col\": " + quote(col) + ", \"op\": \"IN\", \"values\": [" + vals + "]}");
         return true;
      }
      return false;
   }

   // ---------------------------------------------------------------- states
   static bool isKeyMember(const std::string& m) { return m.rfind("keyval", 0) == 0 || m.rfind("gjkeyval", 0) == 0; } // performAggregation / GroupJoinLowering member names
   StateP emitGroupBy(const StateP& st, const J& mapping) {
      if (st->groupbyStep >= 0) return st;
      Stream& in = st->in;
      OutStep g;
      g.op = "groupby";
      g.out = fresh("g");
      std::vector<std::string> keys;
      for (auto& m : mapping.arr) {
         const std::string& member = m.s("member");
         if (!isKeyMember(member)) continue;
         // the scan re-defines the grouped column itself: its binding on the reduced stream is the key
         auto it = in.cols.find(m.at("column").s("displayName"));
         if (it == in.cols.end()) throw Unsupported("group key '" + m.at("column").s("displayName") + "' is not a column of the aggregated stream");
         keys.push_back(ensureCol(in, it->second, m.at("column").s("displayName")));
      }
      std::string kj = "[";
      for (size_t k = 0; k < keys.size(); k++) kj += (k ? ", " : "") + quote(keys[k]);
      kj += "]";
      for (auto& a : st->aggs) {
         OutAgg o;
         o.fn = a.fn;
         if (a.arg) {
            const ExprP sel = stripCast(a.arg);
            std::vector<std::string> when;
            // sum(case when c then x else 0 end): a conditional aggregate (the reference computes the case in a map before the reduce)
            if (a.fn == "sum" && sel->kind == Expr::OP && sel->name == "select" && stripCast(sel->args[2])->kind == Expr::CONST_INT && stripCast(sel->args[2])->i == 0 &&
                condPreds(sel->args[0], when)) {
               const ExprP then = stripCast(sel->args[1]);
               useAll(then);
               o.expr = exprJson(flattenMul(then));
               o.when = "[";
               for (size_t k = 0; k < when.size(); k++) o.when += (k ? ", " : "") + when[k];
               o.when += "]";
               if (then->kind == Expr::CONST_INT && then->cmp == "int32") o.type = "int32"; // integer literals are int32 and SUM keeps the type
            } else {
               useAll(a.arg);
               o.expr = exprJson(flattenMul(a.arg));
            }
         }
         std::string as;
         for (auto& m : mapping.arr)
            if (m.s("member") == a.member) as = sanitize(m.at("column").s("displayName"));
         o.as = as.empty() ? sanitize(st->in.rel + "." + a.member) : as;
         o.used = a.member == "distinct$count"; // (a distinct projection reads no aggregate: its group-by keeps the row count)
         st->aggOut[a.member] = o.as;
         g.aggs.push_back(o);
      }
      g.fields = {{"in", quote(in.rel)}, {"keys", kj}};
      if (keys.empty()) g.fields.push_back({"est_groups", "1"});
      if (!in.preds.empty()) {
         if (!in.bareTable) {
            flush(in);
            g.fields[0].second = quote(in.rel);
         } else { // restrictions of the scanned table fuse into the aggregation kernel (scan → filter → aggregate in one pass)
            std::string p = "[";
            for (size_t k = 0; k < in.preds.size(); k++) p += (k ? ", " : "") + in.preds[k];
            g.fields.push_back({"preds", p + "]"});
         }
      }
      st->groupbyStep = (int) steps.size();
      for (size_t k = 0; k < g.aggs.size(); k++) aggCol[g.aggs[k].as] = {st->groupbyStep, (int) k};
      steps.push_back(g);
      Stream out;
      out.rel = g.out;
      out.names.insert(keys.begin(), keys.end());
      for (auto& a : g.aggs) out.names.insert(a.as);
      if (!keys.empty()) out.unique.push_back(std::set<std::string>(keys.begin(), keys.end()));
      size_t kk = 0;
      for (auto& m : mapping.arr) {
         const std::string& member = m.s("member");
         if (isKeyMember(member)) st->aggOut[member] = keys[kk++];
      }
      st->in = out; // from here on the state IS the aggregated table
      return st;
   }

   // the set operation whose counting map `s` scans: distinct kinds keep one row per key, the ALL kinds min(c1, c2) / max(c1 - c2, 0) rows
   void emitSetOp(Stream& s, const std::string& kind) {
      StateP st = states.at(s.setState);
      flush(st->in);
      flush(*st->setRight);
      OutStep u;
      u.op = "set_op";
      u.out = fresh("u");
      u.fields = {{"kind", quote(kind)}, {"left", quote(st->in.rel)}, {"left_cols", nameList(st->setLeftCols)}, {"right", quote(st->setRight->rel)}, {"right_cols", nameList(st->setRightCols)}};
      steps.push_back(u);
      s.rel = u.out;
      s.setState.clear();
      for (auto it = s.cols.begin(); it != s.cols.end();)
         it = it->second->kind == Expr::SETCNT || (it->second->kind == Expr::OP && it->second->name == "setop") ? s.cols.erase(it) : std::next(it);
   }
   // what a map over the two counters computes (CountingSetOperationLowering, :868-915; arith.cmpi / andi / subi / select: emitter extension E7)
   static std::string classifySet(const ExprP& e0) {
      const ExprP e = stripCast(e0);
      auto cnt = [](const ExprP& x, int which) { return stripCast(x)->kind == Expr::SETCNT && stripCast(x)->i == which; };
      auto zero = [](const ExprP& x) { return stripCast(x)->kind == Expr::CONST_INT && stripCast(x)->i == 0; };
      auto diff = [&](const ExprP& x) { return stripCast(x)->kind == Expr::OP && stripCast(x)->name == "sub" && cnt(stripCast(x)->args[0], 1) && cnt(stripCast(x)->args[1], 2); };
      if (e->kind != Expr::OP) return "";
      if (e->name == "and" && e->args.size() == 2) {
         const ExprP l = stripCast(e->args[0]), r = stripCast(e->args[1]);
         if (l->kind == Expr::OP && l->name == "cmp" && l->cmp == "GT" && cnt(l->args[0], 1) && zero(l->args[1]) && r->kind == Expr::OP && r->name == "cmp" && cnt(r->args[0], 2) && zero(r->args[1]))
            return r->cmp == "GT" ? "intersect" : r->cmp == "EQ" ? "except" : "";
      }
      if (e->name == "select" && e->args.size() == 3) {
         const ExprP c = stripCast(e->args[0]);
         if (c->kind == Expr::OP && c->name == "cmp" && c->cmp == "LT" && diff(c->args[0]) && zero(c->args[1]) && zero(e->args[1]) && diff(e->args[2])) return "except_all";
         if (c->kind == Expr::OP && c->name == "cmp" && c->cmp == "GT" && cnt(c->args[0], 1) && cnt(c->args[1], 2) && cnt(e->args[1], 2) && cnt(e->args[2], 1)) return "intersect_all";
      }
      return "";
   }

   // ---------------------------------------------------------------- one execution step
   struct StepCtx {
      std::vector<StateP> args;
      std::map<std::string, StateP> local; // "<ref>#<resnr>" of states created inside the step
      std::map<std::string, Stream> streams; // by producing sub-operator
      Stream* nested = nullptr; // the stream a nested_map hands to its body
      bool outerBody = false; // the nested_map body being walked ends in a union: an outer / single join (RelAlgToSubOp.cpp:1486-1587)
      bool outerStep = false; // the step itself has a union: an outer join with reverseSides unites the matches with the unmatched build rows
   };
   StateP resolve(const J& acc, StepCtx& c) {
      const std::string& t = acc.s("type");
      if (t == "parentArg") {
         const int64_t n = acc.iOr("argnr", -1);
         if (n < 0 || (size_t) n >= c.args.size() || !c.args[(size_t) n]) throw Unsupported("state argument " + std::to_string(n) + " is not available");
         return c.args[(size_t) n];
      }
      if (t == "node") {
         const std::string id = acc.s("ref") + "#" + std::to_string(acc.iOr("resnr", 0));
         auto it = c.local.find(id);
         if (it != c.local.end()) return it->second;
         auto jt = states.find(id);
         if (jt != states.end()) return jt->second;
         throw Unsupported("state '" + id + "' is produced by a sub-operator this consumer does not know");
      }
      if (t == "nested_map_arg" && c.nested && !c.nested->partState.empty()) return states.at(c.nested->partState); // the per-partition buffer of a window
      throw Unsupported("state access of type '" + t + "'");
   }
   Stream& input(const J& op, StepCtx& c) {
      for (auto& e : op.at("outerEdges").arr)
         if (e.s("type") == "stream") {
            auto it = c.streams.find(e.at("input").s("ref"));
            if (it == c.streams.end()) throw Unsupported("stream input '" + e.at("input").s("ref") + "' is not produced in this step");
            return it->second;
         }
      if (c.nested) return *c.nested;
      throw Unsupported("sub-operator without a stream input");
   }
   static std::string predJson(const std::string& col, const std::string& op, const J* value, const J* values) {
      std::string p = "{\"col\": " + quote(col) + ", \"op\": " + quote(op);
      auto lit = [](const J& v) { return v.kind == J::STR ? quote(v.str) : v.kind == J::NUM && v.isInt ? std::to_string(v.inum) : std::to_string(v.num); };
      if (values) {
         p += ", \"values\": [";
         for (size_t k = 0; k < values->arr.size(); k++) p += (k ? ", " : "") + lit(values->arr[k]);
         p += "]";
      } else if (value && op != "NOTNULL") {
         p += ", \"value\": " + lit(*value);
      }
      return p + "}";
   }

   // which side of the hash join being matched an expression reads: 0 nothing, 1 probe, 2 build, 3 both
   static int sideOf(const ExprP& e) {
      int r = e->kind == Expr::COL ? (e->build ? 2 : 1) : 0;
      for (auto& a : e->args) r |= sideOf(a);
      return r;
   }
   static std::string mirrored(const std::string& op) { return op == "LT" ? "GT" : op == "GT" ? "LT" : op == "LTE" ? "GTE" : op == "GTE" ? "LTE" : op; }
   // one filter inside the nested_map body of a hash join = conjunct(s) of the join predicate (translateSelection emits one
   // map + filter per conjunct, RelAlgToSubOp.cpp:173-258): probe = build equalities are the keys, other comparisons
   // between the sides the residual of the probe, anything else is applied to the joined rows
   void addJoinConjunct(Stream& s, const ExprP& pred) {
      StateP hiv = states.at(s.probeHiv);
      std::vector<ExprP> conj;
      std::function<void(const ExprP&)> split = [&](const ExprP& e0) {
         const ExprP e = stripCast(e0);
         if (e->kind == Expr::OP && e->name == "and")
            for (auto& a : e->args) split(a);
         else
            conj.push_back(e);
      };
      split(pred);
      Stream& b = hiv->source->in;
      for (auto& e : conj) {
         if (e->kind == Expr::OP && e->name == "cmp") {
            ExprP l = stripCast(e->args[0]), r = stripCast(e->args[1]);
            std::string op = e->cmp;
            if (sideOf(l) == 2 && sideOf(r) == 1) {
               std::swap(l, r);
               op = mirrored(op);
            }
            if (sideOf(l) == 1 && sideOf(r) == 2) {
               const std::string bc = ensureCol(b, r, "build_key"), pc = ensureCol(s, l, "probe_key");
               if (op == "EQ") {
                  s.buildKeys.push_back(bc);
                  s.probeKeys.push_back(pc);
                  s.keyDecimal.push_back(l->dtype.find("decimal") != std::string::npos || r->dtype.find("decimal") != std::string::npos);
               } else {
                  s.residual.push_back("{\"probe\": " + quote(pc) + ", \"op\": " + quote(op) + ", \"build\": " + quote(bc) + "}");
               }
               continue;
            }
         }
         s.postJoin.push_back(e);
      }
      s.pending = true;
   }
   // the plan language resolves columns by name: when the rows of `buf` are about to be joined to a stream that already has
   // columns of the same names (self joins, the same table inside a subquery, a group key named like its input column), the
   // buffer's columns are re-materialised under fresh names first
   void separateNames(const Stream& s, const StateP& buf, bool built) {
      Stream& b = buf->in;
      bool clash = false;
      for (auto& n : b.names) clash = clash || s.names.count(n);
      if (!clash) return;
      if (built) throw Unsupported("column names of the two join sides collide and the hash table is already built");
      std::vector<std::pair<std::string, std::string>> cols; // member → current column
      for (auto& kv : buf->members) {
         const Expr::Kind k = stripCast(kv.second)->kind;
         if (k == Expr::HASH || k == Expr::REF || k == Expr::CONST_BOOL || k == Expr::FLAG || k == Expr::MARKER || k == Expr::NULLV) continue;
         cols.push_back({kv.first, ensureCol(b, kv.second, stripSuffix(kv.first))});
      }
      flush(b);
      std::map<std::string, std::string> renamed;
      std::string list = "[";
      for (auto& mc : cols) {
         if (renamed.count(mc.second)) continue;
         const std::string to = s.names.count(mc.second) ? mc.second + "_" + std::to_string(++nval) : mc.second;
         renamed[mc.second] = to;
         list += std::string(list.size() > 1 ? ", " : "") + (to == mc.second ? quote(to) : "{\"col\": " + quote(mc.second) + ", \"as\": " + quote(to) + "}");
      }
      OutStep m;
      m.op = "materialize";
      m.out = fresh("v");
      m.fields = {{"in", quote(b.rel)}, {"cols", list + "]"}};
      steps.push_back(m);
      b.rel = m.out;
      b.bareTable = false;
      b.names.clear();
      for (auto& r : renamed) b.names.insert(r.second);
      for (auto& mc : cols) buf->members[mc.first] = mk(Expr::COL, renamed[mc.second]);
      std::vector<std::set<std::string>> uniq;
      for (auto& u : b.unique) {
         std::set<std::string> r;
         bool all = true;
         for (auto& k : u) {
            auto it = renamed.find(k);
            all = all && it != renamed.end();
            if (it != renamed.end()) r.insert(it->second);
         }
         if (all) uniq.push_back(r);
      }
      b.unique = uniq;
   }
   static std::string nameList(const std::vector<std::string>& v) {
      std::string o = "[";
      for (size_t k = 0; k < v.size(); k++) o += (k ? ", " : "") + quote(v[k]);
      return o + "]";
   }
   // join_build (once per view) + join_probe of the pending join with the given kind; returns the output relation
   std::string emitJoin(Stream& s, const std::string& kind) {
      StateP hiv = states.at(s.probeHiv);
      Stream& b = hiv->source->in;
      if (s.probeKeys.empty()) throw Unsupported("hash join without an equality between the sides");
      if (s.keyDecimal.size() == s.probeKeys.size() && std::count(s.keyDecimal.begin(), s.keyDecimal.end(), false) > 0 && std::count(s.keyDecimal.begin(), s.keyDecimal.end(), true) > 0) {
         // integer keys hash; an equality of decimals (a correlated MIN / MAX joined back, Q2) stays a residual comparison of the pair
         std::vector<std::string> pk, bk;
         for (size_t k = 0; k < s.probeKeys.size(); k++) {
            if (!s.keyDecimal[k]) {
               pk.push_back(s.probeKeys[k]);
               bk.push_back(s.buildKeys[k]);
            } else {
               s.residual.push_back("{\"probe\": " + quote(s.probeKeys[k]) + ", \"op\": \"EQ\", \"build\": " + quote(s.buildKeys[k]) + "}");
            }
         }
         s.probeKeys = pk;
         s.buildKeys = bk;
         s.keyDecimal.assign(pk.size(), false);
      }
      if (hiv->ht.empty()) {
         flush(b);
         bool uniq = false;
         const std::set<std::string> ks(s.buildKeys.begin(), s.buildKeys.end());
         for (auto& u : b.unique) {
            bool sub = !u.empty();
            for (auto& c : u) sub = sub && ks.count(c);
            uniq = uniq || sub;
         }
         OutStep jb;
         jb.op = "join_build";
         jb.out = fresh("h");
         jb.fields = {{"in", quote(b.rel)}, {"keys", nameList(s.buildKeys)}, {"unique", uniq ? "true" : "false"}};
         steps.push_back(jb);
         hiv->ht = jb.out;
         hiv->htKeys = s.buildKeys;
         hiv->htUnique = uniq;
      } else if (hiv->htKeys != s.buildKeys) {
         throw Unsupported("one hash-indexed view probed on two different key lists");
      }
      flush(s);
      OutStep jp;
      jp.op = "join_probe";
      jp.out = fresh("j");
      jp.fields = {{"ht", quote(hiv->ht)}, {"in", quote(s.rel)}, {"keys", nameList(s.probeKeys)}, {"kind", quote(kind)}};
      if (!s.residual.empty()) {
         std::string r = "[";
         for (size_t k = 0; k < s.residual.size(); k++) r += (k ? ", " : "") + s.residual[k];
         jp.fields.push_back({"residual", r + "]"});
      }
      if (kind == "mark") jp.fields.push_back({"mark_as", quote(markName)});
      if (kind != "inner" && !s.postJoin.empty()) throw Unsupported("a " + kind + " join whose predicate has a conjunct on one side only");
      steps.push_back(jp);
      return jp.out;
   }
   // ---------------------------------------------------------------- windows
   static ExprP wref(const char* what, int64_t k = 0) {
      ExprP r = mk(Expr::REF, what);
      r->i = k;
      return r;
   }
   static int64_t frameEnd(const ExprP& r) { // a reference into the continuous view as a frame end: offset from the current row, or unbounded
      if (r->kind != Expr::REF) throw Unsupported("window frame end that is not a view reference");
      if (r->name == "w:begin") return INT64_MIN;
      if (r->name == "w:end") return INT64_MAX;
      if (r->name == "w:cur") return 0;
      if (r->name == "w:off") return r->i;
      throw Unsupported("window frame end that is not a view reference");
   }
   void setFrame(Stream& s, int64_t from, int64_t to, bool haveTo) {
      if (s.winFrame && (s.winFrom != from || (haveTo && s.winHasTo && s.winTo != to))) throw Unsupported("window functions with different frames in one window");
      if (!s.winFrame) s.winTo = 0; // (a rank alone reads only the frame's begin)
      s.winFrame = true;
      s.winFrom = from;
      if (haveTo) {
         s.winTo = to;
         s.winHasTo = true;
      }
   }
   // the functions collected on this stream become ONE window step: partition + order of the view, one frame, one result column each
   void flushWindow(Stream& s) {
      if (s.winFns.empty()) return;
      StateP view = states.at(s.winView);
      flush(s);
      auto end = [](int64_t v) { return v == INT64_MIN ? std::string("\"unbounded_preceding\"") : v == INT64_MAX ? std::string("\"unbounded_following\"") : std::to_string(v); };
      OutStep w;
      w.op = "window";
      w.out = fresh("w");
      w.fields = {{"in", quote(s.rel)}};
      if (!view->partCols.empty()) w.fields.push_back({"partition_by", nameList(view->partCols)});
      if (!view->sortCols.empty()) {
         std::string by = "[";
         for (size_t k = 0; k < view->sortCols.size(); k++) by += (k ? ", " : "") + (view->sortCols[k].second ? "{\"col\": " + quote(view->sortCols[k].first) + ", \"desc\": true}" : quote(view->sortCols[k].first));
         w.fields.push_back({"order_by", by + "]"});
      }
      w.fields.push_back({"frame_from", end(s.winFrom)});
      w.fields.push_back({"frame_to", end(s.winTo)});
      std::string fns = "[";
      for (size_t k = 0; k < s.winFns.size(); k++) {
         const auto& f = s.winFns[k];
         fns += std::string(k ? ", " : "") + "{\"fn\": " + quote(f.fn) + (f.col.empty() ? "" : ", \"col\": " + quote(f.col)) + ", \"as\": " + quote(f.as) + "}";
         s.names.insert(f.as);
      }
      w.fields.push_back({"fns", fns + "]"});
      steps.push_back(w);
      s.rel = w.out;
      s.bareTable = false;
      s.unique.clear();
      s.winFns.clear();
      s.winView.clear();
      s.winLookup.clear();
      s.winFrame = false;
      s.winHasTo = false;
   }

   // the unflagged rows of a build buffer are consumed as rows: an anti join keeping the build side
   void realiseFlag(Stream& s) {
      if (!s.flagAntiPending) return;
      s.flagAntiPending = false;
      StateP buf = states.at(s.flagState);
      if (buf->antiRel.empty()) buf->antiRel = emitJoin(*buf->flagProbe, "anti_build");
      s.rel = buf->antiRel;
      s.preds.clear();
      s.bareTable = false;
      s.flagState.clear();
   }
   // the pending join turns out to be an ordinary (inner) one: some other sub-operator consumes the matched pairs
   void settle(Stream& s) {
      realiseFlag(s);
      flushWindow(s);
      if (!s.pending) return;
      StateP hiv = states.at(s.probeHiv);
      bool marked = false;
      for (auto& kv : s.cols) marked = marked || kv.second->kind == Expr::MARKER;
      if (marked) { // MarkJoinLowering (RelAlgToSubOp.cpp:1376-1408): the marker is read as a value (mark or …), not filtered on: every probe row + a boolean column
         markName = "mark" + std::to_string(++nval);
         s.rel = emitJoin(s, "mark");
         ExprP m = mk(Expr::COL, markName);
         s.names.insert(markName);
         s.pending = false;
         s.inJoinBody = false;
         s.probeHiv.clear();
         s.probeKeys.clear();
         s.buildKeys.clear();
         s.keyDecimal.clear();
         s.residual.clear();
         for (auto it = s.cols.begin(); it != s.cols.end();) {
            if (it->second->kind == Expr::MARKER) it->second = m;
            it = it->second->build ? s.cols.erase(it) : std::next(it);
         }
         return;
      }
      s.rel = emitJoin(s, "inner");
      s.names.insert(hiv->source->in.names.begin(), hiv->source->in.names.end());
      clearJoin(s);
      if (!hiv->htUnique) s.unique.clear(); // probe rows may repeat
      const std::vector<ExprP> post = s.postJoin;
      s.postJoin.clear();
      for (auto& e : post) applyFilter(s, e);
   }
   // the join has been emitted: its build columns are columns of the stream now
   void clearJoin(Stream& s) {
      s.pending = false;
      s.inJoinBody = false;
      s.probeHiv.clear();
      s.probeKeys.clear();
      s.buildKeys.clear();
      s.keyDecimal.clear();
      s.residual.clear();
      for (auto& kv : s.cols)
         if (kv.second->build) {
            auto c = std::make_shared<Expr>(*kv.second);
            c->build = false;
            kv.second = c;
         }
   }
   // a restriction on the rows of a stream: conjunctions of column-vs-constant comparisons / IN / LIKE / NOT NULL, two
   // columns of base tables, a disjunction of such conjunctions (→ filter_dnf), or any arithmetic / boolean expression
   // (→ a computed boolean column + `= true`)
   void applyFilter(Stream& s, const ExprP& e0) {
      const ExprP e = materializeCalls(s, stripCast(e0), "pred");
      std::vector<std::string> ps;
      if (condPreds(e, ps)) {
         s.preds.insert(s.preds.end(), ps.begin(), ps.end());
         return;
      }
      if (e->kind == Expr::OP && e->name == "cmp") {
         const ExprP l = stripCast(e->args[0]), r = stripCast(e->args[1]);
         if (l->kind == Expr::COL && r->kind == Expr::COL && l->base && r->base) {
            use(l->name), use(r->name);
            s.preds.push_back("{\"col\": " + quote(l->name) + ", \"op\": " + quote(e->cmp) + ", \"rhs_col\": " + quote(r->name) + "}");
            return;
         }
      }
      if (e->kind == Expr::OP && e->name == "or") { // disjunctive normal form
         std::string clauses = "[";
         bool ok = true;
         for (size_t k = 0; k < e->args.size() && ok; k++) {
            std::vector<std::string> cl;
            ok = condPreds(e->args[k], cl);
            clauses += std::string(k ? ", " : "") + "[";
            for (size_t j = 0; j < cl.size(); j++) clauses += (j ? ", " : "") + cl[j];
            clauses += "]";
         }
         if (ok) {
            flush(s);
            OutStep f;
            f.op = "filter_dnf";
            f.out = fresh("v");
            f.fields = {{"in", quote(s.rel)}, {"clauses", clauses + "]"}};
            steps.push_back(f);
            s.rel = f.out;
            return;
         }
      }
      const std::string col = ensureCol(s, e, "pred" + std::to_string(++nval));
      s.preds.push_back("{\"col\": " + quote(col) + ", \"op\": \"EQ\", \"value\": 1}");
   }
   // semi / anti join keeping the PROBE rows: the build columns are gone afterwards
   void finishProbeSide(Stream& s, bool anti) {
      s.rel = emitJoin(s, anti ? "anti" : "semi");
      s.pending = false;
      s.inJoinBody = false;
      s.probeHiv.clear();
      s.probeKeys.clear();
      s.buildKeys.clear();
      s.keyDecimal.clear();
      s.residual.clear();
      for (auto it = s.cols.begin(); it != s.cols.end();)
         it = it->second->build || it->second->kind == Expr::MARKER ? s.cols.erase(it) : std::next(it);
   }

   // nested-loop join: `pred` (may be null: cross product) is a conjunction of comparisons between a column of the outer
   // stream and a column of the scanned buffer
   void emitNestedLoop(Stream& s, const ExprP& pred) {
      StateP buf = states.at(s.nlBuild);
      Stream& b = buf->in;
      std::string resid = "[";
      int n = 0;
      std::function<void(const ExprP&)> walk = [&](const ExprP& e0) {
         const ExprP e = stripCast(e0);
         if (e->kind == Expr::OP && e->name == "and") {
            for (auto& a : e->args) walk(a);
            return;
         }
         if (!(e->kind == Expr::OP && e->name == "cmp")) throw Unsupported("nested-loop join predicate that is not a conjunction of comparisons");
         ExprP l = stripCast(e->args[0]), r = stripCast(e->args[1]);
         std::string op = e->cmp;
         if (l->build && !r->build) {
            std::swap(l, r);
            op = op == "LT" ? "GT" : op == "GT" ? "LT" : op == "LTE" ? "GTE" : op == "GTE" ? "LTE" : op;
         }
         if (l->build || !r->build) throw Unsupported("nested-loop join comparison that does not relate the two sides");
         const std::string bc = ensureCol(b, r, "nl_build"), pc = ensureCol(s, l, "nl_probe");
         resid += std::string(n++ ? ", " : "") + "{\"probe\": " + quote(pc) + ", \"op\": " + quote(op) + ", \"build\": " + quote(bc) + "}";
      };
      if (pred) walk(pred);
      if (n > 2) throw Unsupported("nested-loop join with more than two comparison conjuncts");
      flush(b);
      flush(s);
      OutStep j;
      j.op = "join_nl";
      j.out = fresh("j");
      j.fields = {{"in", quote(s.rel)}, {"build", quote(b.rel)}, {"residual", resid + "]"}, {"kind", "\"inner\""}};
      steps.push_back(j);
      s.rel = j.out;
      s.names.insert(b.names.begin(), b.names.end());
      s.nlBuild.clear();
      s.unique.clear();
      for (auto& kv : s.cols)
         if (kv.second->build) {
            auto c = std::make_shared<Expr>(*kv.second);
            c->build = false;
            kv.second = c;
         }
   }

   std::string idOf(const StateP& st, StepCtx& c) {
      for (auto& kv : states)
         if (kv.second == st) return kv.first;
      for (auto& kv : c.local)
         if (kv.second == st) {
            states[kv.first] = st;
            return kv.first;
         }
      throw Unsupported("state without an identity");
   }

   void handle(const J& op, StepCtx& c) {
      const J* kindp = op.get("subop");
      if (!kindp) throw Unsupported("a sub-operator the dump tool has no case for (\"operator\": \"unknown\")");
      const std::string& kind = kindp->str;
      const std::string& ref = op.s("ref");
      auto newState = [&](State::Kind k) {
         auto s = std::make_shared<State>();
         s->kind = k;
         c.local[ref + "#0"] = s;
         return s;
      };
      if (kind == "get_external") {
         const J& meta = op.at("meta");
         StateP s = newState(State::TABLE);
         s->table = meta.s("tableName");
         inputs.insert(s->table);
         for (auto& m : meta.at("mapping").arr) s->memberToIdent[m.s("memberName")] = m.s("identifier");
         if (const J* fs = meta.get("filters"))
            for (auto& f : fs->arr) {
               const std::string& fop = f.s("op");
               if (fop == "UNKNOWN") throw Unsupported("restriction with an unknown FilterOp");
               s->filters.push_back(predJson(f.s("columnName"), fop, f.get("value"), f.get("values")));
            }
         if (const J* pk = meta.get("primaryKey")) // emitter extension E5
            for (auto& k : pk->arr) s->pkey.push_back(k.str);
         if (meta.get("index")) throw Unsupported("get_external over an external hash index");
         return;
      }
      if (kind == "create_thread_local" || kind == "create_simple_state" || kind == "generic_create" || kind == "create_array" || kind == "create_from") {
         newState(State::UNKNOWN);
         return;
      }
      if (kind == "create_heap") {
         StateP s = newState(State::HEAP);
         s->maxRows = op.iOr("maxRows", -1); // emitter extension E4
         if (const J* sb = op.get("sortBy"))
            for (auto& k : sb->arr) s->sortBy.push_back({k.s("member"), k.sOr("direction", "asc") == "desc"});
         if (s->maxRows < 0 || s->sortBy.empty()) throw Unsupported("create_heap without maxRows / sortBy (emitter extension E4)");
         return;
      }
      if (kind == "merge") { // the thread-local instances merged into one state: there is one state on a GPU (a thread-local step input is the state itself)
         c.local[ref + "#0"] = resolve(op.at("accesses").arr.at(0), c);
         return;
      }
      if (kind == "create_hash_indexed_view") {
         StateP src = resolve(op.at("accesses").arr.at(0), c);
         if (src->kind != State::BUFFER) throw Unsupported("hash-indexed view over a state that is not a materialised buffer");
         StateP s = newState(State::HIV);
         s->source = src;
         return;
      }
      if (kind == "create_sorted_view") {
         StateP src = resolve(op.at("accesses").arr.at(0), c);
         if (src->kind != State::BUFFER) throw Unsupported("sorted view over a state that is not a materialised buffer");
         const J* sb = op.get("sortBy"); // emitter extension E4
         if (!sb || sb->arr.empty()) throw Unsupported("create_sorted_view without sortBy (emitter extension E4)");
         StateP s = newState(State::SORTED);
         s->source = src;
         s->in = src->in;
         s->members = src->members;
         if (!src->partPending) flush(s->in);
         std::string by = "[";
         for (size_t k = 0; k < sb->arr.size(); k++) {
            auto it = s->members.find(sb->arr[k].s("member"));
            if (it == s->members.end()) throw Unsupported("sort key '" + sb->arr[k].s("member") + "' is not a member of the buffer");
            const std::string col = ensureCol(s->in, it->second, stripSuffix(it->first));
            it->second = mk(Expr::COL, col);
            s->sortCols.push_back({col, sb->arr[k].sOr("direction", "asc") == "desc"});
            by += (k ? ", " : "") + (sb->arr[k].sOr("direction", "asc") == "desc" ? "{\"col\": " + quote(col) + ", \"desc\": true}" : quote(col));
         }
         s->partCols = src->partCols;
         if (src->partPending) { // one partition's buffer of a window: the window step orders the rows inside their partitions itself
            if (s->in.rel != src->in.rel) throw Unsupported("window ordered by a computed column inside a partition");
            return;
         }
         OutStep so;
         so.op = "sort";
         so.out = fresh("s");
         so.fields = {{"in", quote(s->in.rel)}, {"by", by + "]"}};
         steps.push_back(so);
         s->in.rel = so.out;
         return;
      }
      if (kind == "create_continuous_view") { // the rows of a (sorted) buffer addressable by position: the input of a window evaluation
         StateP src = resolve(op.at("accesses").arr.at(0), c);
         if (src->kind != State::SORTED && src->kind != State::BUFFER) throw Unsupported("continuous view over a state that is neither a buffer nor a sorted view");
         StateP v = newState(State::CONTVIEW);
         v->source = src;
         v->in = src->in;
         v->members = src->members;
         v->sortCols = src->sortCols;
         v->partCols = src->partCols;
         return;
      }
      if (kind == "create_segment_tree_view") {
         StateP src = resolve(op.at("accesses").arr.at(0), c);
         if (src->kind != State::CONTVIEW) throw Unsupported("segment tree over a state that is not a continuous view");
         const J* ag = op.get("aggregates"); // emitter extension E8: the tool prints neither the functions nor their source members
         if (!ag || ag->arr.empty()) throw Unsupported("create_segment_tree_view without its aggregates (emitter extension E8)");
         StateP t = newState(State::SEGTREE);
         t->source = src;
         for (auto& a : ag->arr) t->winAggs.push_back({a.s("member"), a.s("fn"), a.sOr("source", "")});
         return;
      }
      if (kind == "scan_ref") { // every entry of the continuous view, by reference
         StateP st = resolve(op.at("accesses").arr.at(0), c);
         if (st->kind != State::CONTVIEW) throw Unsupported("scan_ref over a state that is not a continuous view");
         Stream s = c.nested && !st->partCols.empty() ? *c.nested : st->in;
         s.winView = idOf(st, c);
         s.cols[op.at("reference").s("displayName")] = wref("w:cur");
         c.streams[ref] = s;
         return;
      }
      if (kind == "get_begin_reference" || kind == "get_end_reference") {
         Stream s = input(op, c);
         if (s.winView.empty()) throw Unsupported(kind + " outside a window evaluation");
         s.cols[op.at("reference").s("displayName")] = wref(kind == "get_begin_reference" ? "w:begin" : "w:end");
         c.streams[ref] = s;
         return;
      }
      if (kind == "offset_reference_by") { // ROWS k PRECEDING / FOLLOWING: the current reference moved by a constant, clamped into the partition
         Stream s = input(op, c);
         auto r = s.cols.find(op.at("reference").s("displayName"));
         auto o = s.cols.find(op.at("offset").s("displayName"));
         if (s.winView.empty() || r == s.cols.end() || r->second->kind != Expr::REF || r->second->name != "w:cur" || o == s.cols.end() || stripCast(o->second)->kind != Expr::CONST_INT)
            throw Unsupported("offset_reference_by that does not move the current row of a window by a constant");
         s.cols[op.at("newRef").s("displayName")] = wref("w:off", stripCast(o->second)->i);
         c.streams[ref] = s;
         return;
      }
      if (kind == "entries_between") { // RankWindowFunc (:2043-2058): entries between the frame begin and the current row
         Stream s = input(op, c);
         auto l = s.cols.find(op.at("leftRef").s("displayName"));
         auto r = s.cols.find(op.at("rightRef").s("displayName"));
         if (s.winView.empty() || l == s.cols.end() || r == s.cols.end() || r->second->kind != Expr::REF || r->second->name != "w:cur") throw Unsupported("entries_between that does not end at the current row of a window");
         ExprP b = mk(Expr::OP, "wbetween");
         b->i = frameEnd(l->second);
         s.cols[op.at("between").s("displayName")] = b;
         c.streams[ref] = s;
         return;
      }
      if (kind == "scan") {
         StateP st = resolve(op.at("accesses").arr.at(0), c);
         const J& mapping = op.at("mapping");
         Stream s;
         if (st->kind == State::TABLE) {
            s.rel = st->table;
            s.bareTable = true;
            s.preds = st->filters;
            if (!st->pkey.empty()) s.unique.push_back(std::set<std::string>(st->pkey.begin(), st->pkey.end()));
            for (auto& kv : st->memberToIdent) s.names.insert(kv.second);
            for (auto& m : mapping.arr) {
               auto it = st->memberToIdent.find(m.s("member"));
               if (it == st->memberToIdent.end()) throw Unsupported("scan of member '" + m.s("member") + "' that get_external does not map");
               ExprP c = mk(Expr::COL, it->second);
               c->base = true;
               c->dtype = m.at("column").sOr("datatype", "");
               s.cols[m.at("column").s("displayName")] = c;
            }
         } else if (st->kind == State::AGG && st->gjRight) {
            // GroupJoinLowering: the left input created one entry per key (its stored columns are ANY aggregates), the right input looked its
            // group up and aggregated into it.  Emitted as: distinct left keys (+ stored columns) → unique hash table → the right input
            // probes it (inner) → the join predicate → group by the key with the aggregates.  The inner behaviour (marker member) keeps
            // exactly the groups this produces; the outer behaviour would need the unmatched left rows with default aggregates.
            const bool outerGj = !st->gjInner; // outer behaviour: every left key comes out, groups without a partner with their default aggregates (count 0, the others NULL)
            // the left side: one row per key; the stored columns are functionally dependent on it and travel as further keys
            // (ANY over a string has no device form, a key has)
            std::map<std::string, std::string> storedName; // gjval member → display name
            for (auto& m : mapping.arr)
               if (m.s("member").rfind("gjval$", 0) == 0) storedName[m.s("member")] = m.at("column").s("displayName");
            {
               std::string lm = "[";
               for (auto& m : mapping.arr)
                  if (isKeyMember(m.s("member"))) lm += std::string(lm.size() > 1 ? ", " : "") + "{\"member\": " + quote(m.s("member")) + ", \"column\": {\"displayName\": " + quote(m.at("column").s("displayName")) + "}}";
               std::vector<AggSpec> keep;
               for (auto& a : st->aggs) {
                  if (a.fn == "any" && storedName.count(a.member) && a.arg) {
                     st->in.cols[storedName[a.member]] = a.arg;
                     lm += ", {\"member\": " + quote("keyval$" + a.member) + ", \"column\": {\"displayName\": " + quote(storedName[a.member]) + "}}";
                  } else {
                     keep.push_back(a);
                  }
               }
               st->aggs = keep;
               if (st->aggs.empty()) {
                  AggSpec a;
                  a.member = "distinct$count";
                  a.fn = "count_star";
                  st->aggs.push_back(a);
               }
               lm += "]";
               JParser lp(lm.c_str());
               const J lmj = lp.value();
               emitGroupBy(st, lmj);
            }
            auto tmp = std::make_shared<State>();
            tmp->kind = State::BUFFER;
            tmp->in = st->in;
            std::vector<std::string> keyMembers, keyNames;
            for (auto& m : mapping.arr)
               if (isKeyMember(m.s("member"))) {
                  keyMembers.push_back(m.s("member"));
                  keyNames.push_back(m.at("column").s("displayName"));
                  tmp->members[m.s("member")] = mk(Expr::COL, st->aggOut.at(m.s("member")));
               }
            for (auto& sp : st->gjStored) {
               auto it = st->aggOut.find("keyval$" + sp.first);
               if (it == st->aggOut.end()) throw Unsupported("group join gathers member '" + sp.first + "' the left input did not store");
               tmp->members[sp.first] = mk(Expr::COL, it->second);
            }
            Stream r = *st->gjRight;
            separateNames(r, tmp, false);
            for (auto& sp : st->gjStored) { // the placeholders handed out by the gather become the (possibly renamed) stored columns
               sp.second->name = tmp->members.at(sp.first)->name;
               sp.second->build = false;
            }
            std::vector<std::string> bk, pk;
            for (size_t k = 0; k < keyMembers.size(); k++) {
               ExprP right;
               for (auto& pr : st->gjPairs)
                  if (pr.first == keyNames[k]) right = pr.second;
               if (!right) throw Unsupported("group join: no renaming names the right key of '" + keyNames[k] + "'");
               bk.push_back(tmp->members.at(keyMembers[k])->name);
               pk.push_back(ensureCol(r, right, "gj_key"));
            }
            flush(tmp->in);
            OutStep jb;
            jb.op = "join_build";
            jb.out = fresh("h");
            jb.fields = {{"in", quote(tmp->in.rel)}, {"keys", nameList(bk)}, {"unique", "true"}};
            steps.push_back(jb);
            flush(r);
            OutStep jp;
            jp.op = "join_probe";
            jp.out = fresh("j");
            jp.fields = {{"ht", quote(jb.out)}, {"in", quote(r.rel)}, {"keys", nameList(pk)}, {"kind", "\"inner\""}};
            steps.push_back(jp);
            r.rel = jp.out;
            r.names.insert(tmp->in.names.begin(), tmp->in.names.end());
            for (auto& e : st->gjFilters) applyFilter(r, e);
            // the final aggregation: keyed by the (right) key columns, carrying the stored columns as ANY
            auto fin = std::make_shared<State>();
            fin->kind = State::AGG;
            fin->in = r;
            fin->aggs = st->gjAggs;
            std::string mj = "[";
            for (size_t k = 0; k < keyMembers.size(); k++) {
               fin->in.cols[keyNames[k]] = mk(Expr::COL, pk[k]);
               mj += std::string(k ? ", " : "") + "{\"member\": " + quote(keyMembers[k]) + ", \"column\": {\"displayName\": " + quote(keyNames[k]) + "}}";
            }
            for (auto& sp : st->gjStored) { // carried as keys (see above)
               auto sn = storedName.find(sp.first);
               if (sn == storedName.end() || outerGj) continue; // gathered for the predicate only / taken from the left side below
               fin->in.cols[sn->second] = sp.second;
               mj += ", {\"member\": " + quote("keyval$" + sp.first) + ", \"column\": {\"displayName\": " + quote(sn->second) + "}}";
            }
            for (auto& m : mapping.arr)
               if (!isKeyMember(m.s("member")) && !storedName.count(m.s("member"))) mj += ", {\"member\": " + quote(m.s("member")) + ", \"column\": {\"displayName\": " + quote(m.at("column").s("displayName")) + "}}";
            mj += "]";
            JParser jp2(mj.c_str());
            const J fm = jp2.value();
            emitGroupBy(fin, fm);
            if (outerGj) { // the left keys (+ stored columns) left-outer-joined with their groups; a count without a group is 0
               std::vector<std::string> gk;
               for (auto& km : keyMembers) gk.push_back(fin->aggOut.at(km));
               Stream left = tmp->in;
               flush(fin->in);
               OutStep jb2;
               jb2.op = "join_build";
               jb2.out = fresh("h");
               jb2.fields = {{"in", quote(fin->in.rel)}, {"keys", nameList(gk)}, {"unique", "true"}};
               steps.push_back(jb2);
               OutStep jp3;
               jp3.op = "join_probe";
               jp3.out = fresh("j");
               jp3.fields = {{"ht", quote(jb2.out)}, {"in", quote(left.rel)}, {"keys", nameList(bk)}, {"kind", "\"left_outer\""}};
               steps.push_back(jp3);
               s = left;
               s.rel = jp3.out;
               s.names.insert(fin->in.names.begin(), fin->in.names.end());
               for (auto& m : mapping.arr) {
                  const std::string& name = m.at("column").s("displayName");
                  if (isKeyMember(m.s("member"))) {
                     s.cols[name] = tmp->members.at(m.s("member"));
                  } else if (storedName.count(m.s("member"))) {
                     auto it = st->aggOut.find("keyval$" + m.s("member"));
                     if (it == st->aggOut.end()) throw Unsupported("group join scans a stored member the left input did not store");
                     s.cols[name] = mk(Expr::COL, it->second);
                  } else {
                     auto it = fin->aggOut.find(m.s("member"));
                     if (it == fin->aggOut.end()) throw Unsupported("group join scans member '" + m.s("member") + "' nothing aggregates");
                     ExprP v = mk(Expr::COL, it->second);
                     bool counts = false;
                     for (auto& a : st->gjAggs) counts = counts || (a.member == m.s("member") && (a.fn == "count" || a.fn == "count_star"));
                     if (counts) { // CountAggrFunc / CountStarAggrFunc start at 0 (createDefaultValue, :1812, :1826)
                        ExprP zero = mk(Expr::CONST_INT), co = mk(Expr::OP, "coalesce");
                        co->args = {v, zero};
                        v = co;
                     }
                     s.cols[name] = v;
                  }
               }
               c.streams[ref] = s;
               return;
            }
            s = fin->in;
            for (auto& m : mapping.arr) {
               auto it = fin->aggOut.find(storedName.count(m.s("member")) ? "keyval$" + m.s("member") : m.s("member"));
               if (it == fin->aggOut.end()) { // the marker member: every group here has a partner
                  ExprP t = mk(Expr::CONST_BOOL);
                  t->i = 1;
                  s.cols[m.at("column").s("displayName")] = t;
                  continue;
               }
               ExprP c2 = mk(Expr::COL, it->second);
               c2->dtype = m.at("column").sOr("datatype", "");
               s.cols[m.at("column").s("displayName")] = c2;
            }
         } else if (st->kind == State::AGG && st->setRight) { // the map both inputs of a set operation counted into
            Stream &a = st->in, &b = *st->setRight;
            st->setLeftCols.clear();
            st->setRightCols.clear();
            std::vector<std::string> names;
            for (auto& m : mapping.arr) {
               if (m.s("member").rfind("keyval", 0) != 0) continue;
               const std::string& name = m.at("column").s("displayName");
               auto ia = a.cols.find(name), ib = b.cols.find(name);
               if (ia == a.cols.end() || ib == b.cols.end()) throw Unsupported("set operation: column '" + name + "' is not defined on both inputs");
               st->setLeftCols.push_back(ensureCol(a, ia->second, name));
               st->setRightCols.push_back(ensureCol(b, ib->second, name));
               names.push_back(name);
            }
            if (names.empty()) throw Unsupported("set operation without columns");
            for (size_t k = 0; k < names.size(); k++) {
               s.cols[names[k]] = mk(Expr::COL, st->setLeftCols[k]);
               s.names.insert(st->setLeftCols[k]);
            }
            s.setState = idOf(st, c);
            if (st->setLeftCounter.empty()) {
               emitSetOp(s, "union"); // UnionDistinctLowering: the keys of the map are the result
            } else {
               for (auto& m : mapping.arr) {
                  const int which = m.s("member") == st->setLeftCounter ? 1 : m.s("member") == st->setRightCounter ? 2 : 0;
                  if (!which) continue;
                  ExprP cnt = mk(Expr::SETCNT);
                  cnt->i = which;
                  s.cols[m.at("column").s("displayName")] = cnt;
               }
            }
         } else if (st->kind == State::AGG) {
            emitGroupBy(st, mapping);
            s = st->in;
            for (auto& m : mapping.arr) {
               auto it = st->aggOut.find(m.s("member"));
               if (it == st->aggOut.end()) throw Unsupported("scan of member '" + m.s("member") + "' the aggregation does not produce");
               ExprP c = mk(Expr::COL, it->second);
               c->dtype = m.at("column").sOr("datatype", "");
               s.cols[m.at("column").s("displayName")] = c;
            }
         } else if (st->kind == State::MARKER) { // the per-row marker of anyTuple: the pending join continues on the probe stream
            s = st->in;
            for (auto& m : mapping.arr) s.cols[m.at("column").s("displayName")] = mk(Expr::MARKER);
         } else if (st->kind == State::CONTVIEW) { // every row of the (partition's) view once: the static aggregation of an UNBOUNDED … UNBOUNDED frame (:2497-2499)
            s = c.nested && !st->partCols.empty() ? *c.nested : st->in;
            for (auto& m : mapping.arr) {
               auto it = st->members.find(m.s("member"));
               if (it == st->members.end()) throw Unsupported("scan of member '" + m.s("member") + "' that was never materialised");
               s.cols[m.at("column").s("displayName")] = it->second;
            }
            s.winStatic = idOf(st, c);
         } else if (st->kind == State::BUFFER && st->partPending) { // the map keyed by the PARTITION BY columns whose value is the partition's buffer
            s = st->in;
            st->partCols.clear();
            for (auto& m : mapping.arr) {
               const std::string& name = m.at("column").s("displayName");
               if (isKeyMember(m.s("member"))) {
                  auto it = st->in.cols.find(name);
                  if (it == st->in.cols.end()) throw Unsupported("partition key '" + name + "' is not a column of the windowed stream");
                  st->partCols.push_back(ensureCol(st->in, it->second, name));
               } else {
                  s.cols[name] = wref("w:partbuf");
               }
            }
            s = st->in; // (ensureCol may have added columns)
            for (auto& m : mapping.arr)
               if (!isKeyMember(m.s("member"))) s.cols[m.at("column").s("displayName")] = wref("w:partbuf");
            s.partState = idOf(st, c);
         } else if (st->kind == State::BUFFER || st->kind == State::SORTED || st->kind == State::HEAP || st->kind == State::RESULT) {
            if (st->kind == State::HEAP && !st->emitted) { // the heap keeps the best maxRows rows: a top-k over what was materialised
               flush(st->in);
               std::string by = "[";
               for (size_t k = 0; k < st->sortBy.size(); k++) {
                  auto it = st->members.find(st->sortBy[k].first);
                  if (it == st->members.end()) throw Unsupported("heap sort key '" + st->sortBy[k].first + "' was not materialised");
                  const std::string col = ensureCol(st->in, it->second, stripSuffix(it->first));
                  it->second = mk(Expr::COL, col);
                  by += (k ? ", " : "") + (st->sortBy[k].second ? "{\"col\": " + quote(col) + ", \"desc\": true}" : quote(col));
               }
               OutStep tk;
               tk.op = "topk";
               tk.out = fresh("t");
               tk.fields = {{"in", quote(st->in.rel)}, {"by", by + "]"}, {"k", std::to_string(st->maxRows)}};
               steps.push_back(tk);
               st->in.rel = tk.out;
               st->emitted = true;
            }
            if (c.nested && st->kind == State::BUFFER) { // translateNLJ: the buffer is scanned once per tuple of the outer stream
               s = *c.nested;
               settle(s);
               if (!s.nlBuild.empty()) throw Unsupported("two nested scans in one nested_map body");
               s.nlBuild = idOf(st, c);
               separateNames(s, st, false);
               for (auto& m : mapping.arr) {
                  auto it = st->members.find(m.s("member"));
                  if (it == st->members.end()) throw Unsupported("scan of member '" + m.s("member") + "' that was never materialised");
                  auto e = std::make_shared<Expr>(*stripCast(it->second));
                  if (e->kind != Expr::COL) {
                     const std::string colname = ensureCol(st->in, it->second, stripSuffix(it->first));
                     it->second = mk(Expr::COL, colname);
                     e = std::make_shared<Expr>(*it->second);
                  }
                  e->build = true;
                  s.cols[m.at("column").s("displayName")] = e;
               }
               c.streams[ref] = s;
               return;
            }
            s.rel = st->in.rel;
            s.preds = st->in.preds;
            s.unique = st->in.unique;
            s.names = st->in.names;
            s.bareTable = st->in.bareTable;
            for (auto& m : mapping.arr) {
               auto it = st->members.find(m.s("member"));
               if (it == st->members.end()) throw Unsupported("scan of member '" + m.s("member") + "' that was never materialised");
               s.cols[m.at("column").s("displayName")] = it->second;
               if (!st->flagMember.empty() && m.s("member") == st->flagMember) { // the flag a reversed semi / anti join has set
                  s.cols[m.at("column").s("displayName")] = mk(Expr::FLAG);
                  s.flagState = idOf(st, c);
               }
            }
         } else {
            throw Unsupported("scan of a state nothing was written to");
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "nested_map") { // the body runs once per tuple of the input stream: inline it
         Stream s = input(op, c);
         if (!s.setState.empty()) { // INTERSECT ALL / EXCEPT ALL: the body generates `repeat` copies of the key (generate + scf.for, :893-911)
            std::string kind;
            for (auto& kv : s.cols)
               if (kv.second->kind == Expr::OP && kv.second->name == "setop") kind = kv.second->cmp;
            if (kind != "intersect_all" && kind != "except_all") throw Unsupported("nested_map over the counters of a set operation without a repeat count");
            emitSetOp(s, kind);
            c.streams[ref] = s;
            return;
         }
         Stream* outer = c.nested;
         const bool outerWas = c.outerBody;
         c.nested = &s;
         c.outerBody = false;
         std::string last;
         if (const J* body = op.get("subops")) {
            for (auto& b : body->arr) c.outerBody = c.outerBody || (b.get("subop") && b.s("subop") == "union");
            for (auto& b : body->arr) {
               handle(b, c);
               if (b.get("subop") && c.streams.count(b.s("ref"))) last = b.s("ref");
            }
         }
         c.nested = outer;
         c.outerBody = outerWas;
         c.streams[ref] = last.empty() ? s : c.streams[last];
         c.streams[ref].inJoinBody = false;
         flushWindow(c.streams[ref]); // a window evaluated per partition inside the body
         if (!c.streams[ref].nlBuild.empty()) emitNestedLoop(c.streams[ref], nullptr); // no predicate in the body: a cross product
         return;
      }
      if (kind == "scan_list") {
         Stream s = input(op, c);
         if (s.probeHiv.empty()) throw Unsupported("scan_list outside a hash-indexed-view lookup");
         s.cols[op.at("elem").s("displayName")] = mk(Expr::REF);
         c.streams[ref] = s;
         return;
      }
      if (kind == "unwrap_optional_ref") { // group join: tuples without a group are dropped — the inner join emitted for the group join does that
         Stream s = input(op, c);
         if (s.gjState.empty()) throw Unsupported("unwrap_optional_ref outside a group join");
         c.streams[ref] = s;
         return;
      }
      if (kind == "gather" && !input(op, c).winView.empty()) {
         Stream s = input(op, c);
         auto r = s.cols.find(op.at("reference").s("displayName"));
         if (r == s.cols.end() || r->second->kind != Expr::REF) throw Unsupported("gather through an undefined reference");
         StateP view = states.at(s.winView);
         if (r->second->name == "w:cur") { // the columns of the current row
            for (auto& m : op.at("mapping").arr) {
               auto it = view->members.find(m.s("member"));
               if (it == view->members.end()) throw Unsupported("gather of member '" + m.s("member") + "' that was never materialised");
               s.cols[m.at("column").s("displayName")] = it->second;
            }
         } else if (r->second->name == "w:lookup") { // the aggregates of the frame, from the segment tree
            StateP seg = states.at(s.winLookup);
            for (auto& m : op.at("mapping").arr) {
               const State::WinAgg* a = nullptr;
               for (auto& x : seg->winAggs)
                  if (x.member == m.s("member")) a = &x;
               if (!a) throw Unsupported("gather of member '" + m.s("member") + "' the segment tree does not aggregate");
               static const std::set<std::string> known = {"sum", "min", "max", "count", "count_star"};
               if (!known.count(a->fn)) throw Unsupported("window aggregate '" + a->fn + "'");
               Stream::WinFn f;
               f.fn = a->fn;
               if (a->fn != "count_star") {
                  auto it = view->members.find(a->source);
                  if (it == view->members.end()) throw Unsupported("window aggregate over member '" + a->source + "' that was never materialised");
                  f.col = ensureCol(s, it->second, stripSuffix(a->source));
               }
               f.as = sanitize(m.at("column").s("displayName"));
               s.winFns.push_back(f);
               s.cols[m.at("column").s("displayName")] = mk(Expr::COL, f.as);
            }
         } else if (r->second->name == "w:static") {
            StateP agg = states.at(s.winLookup);
            setFrame(s, INT64_MIN, INT64_MAX, true);
            for (auto& m : op.at("mapping").arr) {
               const AggSpec* a = nullptr;
               for (auto& x : agg->aggs)
                  if (x.member == m.s("member")) a = &x;
               static const std::set<std::string> known = {"sum", "min", "max", "count", "count_star"};
               if (!a || !known.count(a->fn)) throw Unsupported("window aggregate of member '" + m.s("member") + "'");
               Stream::WinFn f;
               f.fn = a->fn;
               if (a->arg) f.col = ensureCol(s, a->arg, stripSuffix(a->member));
               f.as = sanitize(m.at("column").s("displayName"));
               s.winFns.push_back(f);
               s.cols[m.at("column").s("displayName")] = mk(Expr::COL, f.as);
            }
         } else {
            throw Unsupported("gather through a frame reference");
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "gather" && !input(op, c).gjState.empty()) { // the left input's stored columns: resolved when the group join is emitted
         Stream s = input(op, c);
         StateP st = states.at(s.gjState);
         for (auto& m : op.at("mapping").arr) {
            ExprP ph = mk(Expr::COL, "?" + m.s("member"));
            st->gjStored.push_back({m.s("member"), ph});
            s.cols[m.at("column").s("displayName")] = ph;
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "gather") {
         Stream s = input(op, c);
         if (!s.constState.empty()) { // constant single join (SingleJoinLowering, constantJoin): every tuple meets the one scattered row
            StateP st = states.at(s.constState);
            Stream& b = st->in;
            separateNames(s, st, false);
            for (auto& m : op.at("mapping").arr) {
               auto it = st->members.find(m.s("member"));
               if (it == st->members.end()) throw Unsupported("gather of member '" + m.s("member") + "' that was never scattered");
               const std::string col = ensureCol(b, it->second, stripSuffix(it->first));
               it->second = mk(Expr::COL, col);
               s.cols[m.at("column").s("displayName")] = it->second;
            }
            flush(b);
            flush(s);
            OutStep j;
            j.op = "join_nl";
            j.out = fresh("j");
            j.fields = {{"in", quote(s.rel)}, {"build", quote(b.rel)}, {"residual", "[]"}, {"kind", "\"inner\""}};
            steps.push_back(j);
            s.rel = j.out;
            s.names.insert(b.names.begin(), b.names.end());
            s.constState.clear();
            c.streams[ref] = s;
            return;
         }
         if (s.probeHiv.empty()) throw Unsupported("gather outside a hash join");
         s.inJoinBody = true;
         StateP hiv = states.at(s.probeHiv);
         separateNames(s, hiv->source, !hiv->ht.empty());
         for (auto& m : op.at("mapping").arr) {
            auto it = hiv->source->members.find(m.s("member"));
            if (it == hiv->source->members.end()) throw Unsupported("gather of member '" + m.s("member") + "' that the build side did not materialise");
            auto e = std::make_shared<Expr>(*stripCast(it->second));
            if (e->kind != Expr::COL) { // a computed build column: give it a name on the build relation first
               const std::string col = ensureCol(hiv->source->in, it->second, stripSuffix(it->first));
               it->second = mk(Expr::COL, col);
               e = std::make_shared<Expr>(*it->second);
            }
            e->build = true;
            s.cols[m.at("column").s("displayName")] = e;
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "combine_tuple") { // emitter extension E6: a no-op for column bindings
         c.streams[ref] = input(op, c);
         return;
      }
      if (kind == "renaming") {
         Stream s = input(op, c);
         for (auto& r : op.at("renamed").arr) {
            auto it = s.cols.find(r.at("old").s("displayName"));
            if (it == s.cols.end()) throw Unsupported("renaming of an undefined column");
            s.cols[r.at("new").s("displayName")] = it->second;
            if (!s.gjState.empty()) states.at(s.gjState)->gjPairs.push_back({r.at("new").s("displayName"), it->second}); // left key := the right key it equals
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "map") {
         Stream s = input(op, c);
         if (s.pending && !s.inJoinBody) { // `map … = true` is the first sub-operator of the marker idioms; anything else consumes the pairs
            bool marker = true;
            for (auto& cm : op.at("computed").arr) marker = marker && convert(cm.at("expression"), s)->kind == Expr::CONST_BOOL;
            if (!marker) settle(s);
         }
         if (s.flagAntiPending) {
            bool nulls = true, copies = true;
            for (auto& cm : op.at("computed").arr) {
               const Expr::Kind k = stripCast(convert(cm.at("expression"), s))->kind;
               nulls = nulls && k == Expr::NULLV;
               copies = copies && k == Expr::COL;
            }
            if (nulls) { // the unmatched BUILD rows of an outer join with reverseSides: null-extended and united with the matches below
               s.flagAntiPending = false;
               s.antiBranch = true;
            } else if (!copies) { // (as-nullable copies of the build columns precede the NULLs of a full outer join: still undecided)
               realiseFlag(s);
            }
         }
         if (!s.winFns.empty() || !s.winView.empty()) { // a map that is not part of the window evaluation consumes its results
            bool part = true;
            for (auto& cm : op.at("computed").arr) {
               const ExprP e = stripCast(convert(cm.at("expression"), s));
               part = part && (e->kind == Expr::CONST_INT || (e->kind == Expr::OP && e->name == "add" && stripCast(e->args[0])->kind == Expr::OP && stripCast(e->args[0])->name == "wbetween"));
            }
            if (!part) flushWindow(s);
         }
         s.lastMapped.clear();
         for (auto& cm : op.at("computed").arr) {
            const std::string& name = cm.at("computed").s("displayName");
            ExprP e = convert(cm.at("expression"), s);
            s.lastMapped.push_back(name);
            if (!s.winView.empty() && stripCast(e)->kind == Expr::OP && stripCast(e)->name == "add" && stripCast(stripCast(e)->args[0])->kind == Expr::OP &&
                stripCast(stripCast(e)->args[0])->name == "wbetween" && stripCast(stripCast(e)->args[1])->kind == Expr::CONST_INT && stripCast(stripCast(e)->args[1])->i == 1) {
               // rank = entries between the frame begin and the current row + 1 (arith.addi: emitter extension E7)
               setFrame(s, stripCast(stripCast(e)->args[0])->i, 0, false);
               Stream::WinFn f;
               f.fn = "rank";
               f.as = sanitize(name);
               s.winFns.push_back(f);
               s.cols[name] = mk(Expr::COL, f.as);
               continue;
            }
            if (!s.setState.empty()) { // the predicate / repeat count over the counters of INTERSECT / EXCEPT
               const std::string kind = classifySet(e);
               if (kind.empty()) throw Unsupported("expression over the counters of a set operation that is neither INTERSECT nor EXCEPT");
               ExprP r = mk(Expr::OP, "setop");
               r->cmp = kind;
               s.cols[name] = r;
               continue;
            }
            // sum(x) / count(x) over one aggregation = avg(x) (the frontend's expansion of avg)
            ExprP d = stripCast(e);
            if (d->kind == Expr::OP && d->name == "div") {
               ExprP n = stripCast(d->args[0]), m = stripCast(d->args[1]);
               auto a = n->kind == Expr::COL ? aggCol.find(n->name) : aggCol.end();
               auto b = m->kind == Expr::COL ? aggCol.find(m->name) : aggCol.end();
               if (a != aggCol.end() && b != aggCol.end() && a->second.first == b->second.first) {
                  OutStep& g = steps[(size_t) a->second.first];
                  const OutAgg& sum = g.aggs[(size_t) a->second.second];
                  const OutAgg& cnt = g.aggs[(size_t) b->second.second];
                  if (sum.fn == "sum" && (cnt.fn == "count_star" || (cnt.fn == "count" && cnt.expr == sum.expr))) {
                     OutAgg avg;
                     avg.fn = "avg";
                     avg.expr = sum.expr;
                     avg.as = sanitize(name);
                     aggCol[avg.as] = {a->second.first, (int) g.aggs.size()};
                     g.aggs.push_back(avg);
                     e = mk(Expr::COL, avg.as);
                  }
               }
            }
            s.cols[name] = e;
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "filter") {
         Stream s = input(op, c);
         if (op.sOr("semantic", "all_true") != "all_true") { // none of the columns true: only the anti forms of the marker idioms
            for (auto& col : op.at("columns").arr) {
               auto it = s.cols.find(col.s("displayName"));
               if (it == s.cols.end()) throw Unsupported("filter on an undefined column");
               const Expr::Kind k = stripCast(it->second)->kind;
               if (k == Expr::MARKER && s.pending) {
                  if (c.outerBody) s.antiBranch = true; // outer / single join: the partner-less rows are null-extended and united with the matches
                  else finishProbeSide(s, true);
               } else if (k == Expr::FLAG && c.outerStep) {
                  s.flagAntiPending = true; // decided by what consumes the rows (realiseFlag)
               } else if (k == Expr::FLAG) {
                  StateP buf = states.at(s.flagState);
                  if (buf->antiRel.empty()) buf->antiRel = emitJoin(*buf->flagProbe, "anti_build");
                  s.rel = buf->antiRel;
                  s.preds.clear();
                  s.bareTable = false;
                  s.flagState.clear();
               } else throw Unsupported("filter with all_false semantic on a computed predicate");
            }
            c.streams[ref] = s;
            return;
         }
         for (auto& col : op.at("columns").arr) {
            auto it = s.cols.find(col.s("displayName"));
            if (it == s.cols.end()) throw Unsupported("filter on an undefined column");
            const ExprP e = it->second;
            const ExprP p = stripCast(e);
            if (p->kind == Expr::MARKER) { // anyTuple's marker: the probe row has (all_true) / lacks (all_false) a partner
               if (!s.pending) throw Unsupported("marker filter without a pending hash join");
               finishProbeSide(s, false);
               continue;
            }
            if (p->kind == Expr::OP && p->name == "not" && stripCast(p->args[0])->kind == Expr::MARKER) { // `not mark` of a mark join
               if (!s.pending) throw Unsupported("marker filter without a pending hash join");
               finishProbeSide(s, true);
               continue;
            }
            if (p->kind == Expr::FLAG) { // flag member of a build buffer: build rows with a partner
               StateP buf = states.at(s.flagState);
               if (buf->semiRel.empty()) buf->semiRel = emitJoin(*buf->flagProbe, "semi_build");
               s.rel = buf->semiRel;
               s.preds.clear(); // (the build side's restrictions were applied when its hash table was built)
               s.bareTable = false;
               s.flagState.clear();
               continue;
            }
            if (p->kind == Expr::CONST_BOOL && p->i) continue; // (the `matched` marker of an inner group join: true for every group the join produced)
            if (!s.gjState.empty()) { // the group join's predicate: applied to the joined rows before they are aggregated
               states.at(s.gjState)->gjFilters.push_back(e);
               continue;
            }
            if (p->kind == Expr::OP && p->name == "setop") { // INTERSECT / EXCEPT (distinct): keep the keys whose counters satisfy the predicate
               if (s.setState.empty() || (p->cmp != "intersect" && p->cmp != "except")) throw Unsupported("filter on the counters of a set operation that is not INTERSECT / EXCEPT (distinct)");
               emitSetOp(s, p->cmp);
               continue;
            }
            if (s.inJoinBody && !s.probeHiv.empty()) {
               addJoinConjunct(s, e);
               continue;
            }
            if (!s.nlBuild.empty()) {
               emitNestedLoop(s, e);
               continue;
            }
            settle(s);
            applyFilter(s, e);
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "union") {
         std::vector<Stream*> ins;
         for (auto& e : op.at("outerEdges").arr)
            if (e.s("type") == "stream") {
               auto it = c.streams.find(e.at("input").s("ref"));
               if (it == c.streams.end()) throw Unsupported("stream input '" + e.at("input").s("ref") + "' is not produced in this step");
               ins.push_back(&it->second);
            }
         if (ins.size() != 2) throw Unsupported("union of " + std::to_string(ins.size()) + " streams");
         Stream* m = ins[0]->antiBranch ? ins[1] : ins[0];
         Stream* n = ins[0]->antiBranch ? ins[0] : ins[1];
         // matches (columns mapped to their nullable copies) ∪ partner-less probe rows (the same columns mapped to NULL) of ONE
         // pending hash join = a left outer join (OuterJoinLowering / SingleJoinLowering without reverseSides)
         if (m->pending && !m->antiBranch && !m->probeHiv.empty() && n->antiBranch && !n->flagState.empty() && states.count(n->flagState) &&
             states.at(m->probeHiv)->source == states.at(n->flagState)) {
            // OuterJoinLowering with reverseSides (:1511-1525): the probing side's matches (the preserved BUILD side carries a flag they set) ∪ the
            // build rows whose flag stayed false, the probe side's columns NULL = the inner pairs followed by the unmatched build rows
            Stream s = *m;
            s.rel = emitJoin(s, s.fullOuter ? "full_outer" : "right_outer"); // (FullOuterJoinLowering, :1446-1484: the body united the matches with the partner-less probe rows)
            s.fullOuter = false;
            const Stream& bs = states.at(s.probeHiv)->source->in;
            s.names.insert(bs.names.begin(), bs.names.end());
            clearJoin(s);
            s.unique.clear();
            c.streams[ref] = s;
            return;
         }
         if (!m->pending && !n->pending && !m->antiBranch && !n->antiBranch && !m->lastMapped.empty() && m->lastMapped == n->lastMapped) {
            // UnionAllLowering (:622-634): both inputs mapped to the result columns (as nullable), then united
            Stream a = *m, b = *n;
            std::vector<std::string> lc, rc;
            for (auto& name : a.lastMapped) {
               lc.push_back(ensureCol(a, a.cols.at(name), name));
               rc.push_back(ensureCol(b, b.cols.at(name), name));
            }
            flush(a);
            flush(b);
            OutStep u;
            u.op = "set_op";
            u.out = fresh("u");
            u.fields = {{"kind", "\"union_all\""}, {"left", quote(a.rel)}, {"left_cols", nameList(lc)}, {"right", quote(b.rel)}, {"right_cols", nameList(rc)}};
            steps.push_back(u);
            Stream s;
            s.rel = u.out;
            for (size_t k = 0; k < lc.size(); k++) {
               s.cols[a.lastMapped[k]] = mk(Expr::COL, lc[k]);
               s.names.insert(lc[k]);
            }
            c.streams[ref] = s;
            return;
         }
         if (!(m->pending && !m->antiBranch && n->pending && n->antiBranch && m->probeHiv == n->probeHiv && !m->probeHiv.empty()))
            throw Unsupported("union of two streams that are neither the halves of an outer join nor two mapped inputs of a UNION ALL");
         Stream s = *m;
         for (auto& kv : n->cols)
            if (kv.second->kind == Expr::NULLV && !s.cols.count(kv.first)) throw Unsupported("outer join: column '" + kv.first + "' is NULL on one side and undefined on the other");
         if (!states.at(s.probeHiv)->source->flagMember.empty()) { // the build entries carry a flag the matches set: a FULL outer join, emitted with its third part
            s.fullOuter = true;
            c.streams[ref] = s;
            return;
         }
         s.rel = emitJoin(s, "left_outer");
         {
            const Stream& bs = states.at(s.probeHiv)->source->in;
            s.names.insert(bs.names.begin(), bs.names.end());
         }
         clearJoin(s);
         for (auto it = s.cols.begin(); it != s.cols.end();) it = it->second->kind == Expr::MARKER ? s.cols.erase(it) : std::next(it);
         c.streams[ref] = s;
         return;
      }
      if (kind == "lookup" || kind == "lookup_or_insert") {
         Stream s = input(op, c);
         StateP st = resolve(op.at("accesses").arr.at(0), c);
         const std::string stateType = op.sOr("stateType", "");
         std::string id;
         for (auto& kv : c.local)
            if (kv.second == st) id = kv.first;
         if (id.empty())
            for (auto& kv : states)
               if (kv.second == st) id = kv.first;
         if (stateType == "SimpleState" && kind == "lookup" && st->kind == State::AGG && !s.winView.empty() && st->in.winStatic == s.winView) {
            // the whole partition's aggregates, computed once by a scan of the view (frame UNBOUNDED PRECEDING … UNBOUNDED FOLLOWING)
            states[id] = st;
            s.winLookup = id;
            s.cols[op.at("reference").s("displayName")] = wref("w:static");
            c.streams[ref] = s;
            return;
         }
         if (stateType == "SegmentTreeView" && kind == "lookup") { // the aggregates of the frame [keys[0], keys[1]] (the keys: emitter extension E8)
            const J* keys = op.get("keys");
            if (st->kind != State::SEGTREE || s.winView.empty() || !keys || keys->arr.size() != 2) throw Unsupported("segment-tree lookup without its frame references (emitter extension E8)");
            auto b = s.cols.find(keys->arr[0].s("displayName")), e = s.cols.find(keys->arr[1].s("displayName"));
            if (b == s.cols.end() || e == s.cols.end()) throw Unsupported("segment-tree lookup through undefined references");
            setFrame(s, frameEnd(b->second), frameEnd(e->second), true);
            states[id] = st;
            s.winLookup = id;
            s.cols[op.at("reference").s("displayName")] = wref("w:lookup");
            c.streams[ref] = s;
            return;
         }
         if (!(stateType == "SimpleState" && kind == "lookup")) settle(s);
         if (stateType == "HashIndexedView" && kind == "lookup") {
            if (st->kind != State::HIV) throw Unsupported("lookup into a hash-indexed view that was not created from a buffer");
            if (!s.probeHiv.empty()) throw Unsupported("nested hash-indexed-view lookups");
            states[id] = st;
            s.probeHiv = id;
         } else if (stateType == "HashMap" && kind == "lookup") { // the second input of a group join looks its group up (an optional reference)
            if (st->kind != State::AGG || st->gjRight) throw Unsupported("lookup into a hash map that no other pipeline has filled");
            states[id] = st;
            s.gjState = id;
         } else if (stateType == "SimpleState" && kind == "lookup" && st->kind == State::CONST1) {
            states[id] = st;
            s.constState = id;
         } else if ((stateType == "SimpleState" && kind == "lookup") || (stateType == "HashMap" && kind == "lookup_or_insert")) {
            if (st->kind != State::UNKNOWN && !(st->kind == State::AGG && kind == "lookup_or_insert" && !st->setRight)) throw Unsupported("aggregation into a state that is already in use");
            states[id] = st;
            s.aggState = id;
         } else {
            throw Unsupported(kind + " on a " + (stateType.empty() ? "state" : stateType));
         }
         s.cols[op.at("reference").s("displayName")] = mk(Expr::REF);
         c.streams[ref] = s;
         return;
      }
      if (kind == "reduce") {
         Stream s = input(op, c);
         settle(s);
         if (const J* mat = op.get("materialized")) { // WindowLowering with PARTITION BY: the reduce appends the tuple to its partition's buffer (emitter extension E9)
            if (s.aggState.empty()) throw Unsupported("reduce without a preceding lookup into an aggregation state");
            StateP st = states.at(s.aggState);
            if (st->kind != State::UNKNOWN) throw Unsupported("two pipelines reduce into one state");
            st->kind = State::BUFFER;
            st->partPending = true;
            for (auto& m : mat->arr) {
               auto it = s.cols.find(m.at("column").s("displayName"));
               if (it == s.cols.end()) throw Unsupported("materialize of an undefined column '" + m.at("column").s("displayName") + "'");
               st->members[m.s("member")] = it->second;
            }
            s.aggState.clear();
            st->in = s;
            c.streams[ref] = s;
            return;
         }
         if (s.aggState.empty() && s.gjState.empty()) throw Unsupported("reduce without a preceding lookup into an aggregation state");
         const bool gj = s.aggState.empty();
         StateP st = states.at(gj ? s.gjState : s.aggState);
         std::vector<AggSpec> mine;
         for (auto& u : op.at("updated").arr) {
            AggSpec a;
            a.member = u.s("member");
            ExprP e = convert(u.at("expression"), s);
            auto isMember = [&](const ExprP& x) { return x->kind == Expr::MEMBER && x->name == a.member; };
            if (isMember(e)) continue; // a member returned unchanged (the other input's counter of a set operation)
            // SumAggrFunc / CountAggrFunc / CountStarAggrFunc / Min / Max bodies (RelAlgToSubOp.cpp:1805-2020)
            if (e->kind == Expr::OP && e->name == "select" && e->args[0]->kind == Expr::OP && e->args[0]->name == "isnull" && isMember(e->args[0]->args[0]) &&
                e->args[2]->kind == Expr::OP && e->args[2]->name == "add") { // nullable state: isnull(state) ? arg : state + arg
               a.fn = "sum";
               a.arg = e->args[1];
            } else if (e->kind == Expr::OP && e->name == "add" && isMember(e->args[0])) {
               if (e->args[1]->kind == Expr::CONST_INT && e->args[1]->i == 1) a.fn = "count_star";
               else {
                  a.fn = "sum";
                  a.arg = e->args[1];
               }
            } else if (e->kind == Expr::OP && e->name == "add" && e->args[0]->kind == Expr::OP && e->args[0]->name == "select" && e->args[0]->args[0]->kind == Expr::OP &&
                       e->args[0]->args[0]->name == "isnull" && isMember(e->args[0]->args[0]->args[0]) && e->args[1]->kind == Expr::OP && e->args[1]->name == "select" &&
                       e->args[1]->args[0]->kind == Expr::OP && e->args[1]->args[0]->name == "isnull") {
               // nullable state, nullable argument: (isnull(state) ? 0 : val(state)) + (isnull(arg) ? 0 : val(arg)) — NULLs do not count
               a.fn = "sum";
               a.arg = e->args[1]->args[0]->args[0];
            } else if (e->kind == Expr::OP && e->name == "select" && e->args[0]->kind == Expr::OP && e->args[0]->name == "isnull" && !isMember(e->args[0]->args[0]) && isMember(e->args[1]) &&
                       e->args[2]->kind == Expr::OP && e->args[2]->name == "add" && isMember(e->args[2]->args[0])) { // CountAggrFunc over a nullable argument: isnull(arg) ? state : state + 1
               a.fn = "count";
               a.arg = e->args[0]->args[0];
            } else if (e->kind == Expr::OP && e->name == "select" && isMember(e->args[2]) && e->args[0]->kind == Expr::OP && (e->args[0]->name == "cmp" || e->args[0]->name == "or")) {
               // Min / Max: state > arg ? arg : state; with a nullable state the condition is (state > arg) or isnull(state) (arith.ori, emitter extension E7)
               ExprP cmp = e->args[0];
               if (cmp->name == "or") {
                  ExprP found;
                  for (auto& x : cmp->args)
                     if (x->kind == Expr::OP && x->name == "cmp") found = x;
                  if (!found) throw Unsupported("aggregate body with an unrecognised condition");
                  cmp = found;
               }
               if (!isMember(cmp->args[0])) throw Unsupported("aggregate body with an unrecognised comparison");
               a.fn = cmp->cmp == "GT" ? "min" : cmp->cmp == "LT" ? "max" : "";
               a.arg = e->args[1];
               if (a.fn.empty()) throw Unsupported("aggregate body with an unrecognised comparison");
            } else if (e->kind == Expr::COL || (e->kind == Expr::OP && e->name == "cast" && stripCast(e)->kind == Expr::COL)) { // AnyAggrFunc: the state becomes the argument
               a.fn = "any";
               a.arg = e;
            } else {
               throw Unsupported("aggregate body of member '" + a.member + "' is not sum / count / min / max");
            }
            if (a.arg && (a.arg->kind == Expr::UNKNOWN || a.arg->kind == Expr::MEMBER)) throw Unsupported("aggregate argument without a device form");
            mine.push_back(a);
         }
         if (gj) { // the aggregates of a group join, over the tuples of this input that found their group
            st->gjAggs = mine;
            st->gjRight = std::make_shared<Stream>(s);
            st->gjRight->gjState.clear();
            c.streams[ref] = s;
            return;
         }
         if (st->kind == State::AGG) { // the second input of UnionDistinctLowering / CountingSetOperationLowering (:636-915) counts into the same map
            bool counters = !st->setRight && mine.size() <= 1 && st->aggs.size() == 1 && st->aggs[0].fn == "count_star" && (mine.empty() || mine[0].fn == "count_star");
            if (!counters || (mine.empty() != (st->aggs[0].member == "distinct$count"))) throw Unsupported("two pipelines reduce into one state");
            st->setRight = std::make_shared<Stream>(s);
            st->setRight->aggState.clear();
            st->setLeftCounter = mine.empty() ? "" : st->aggs[0].member;
            st->setRightCounter = mine.empty() ? "" : mine[0].member;
            s.aggState.clear();
            c.streams[ref] = s;
            return;
         }
         if (st->kind != State::UNKNOWN) throw Unsupported("two pipelines reduce into one state");
         st->aggs = mine;
         if (st->aggs.empty()) { // ProjectionDistinctLowering: a map of keys only
            AggSpec a;
            a.member = "distinct$count";
            a.fn = "count_star";
            st->aggs.push_back(a);
         }
         st->kind = State::AGG;
         s.aggState.clear();
         st->in = s;
         c.streams[ref] = s;
         return;
      }
      if (kind == "materialize") {
         Stream s = input(op, c);
         StateP st = resolve(op.at("accesses").arr.at(0), c);
         const std::string stateType = op.sOr("stateType", "");
         settle(s);
         if (!s.probeHiv.empty()) throw Unsupported("materialize between a lookup and its join predicate");
         if (stateType == "Heap") {
            if (st->kind != State::HEAP) throw Unsupported("materialize into a heap create_heap did not describe");
         } else if (stateType == "Buffer" || stateType == "ResultTable") {
            if (st->kind != State::UNKNOWN) throw Unsupported("two pipelines materialise into one " + stateType);
            st->kind = stateType == "Buffer" ? State::BUFFER : State::RESULT;
         } else {
            throw Unsupported("materialize into a " + stateType);
         }
         std::vector<std::string> order;
         for (auto& m : op.at("mapping").arr) {
            auto it = s.cols.find(m.at("column").s("displayName"));
            if (it == s.cols.end()) throw Unsupported("materialize of an undefined column '" + m.at("column").s("displayName") + "'");
            st->members[m.s("member")] = it->second;
            order.push_back(m.s("member"));
         }
         st->in = s;
         if (st->kind == State::RESULT) {
            flush(st->in);
            std::string cols = "[";
            for (size_t k = 0; k < order.size(); k++) {
               ExprP& e = st->members[order[k]];
               if (stripCast(e)->kind == Expr::HASH || stripCast(e)->kind == Expr::REF) throw Unsupported("result column that is a hash or a reference");
               const std::string col = ensureCol(st->in, e, stripSuffix(order[k]));
               e = mk(Expr::COL, col);
               cols += (k ? ", " : "") + quote(col);
            }
            OutStep m;
            m.op = "materialize";
            m.out = "result";
            m.fields = {{"in", quote(st->in.rel)}, {"cols", cols + "]"}};
            steps.push_back(m);
            result = "result";
         }
         c.streams[ref] = s;
         return;
      }
      if (kind == "scatter") { // only as part of the two marker idioms of semi / anti joins
         Stream s = input(op, c);
         if (!s.gjState.empty()) { // inner group join: the matched groups get their marker set
            states.at(s.gjState)->gjInner = true;
            c.streams[ref] = s;
            return;
         }
         if (!s.pending && !s.aggState.empty() && s.probeHiv.empty()) { // constant single join: the one row of this stream becomes the state
            StateP st = states.at(s.aggState);
            if (st->kind != State::UNKNOWN) throw Unsupported("scatter into a state that is already in use");
            st->kind = State::CONST1;
            for (auto& m : op.at("mapping").arr) {
               auto it = s.cols.find(m.at("column").s("displayName"));
               if (it == s.cols.end()) throw Unsupported("scatter of an undefined column");
               st->members[m.s("member")] = it->second;
            }
            s.aggState.clear();
            st->in = s;
            c.streams[ref] = s;
            return;
         }
         if (!s.pending) throw Unsupported("scatter outside the marker idiom of a semi / anti join");
         for (auto& m : op.at("mapping").arr) {
            auto it = s.cols.find(m.at("column").s("displayName"));
            if (it == s.cols.end() || it->second->kind != Expr::CONST_BOOL || !it->second->i) throw Unsupported("scatter of anything but the constant true");
         }
         if (!s.aggState.empty()) { // anyTuple (RelAlgToSubOp.cpp:1296-1305): a per-probe-row marker state → semi / anti keeping the probe side
            StateP st = states.at(s.aggState);
            st->kind = State::MARKER;
            s.aggState.clear();
            st->in = s;
         } else { // translateNLWithMarker: the flag member of the matched build entry → semi / anti keeping the BUILD side
            StateP hiv = states.at(s.probeHiv);
            StateP buf = hiv->source;
            const std::string& member = op.at("mapping").arr.at(0).s("member");
            if (!buf->members.count(member)) throw Unsupported("scatter into member '" + member + "' that the build side did not materialise");
            if (!buf->flagMember.empty()) throw Unsupported("two joins set flags in one build buffer");
            buf->flagMember = member;
            buf->flagProbe = std::make_shared<Stream>(s); // the join is emitted when the flag is read: all_true → semi_build, all_false → anti_build
            s.pending = false;
         }
         c.streams[ref] = s;
         return;
      }
      throw Unsupported("sub-operator '" + kind + "' has no device pattern");
   }

   void addReport(const std::string& ref, bool gpu, const std::string& reason, size_t nSubops) {
      if (!report.empty()) report += ", ";
      report += "{\"ref\": " + quote(ref) + ", \"subops\": " + std::to_string(nSubops) + ", \"target\": " + (gpu ? "\"gpu\"" : "\"cpu\"") + (reason.empty() ? "" : ", \"reason\": " + quote(reason)) + "}";
   }

   bool run(const J& plan, std::string* err) {
      if (plan.kind != J::ARR) throw std::runtime_error("subop dump: the document must be the array ToJson::run prints");
      // the manifest of the patched emitter (integration/mlir-subop-to-json.patch, --gpu-manifest): which of the extensions E1 … E10 produced this
      // document.  A dump without one comes from the unpatched tool, whose output this consumer would MIS-translate silently in two places (db.sub
      // printed as " + ", db.between without its inclusivity flags) — refused as a whole
      bool manifest = false;
      for (auto& node : plan.arr)
         if (node.kind == J::OBJ && node.sOr("type", "") == "emitter_manifest") {
            manifest = true;
            if (const J* x = node.get("extensions"))
               for (auto& v : x->arr) ext.insert(v.str);
         }
      if (!manifest) {
         *err = "the dump carries no emitter manifest ({\"type\": \"emitter_manifest\", \"extensions\": [...]}, written by the patched tool: "
                "integration/mlir-subop-to-json.patch): the unpatched mlir-subop-to-json prints db.sub as ' + ' (needs E1) and drops the inclusivity flags of "
                "db.between (needs E10) — refusing the document instead of mis-translating it";
         addReport("", false, *err, 0);
         return false;
      }
      bool ok = true;
      for (auto& node : plan.arr) {
         if (node.sOr("type", "") == "emitter_manifest") continue;
         const std::string& ref = node.s("ref");
         StepCtx c;
         const bool isStep = node.sOr("type", "") == "execution_step";
         size_t nSub = isStep ? node.at("subops").arr.size() : 1;
         if (!ok) { // after the first unsupported step nothing downstream can be placed
            addReport(ref, false, "depends on a step that stays on the CPU", nSub);
            continue;
         }
         try {
            if (isStep) {
               for (auto& e : node.at("outerEdges").arr) {
                  if (e.s("type") != "requiredInput") continue;
                  const std::string id = e.at("input").s("ref") + "#" + std::to_string(e.at("input").iOr("resnr", 0));
                  const size_t argnr = (size_t) e.at("output").iOr("argnr", 0);
                  if (c.args.size() <= argnr) c.args.resize(argnr + 1);
                  auto it = states.find(id);
                  c.args[argnr] = it == states.end() ? nullptr : it->second;
               }
               for (auto& op : node.at("subops").arr) c.outerStep = c.outerStep || (op.get("subop") && op.s("subop") == "union");
               for (auto& op : node.at("subops").arr) handle(op, c);
               if (const J* ie = node.get("innerEdges"))
                  for (auto& e : ie->arr) {
                     if (e.s("type") != "resultEdge") continue;
                     const std::string id = e.at("input").s("ref") + "#" + std::to_string(e.at("input").iOr("resnr", 0));
                     auto it = c.local.find(id);
                     if (it == c.local.end()) throw Unsupported("step result '" + id + "' is not a state this consumer tracks");
                     states[ref + "#" + std::to_string(e.at("output").iOr("resnr", 0))] = it->second;
                  }
            } else {
               handle(node, c);
               for (auto& kv : c.local) states[kv.first] = kv.second;
            }
            addReport(ref, true, "", nSub);
         } catch (const Unsupported& u) {
            ok = false;
            addReport(ref, false, u.what(), nSub);
            if (err->empty()) *err = "step " + ref + ": " + u.what();
         }
      }
      if (ok && result.empty()) {
         ok = false;
         *err = "the dump never materialises into a ResultTable";
      }
      return ok;
   }

   std::string planJson(const std::string& name) const {
      std::string o = "{\"name\": " + quote(name) + ", \"doc\": \"translated from a mlir-subop-to-json dump by ldb_subop_translate\", \"inputs\": [";
      size_t k = 0;
      for (auto& t : inputs) o += (k++ ? ", " : "") + quote(t);
      o += "],\n \"steps\": [\n";
      for (size_t s = 0; s < steps.size(); s++) {
         const OutStep& st = steps[s];
         o += "  {\"op\": " + quote(st.op);
         for (auto& f : st.fields) o += ", " + quote(f.first) + ": " + f.second;
         if (st.op == "groupby") {
            o += ", \"aggs\": [";
            bool first = true;
            for (auto& a : st.aggs) {
               if (!a.used) continue; // aggregates nothing downstream reads (the sum / count halves of an avg)
               o += std::string(first ? "" : ", ") + "{\"fn\": " + quote(a.fn) + (a.expr.empty() ? "" : ", \"expr\": " + a.expr) + (a.when.empty() ? "" : ", \"when\": " + a.when) +
                    (a.type.empty() ? "" : ", \"type\": " + quote(a.type)) + ", \"as\": " + quote(a.as) + "}";
               first = false;
            }
            o += "]";
         }
         o += ", \"out\": " + quote(st.out) + "}" + (s + 1 < steps.size() ? ",\n" : "\n");
      }
      return o + " ], \"result\": " + quote(result) + "}\n";
   }
};

std::string g_subop_err, g_subop_report;

} // namespace

// the translation of one dump: LDB_OK and the plan text (NUL-terminated) in plan_out; LDB_ERR_UNSUPPORTED when a step has
// no device pattern (ldb_subop_report() says which); LDB_ERR_INVALID for a malformed document or a too-small buffer
// (*needed receives the size to retry with)
extern "C" int32_t ldb_subop_translate(const char* dump_json, const char* name, char* plan_out, int64_t cap, int64_t* needed) {
   g_subop_err.clear();
   g_subop_report = "[]";
   if (needed) *needed = 0;
   if (!dump_json) {
      g_subop_err = "null dump";
      return LDB_ERR_INVALID;
   }
   try {
      JParser jp(dump_json);
      const J doc = jp.value();
      Translator t;
      const bool ok = t.run(doc, &g_subop_err);
      g_subop_report = "[" + t.report + "]";
      if (!ok) return LDB_ERR_UNSUPPORTED;
      const std::string text = t.planJson(name ? name : "subop_dump");
      if (needed) *needed = (int64_t) text.size() + 1;
      if (!plan_out || cap < (int64_t) text.size() + 1) {
         g_subop_err = "output buffer too small";
         return LDB_ERR_INVALID;
      }
      memcpy(plan_out, text.c_str(), text.size() + 1);
      return LDB_OK;
   } catch (const std::exception& e) {
      g_subop_err = e.what();
      return LDB_ERR_INVALID;
   }
}
extern "C" const char* ldb_subop_last_error(void) { return g_subop_err.c_str(); }
// per execution step: {"ref", "subops", "target": "gpu" | "cpu", "reason"} — the placement handleExecutionStepGPU would make
extern "C" const char* ldb_subop_report(void) { return g_subop_report.c_str(); }
