// ldb_host.hpp — C++ host-side mirror of the reference's scan/filter interface and the plan layer
// that drives liblingodb_gpu.so through its C-ABI only (no HIP, no torch here).
//
// Mirrors (same names, argument meaning, error behaviour = std::runtime_error):
//   lingodb::runtime::FilterOp / FilterDescription   include/lingodb/runtime/storage/TableStorage.h:14-36
//   Restrictions::create type dispatch                src/runtime/storage/Restrictions.cpp:392-521
//   parseDate32                                       src/runtime/storage/Restrictions.cpp:17-25
//   decimal result-type rules                         src/compiler/frontend/sql_analyzer.cpp:3058-3159, 2617-2642
#pragma once
#include "../../include/lingodb_gpu.h"
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

namespace lingodb::runtime::gpu {

enum class FilterOp : uint8_t { EQ, NEQ, LT, LTE, GT, GTE, NOTNULL, IN };

struct FilterDescription {
   std::string columnName;
   FilterOp op;
   std::variant<std::string, int64_t, double> value;
   std::variant<std::vector<std::string>, std::vector<int64_t>, std::vector<double>> values;
};

int32_t parseDate32(std::string str); // "YYYY-MM-DD" (also "YYYY-M-DD") → days since epoch
// decimal literal → unscaled 128-bit integer at `scale` (arrow::Decimal128::FromString + Rescale)
__int128 parseDecimal(const std::string& s, int32_t scale);

// Typed predicate list for one table: owns the string / IN-list storage the C descriptors point to.
class Restrictions {
   public:
   static std::unique_ptr<Restrictions> create(const std::vector<FilterDescription>& filterDescs, const ldb_table* table, int32_t side = 0);
   const ldb_filter_desc* data() const { return descs.data(); }
   int32_t size() const { return (int32_t) descs.size(); }

   private:
   std::vector<ldb_filter_desc> descs;
   std::vector<std::unique_ptr<std::string>> strings;
   std::vector<std::unique_ptr<std::vector<int64_t>>> inInts;
   std::vector<std::unique_ptr<std::vector<const char*>>> inStrPtrs;
   std::vector<std::unique_ptr<std::vector<int32_t>>> inStrLens;
};

// decimal(p, s) typing (SQLTypeUtils)
struct DecimalType {
   int32_t p, s;
   bool wide() const { return p >= 19; } // storage width rule, LowerToStd.cpp:1479-1486
};
DecimalType adaptedAfterMulDiv(int64_t p, int64_t s); // getAdaptedDecimalPAndSAfterMulDiv :3145-3159
DecimalType typeAfterMul(DecimalType a, DecimalType b); // :3099-3106
DecimalType typeAfterDiv(DecimalType a, DecimalType b); // :3088-3097
DecimalType higherDecimalType(DecimalType a, DecimalType b); // getHigherDecimalType :3058-3065 (add/sub/compare)
DecimalType avgType(DecimalType arg); // :2636-2642 (division by decimal(19,0))
int64_t pow10i(int k);

// RAII wrappers used by the plans
struct CtxError : std::runtime_error {
   using std::runtime_error::runtime_error;
};
void check(int32_t status, const char* what);

} // namespace lingodb::runtime::gpu

extern "C" {
// The single-GPU TPC-H plans are data: lingo-db_amd/plans/tpch/qN.json run by ldb_plan_run_json (below).
// The functions here are the PIECES of the multi-GPU plans (shard-local parts and merges between the
// exchanges).  Inputs: device tables with the TPC-H column names; result: device table
// (caller releases).
// Q5 pieces (multi-GPU all-gathers between them)
int32_t ldb_plan_tpch_q5_customers(ldb_ctx* ctx, const ldb_table* customer, const ldb_table* nation, const ldb_table* region, ldb_table** result);
int32_t ldb_plan_tpch_q5_suppliers(ldb_ctx* ctx, const ldb_table* supplier, const ldb_table* nation, const ldb_table* region, ldb_table** result);
int32_t ldb_plan_tpch_q5_local(ldb_ctx* ctx, const ldb_table* custs, const ldb_table* supps, const ldb_table* orders, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q5_final(ldb_ctx* ctx, const ldb_table* partials, const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q7_customers(ldb_ctx* ctx, const ldb_table* customer, const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q7_suppliers(ldb_ctx* ctx, const ldb_table* supplier, const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q7_local(ldb_ctx* ctx, const ldb_table* custs, const ldb_table* supps, const ldb_table* orders, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q7_final(ldb_ctx* ctx, const ldb_table* partials, const ldb_table* nation, ldb_table** result);
// Q11 pieces (suppliers → groups → [partition → all-to-all → merge] → total → filter → sort)
int32_t ldb_plan_tpch_q11_suppliers(ldb_ctx* ctx, const ldb_table* supplier, const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q11_groups(ldb_ctx* ctx, const ldb_table* suppkeys, const ldb_table* partsupp, ldb_table** result);
int32_t ldb_plan_tpch_q11_partition(ldb_ctx* ctx, const ldb_table* groups, int32_t world, ldb_table** result, int64_t* counts);
int32_t ldb_plan_tpch_q11_merge(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result);
int32_t ldb_plan_tpch_q11_total(ldb_ctx* ctx, const ldb_table* groups, ldb_table** result);
int32_t ldb_plan_tpch_q11_filter(ldb_ctx* ctx, const ldb_table* groups, const ldb_table* totals, ldb_table** result);
int32_t ldb_plan_tpch_q11_sort(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result);
// Q14 pieces (promo part keys → [all-gather] → local partial sums → [all-gather] → final ratio)
int32_t ldb_plan_tpch_q14_promo(ldb_ctx* ctx, const ldb_table* part, ldb_table** result);
int32_t ldb_plan_tpch_q14_local(ldb_ctx* ctx, const ldb_table* promokeys, const ldb_table* partkeys, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q14_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result);
// Q8 pieces (part keys, customers of the region → [all-gather] → local partial sums per year → [all-gather] → final)
int32_t ldb_plan_tpch_q8_parts(ldb_ctx* ctx, const ldb_table* part, ldb_table** result);
int32_t ldb_plan_tpch_q8_customers(ldb_ctx* ctx, const ldb_table* customer, const ldb_table* nation, const ldb_table* region, ldb_table** result);
int32_t ldb_plan_tpch_q8_local(ldb_ctx* ctx, const ldb_table* partkeys, const ldb_table* custs, const ldb_table* supplier, const ldb_table* orders, const ldb_table* lineitem,
                               const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q8_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result);
const char* ldb_plan_last_error(void);
// Plans as data (ldb_plan.cpp): a JSON step list — the shape of the reference's execution-step dump
// (tools/ct/mlir-subop-to-json.cpp) at the granularity of the C-ABI — interpreted over the named
// input tables; *result = the table the plan names as its result (caller releases).
int32_t ldb_plan_run_json(ldb_ctx* ctx, const char* plan_json, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables, ldb_table** result);
const char* ldb_plan_json_last_error(void);
// structure check without a device: parse, known steps with their required fields, values defined before use
int32_t ldb_plan_json_check(const char* plan_json, const char* const* input_names, int32_t n_inputs);
// multi-GPU pieces: shard-local partial plans + merges of the exchanged partial tables
int32_t ldb_plan_tpch_q1_partial(ldb_ctx* ctx, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q1_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result);
int32_t ldb_plan_tpch_q6_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result);
int32_t ldb_plan_tpch_q3_customers(ldb_ctx* ctx, const ldb_table* customer, ldb_table** result);
int32_t ldb_plan_tpch_q3_local(ldb_ctx* ctx, const ldb_table* custkeys, const ldb_table* orders, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q3_final(ldb_ctx* ctx, const ldb_table* tops, ldb_table** result);
int32_t ldb_plan_tpch_q4_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result);
int32_t ldb_plan_tpch_q12_final(ldb_ctx* ctx, const ldb_table* partials, ldb_table** result);
// Q10 pieces: shard-local (o_custkey, revenue) groups; [multi-GPU: partition + merge on the key;] top 20; names; order
int32_t ldb_plan_tpch_q10_local(ldb_ctx* ctx, const ldb_table* orders, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q10_partition(ldb_ctx* ctx, const ldb_table* groups, int32_t world, ldb_table** result, int64_t* counts);
int32_t ldb_plan_tpch_q10_merge(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result);
int32_t ldb_plan_tpch_q10_top(ldb_ctx* ctx, const ldb_table* groups, ldb_table** result);
int32_t ldb_plan_tpch_q10_names(ldb_ctx* ctx, const ldb_table* top20, const ldb_table* customer, const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q10_final(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result);
// Q15 pieces: shard-local (l_suppkey, revenue) groups; [multi-GPU: partition + merge;] best group; groups equal to it; supplier join + order
int32_t ldb_plan_tpch_q15_local(ldb_ctx* ctx, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q15_partition(ldb_ctx* ctx, const ldb_table* groups, int32_t world, ldb_table** result, int64_t* counts);
int32_t ldb_plan_tpch_q15_merge(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result);
int32_t ldb_plan_tpch_q15_max(ldb_ctx* ctx, const ldb_table* groups, ldb_table** result);
int32_t ldb_plan_tpch_q15_winners(ldb_ctx* ctx, const ldb_table* groups, const ldb_table* best, ldb_table** result);
int32_t ldb_plan_tpch_q15_final(ldb_ctx* ctx, const ldb_table* winners, const ldb_table* supplier, ldb_table** result);
int32_t ldb_plan_tpch_q18_local(ldb_ctx* ctx, const ldb_table* orders, const ldb_table* lineitem, ldb_table** result);
int32_t ldb_plan_tpch_q18_mid(ldb_ctx* ctx, const ldb_table* tops, ldb_table** result);
int32_t ldb_plan_tpch_q18_names(ldb_ctx* ctx, const ldb_table* top100, const ldb_table* customer, ldb_table** result);
int32_t ldb_plan_tpch_q18_final(ldb_ctx* ctx, const ldb_table* rows, ldb_table** result);
// Q9: green part keys → (all-gather) → lineitem side / partsupp side partitioned by hash-radix of the
// part key (counts[world] rows per destination) → (two all-to-alls) → local joins + partial sums →
// (all-gather) → final
int32_t ldb_plan_tpch_q9_green(ldb_ctx* ctx, const ldb_table* part, ldb_table** result);
int32_t ldb_plan_tpch_q9_lineitem_side(ldb_ctx* ctx, const ldb_table* greenkeys, const ldb_table* lineitem, const ldb_table* orders, int32_t world, ldb_table** result,
                                       int64_t* counts);
int32_t ldb_plan_tpch_q9_partsupp_side(ldb_ctx* ctx, const ldb_table* greenkeys, const ldb_table* partsupp, int32_t world, ldb_table** result, int64_t* counts);
int32_t ldb_plan_tpch_q9_join(ldb_ctx* ctx, const ldb_table* lrows, const ldb_table* psrows, const ldb_table* supplier, const ldb_table* nation, ldb_table** result);
int32_t ldb_plan_tpch_q9_final(ldb_ctx* ctx, const ldb_table* partials, const ldb_table* nation, ldb_table** result);
// test hooks for the host logic (date / decimal parsing, decimal typing rules)
int32_t ldb_host_parse_date32(const char* s, int32_t* out);
int32_t ldb_host_parse_decimal(const char* s, int32_t scale, int64_t* lo, int64_t* hi);
void ldb_host_decimal_type(int32_t op, int32_t p1, int32_t s1, int32_t p2, int32_t s2, int32_t* p, int32_t* s);
}
