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
const char* ldb_plan_last_error(void);
// Plans as data (ldb_plan.cpp): a JSON step list — the shape of the reference's execution-step dump
// (tools/ct/mlir-subop-to-json.cpp) at the granularity of the C-ABI — interpreted over the named
// input tables; *result = the table the plan names as its result (caller releases).
int32_t ldb_plan_run_json(ldb_ctx* ctx, const char* plan_json, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables, ldb_table** result);
int32_t ldb_plan_run_json_comm(ldb_ctx* ctx, ldb_comm* comm, const char* plan_json, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables,
                               ldb_table** result);
const char* ldb_plan_json_last_error(void);
// Prepared plans: parse once, execute many times.  Executions after the first over the same input tables and options run
// without waiting for the device between operators (read-back trace, include/lingodb_gpu.h) and without descriptor uploads.
typedef struct ldb_plan ldb_plan;
int32_t ldb_plan_prepare(ldb_ctx* ctx, const char* plan_json, ldb_plan** out);
int32_t ldb_plan_execute(ldb_plan* plan, ldb_comm* comm, const char* const* table_names, const ldb_table* const* tables, int32_t n_tables, ldb_table** result);
int32_t ldb_plan_release(ldb_plan* plan);
int32_t ldb_plan_stats(const ldb_plan* plan, int64_t* executions, int64_t* replays, int64_t* misses, int64_t* readbacks);
// over the replayed executions so far: host time spent issuing the plans / waiting in the single check at their ends
int32_t ldb_plan_times(const ldb_plan* plan, double* issue_ms, double* wait_ms);
// structure check without a device: parse, known steps with their required fields, values defined before use
int32_t ldb_plan_json_check(const char* plan_json, const char* const* input_names, int32_t n_inputs);
// ldb_subop.cpp: consumer of the reference's sub-operator dump (tools/ct/mlir-subop-to-json.cpp).  Translates the
// execution_step / subops document into the step list above; LDB_ERR_UNSUPPORTED when an execution step has no device
// pattern (ldb_subop_report: per step {"ref","subops","target":"gpu"|"cpu","reason"}), LDB_ERR_INVALID for a malformed
// document or a too-small buffer (*needed = bytes to retry with).
int32_t ldb_subop_translate(const char* dump_json, const char* name, char* plan_out, int64_t cap, int64_t* needed);
const char* ldb_subop_last_error(void);
const char* ldb_subop_report(void);
// test hooks for the host logic (date / decimal parsing, decimal typing rules)
int32_t ldb_host_parse_date32(const char* s, int32_t* out);
int32_t ldb_host_parse_decimal(const char* s, int32_t scale, int64_t* lo, int64_t* hi);
void ldb_host_decimal_type(int32_t op, int32_t p1, int32_t s1, int32_t p2, int32_t s2, int32_t* p, int32_t* s);
}
