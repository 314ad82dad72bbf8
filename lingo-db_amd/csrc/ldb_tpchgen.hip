// ldb_tpchgen.hip — device generator for the TPC-H-shaped synthetic database (include/ldb_tpchgen.h).
// Benchmark / test support: fills HBM-resident tables directly so SF100 needs no host data.
// Not part of the operator path; the data definition is shared with the host generator.
#include "ldb_internal.h"
#include "../../include/ldb_tpchgen.h"
#include <memory>

struct GenCols {
   void* values[16];
   int32_t width[16]; // device bytes per value (8 or 16 for decimals)
   uint64_t mask; // bit c set = column generated (values[c] valid)
};

__device__ __forceinline__ void put_dec(const GenCols& g, int c, uint64_t i, int64_t v) {
   switch (g.width[c]) {
      case 1: ((int8_t*) g.values[c])[i] = (int8_t) v; break; // (narrow level 2: the generator knows each column's value range, gen_width below)
      case 2: ((int16_t*) g.values[c])[i] = (int16_t) v; break;
      case 4: ((int32_t*) g.values[c])[i] = (int32_t) v; break;
      case 8: ((int64_t*) g.values[c])[i] = v; break;
      default:
         ((int64_t*) g.values[c])[2 * i] = v;
         ((int64_t*) g.values[c])[2 * i + 1] = v >> 63;
   }
}
__device__ __forceinline__ void put_i32(const GenCols& g, int c, uint64_t i, int32_t v) {
   if (g.width[c] == 1) ((int8_t*) g.values[c])[i] = (int8_t) v; // a char(1) column at one byte (narrow level 2)
   else ((int32_t*) g.values[c])[i] = v;
}
#define HAS(c) ((g.mask >> (c)) & 1)

__global__ void k_gen_lineitem(GenCols g, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      ldb_tpch_lineitem li;
      ldb_tpch_lineitem_row(row0 + (int64_t) i, n_orders, &li);
      if (HAS(L_ORDERKEY)) put_i32(g, L_ORDERKEY, i, li.orderkey);
      if (HAS(L_PARTKEY)) put_i32(g, L_PARTKEY, i, li.partkey);
      if (HAS(L_SUPPKEY)) put_i32(g, L_SUPPKEY, i, li.suppkey);
      if (HAS(L_LINENUMBER)) put_i32(g, L_LINENUMBER, i, li.linenumber);
      if (HAS(L_QUANTITY)) put_dec(g, L_QUANTITY, i, li.quantity);
      if (HAS(L_EXTENDEDPRICE)) put_dec(g, L_EXTENDEDPRICE, i, li.extendedprice);
      if (HAS(L_DISCOUNT)) put_dec(g, L_DISCOUNT, i, li.discount);
      if (HAS(L_TAX)) put_dec(g, L_TAX, i, li.tax);
      if (HAS(L_RETURNFLAG)) put_i32(g, L_RETURNFLAG, i, li.returnflag);
      if (HAS(L_LINESTATUS)) put_i32(g, L_LINESTATUS, i, li.linestatus);
      if (HAS(L_SHIPDATE)) put_i32(g, L_SHIPDATE, i, li.shipdate);
      if (HAS(L_COMMITDATE)) put_i32(g, L_COMMITDATE, i, li.commitdate);
      if (HAS(L_RECEIPTDATE)) put_i32(g, L_RECEIPTDATE, i, li.receiptdate);
   }
}
__global__ void k_gen_orders(GenCols g, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t oi = row0 + (int64_t) i;
      if (HAS(O_ORDERKEY)) put_i32(g, O_ORDERKEY, i, ldb_tpch_orderkey(oi));
      if (HAS(O_CUSTKEY)) put_i32(g, O_CUSTKEY, i, ldb_tpch_o_custkey(oi, n_orders));
      if (HAS(O_ORDERSTATUS) || HAS(O_TOTALPRICE)) {
         int32_t st;
         int64_t tp;
         ldb_tpch_order_derived(oi, n_orders, &st, &tp);
         if (HAS(O_ORDERSTATUS)) put_i32(g, O_ORDERSTATUS, i, st);
         if (HAS(O_TOTALPRICE)) put_dec(g, O_TOTALPRICE, i, tp);
      }
      if (HAS(O_ORDERDATE)) put_i32(g, O_ORDERDATE, i, ldb_tpch_o_orderdate(oi));
      if (HAS(O_SHIPPRIORITY)) put_i32(g, O_SHIPPRIORITY, i, 0);
   }
}
__global__ void k_gen_customer(GenCols g, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t ci = row0 + (int64_t) i;
      if (HAS(C_CUSTKEY)) put_i32(g, C_CUSTKEY, i, (int32_t) (ci + 1));
      if (HAS(C_NATIONKEY)) put_i32(g, C_NATIONKEY, i, ldb_tpch_c_nationkey(ci));
      if (HAS(C_ACCTBAL)) put_dec(g, C_ACCTBAL, i, ldb_tpch_c_acctbal(ci));
   }
}
__global__ void k_gen_part(GenCols g, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t pk = row0 + (int64_t) i + 1;
      if (HAS(P_PARTKEY)) put_i32(g, P_PARTKEY, i, (int32_t) pk);
      if (HAS(P_SIZE)) put_i32(g, P_SIZE, i, (int32_t) ldb_uniform(LDB_TPCH_PART, P_SIZE, (uint64_t) (pk - 1), 1, 50));
      if (HAS(P_RETAILPRICE)) put_dec(g, P_RETAILPRICE, i, ldb_tpch_retailprice(pk));
   }
}
__global__ void k_gen_supplier(GenCols g, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t si = row0 + (int64_t) i;
      if (HAS(S_SUPPKEY)) put_i32(g, S_SUPPKEY, i, (int32_t) (si + 1));
      if (HAS(S_NATIONKEY)) put_i32(g, S_NATIONKEY, i, (int32_t) (ldb_rnd(LDB_TPCH_SUPPLIER, S_NATIONKEY, (uint64_t) si) % 25));
      if (HAS(S_ACCTBAL)) put_dec(g, S_ACCTBAL, i, ldb_uniform(LDB_TPCH_SUPPLIER, S_ACCTBAL, (uint64_t) si, -99999, 999999));
   }
}
__global__ void k_gen_partsupp(GenCols g, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t r = row0 + (int64_t) i;
      int64_t pk = r / 4 + 1;
      int32_t j = (int32_t) (r % 4);
      if (HAS(PS_PARTKEY)) put_i32(g, PS_PARTKEY, i, (int32_t) pk);
      if (HAS(PS_SUPPKEY)) put_i32(g, PS_SUPPKEY, i, ldb_tpch_ps_suppkey(pk, j, n_orders));
      if (HAS(PS_AVAILQTY)) put_i32(g, PS_AVAILQTY, i, (int32_t) ldb_uniform(LDB_TPCH_PARTSUPP, PS_AVAILQTY, (uint64_t) r, 1, 9999));
      if (HAS(PS_SUPPLYCOST)) put_dec(g, PS_SUPPLYCOST, i, ldb_uniform(LDB_TPCH_PARTSUPP, PS_SUPPLYCOST, (uint64_t) r, 100, 100000));
   }
}
__global__ void k_gen_nation(GenCols g, int64_t row0, uint64_t n) {
   const int32_t region[25] = {0, 1, 1, 1, 4, 0, 3, 3, 2, 2, 4, 4, 2, 4, 0, 0, 0, 1, 2, 3, 4, 2, 3, 3, 1};
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t r = row0 + (int64_t) i;
      if (HAS(N_NATIONKEY)) put_i32(g, N_NATIONKEY, i, (int32_t) r);
      if (HAS(N_REGIONKEY)) put_i32(g, N_REGIONKEY, i, region[r % 25]);
   }
}
__global__ void k_gen_region(GenCols g, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      if (HAS(R_REGIONKEY)) put_i32(g, R_REGIONKEY, i, (int32_t) (row0 + (int64_t) i));
}

// ---- string columns: domain index → length → exclusive scan → fill
struct StrDomain {
   int32_t n;
   int32_t off[27];
   char blob[400];
};
__global__ void k_gen_str_lens(StrDomain dom, int32_t table, int32_t col, int64_t row0, uint64_t n, int64_t* lens) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int32_t k = ldb_tpch_str_idx(table, col, row0 + (int64_t) i);
      lens[i] = dom.off[k + 1] - dom.off[k];
   }
}
__global__ void k_gen_str_fill(StrDomain dom, int32_t table, int32_t col, int64_t row0, uint64_t n, const int64_t* offs, char* out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int32_t k = ldb_tpch_str_idx(table, col, row0 + (int64_t) i);
      int32_t len = dom.off[k + 1] - dom.off[k];
      int64_t o = offs[i];
      for (int32_t b = 0; b < len; b++) out[o + b] = dom.blob[dom.off[k] + b];
   }
}

// generated text columns (ldb_tpch_text): lengths → scan → fill, each thread renders its row into a
// small local buffer
__global__ void k_gen_text_lens(int32_t table, int32_t col, int64_t row0, uint64_t n, int64_t* lens) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      char buf[LDB_TPCH_TEXT_MAX];
      lens[i] = ldb_tpch_text(table, col, row0 + (int64_t) i, buf);
   }
}
__global__ void k_gen_text_fill(int32_t table, int32_t col, int64_t row0, uint64_t n, const int64_t* offs, char* out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      char buf[LDB_TPCH_TEXT_MAX];
      const int32_t len = ldb_tpch_text(table, col, row0 + (int64_t) i, buf);
      const int64_t o = offs[i];
      for (int32_t b = 0; b < len; b++) out[o + b] = buf[b];
   }
}
// bench support: keys of uniformly random orders (LDB_TPCH_PROBEKEYS)
__global__ void k_gen_probekeys(int32_t* out, int64_t n_orders, int64_t row0, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      out[i] = ldb_tpch_orderkey((int64_t) (ldb_rnd(LDB_TPCH_PROBEKEYS, 0, (uint64_t) (row0 + (int64_t) i)) % (uint64_t) n_orders));
}

// c_name = "Customer#%09d": fixed width, so offsets are closed-form
__global__ void k_gen_cname(int64_t row0, uint64_t n, int64_t* offs, char* out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i <= n; i += (uint64_t) gridDim.x * blockDim.x) {
      offs[i] = (int64_t) i * LDB_TPCH_CNAME_LEN;
      if (i < n) ldb_tpch_c_name(row0 + (int64_t) i + 1, out + i * LDB_TPCH_CNAME_LEN);
   }
}

// p_name = five colour words separated by blanks: variable width → lengths, scan, fill
struct ColorDomain {
   int32_t off[LDB_TPCH_NCOLORS + 1];
   char blob[704];
};
__global__ void k_gen_pname_lens(ColorDomain dom, int32_t col, int32_t words, int64_t row0, uint64_t n, int64_t* lens) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t len = words - 1;
      for (int j = 0; j < words; j++) {
         int32_t k = ldb_tpch_wordcol_word(col, row0 + (int64_t) i, j);
         len += dom.off[k + 1] - dom.off[k];
      }
      lens[i] = len;
   }
}
__global__ void k_gen_pname_fill(ColorDomain dom, int32_t col, int32_t words, int64_t row0, uint64_t n, const int64_t* offs, char* out) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
      int64_t o = offs[i];
      for (int j = 0; j < words; j++) {
         int32_t k = ldb_tpch_wordcol_word(col, row0 + (int64_t) i, j);
         if (j) out[o++] = ' ';
         for (int32_t b = dom.off[k]; b < dom.off[k + 1]; b++) out[o++] = dom.blob[b];
      }
   }
}

static const char* const* domain_strings(int32_t table, int32_t col) {
   if (table == LDB_TPCH_LINEITEM && col == L_SHIPINSTRUCT) return ldb_tpch_instructs;
   if (table == LDB_TPCH_LINEITEM && col == L_SHIPMODE) return ldb_tpch_shipmodes;
   if (table == LDB_TPCH_ORDERS && col == O_ORDERPRIORITY) return ldb_tpch_priorities;
   if (table == LDB_TPCH_CUSTOMER && col == C_MKTSEGMENT) return ldb_tpch_segments;
   if (table == LDB_TPCH_NATION && col == N_NAME) return ldb_tpch_nations;
   if (table == LDB_TPCH_REGION && col == R_NAME) return ldb_tpch_regions;
   return nullptr;
}

struct ColDef {
   const char* name;
   ldb_coltype type;
};
#define CT_I32 {LDB_T_INT32, 0, 0, 0}
#define CT_DEC {LDB_T_DECIMAL128, 12, 2, 0}
#define CT_DATE {LDB_T_DATE32, 0, 0, 0}
#define CT_CH {LDB_T_CHAR4, 0, 0, 0}
#define CT_STR {LDB_T_UTF8, 0, 0, 0}
static const ColDef LINEITEM_COLS[L_NCOLS] = {{"l_orderkey", CT_I32}, {"l_partkey", CT_I32}, {"l_suppkey", CT_I32}, {"l_linenumber", CT_I32}, {"l_quantity", CT_DEC}, {"l_extendedprice", CT_DEC}, {"l_discount", CT_DEC}, {"l_tax", CT_DEC}, {"l_returnflag", CT_CH}, {"l_linestatus", CT_CH}, {"l_shipdate", CT_DATE}, {"l_commitdate", CT_DATE}, {"l_receiptdate", CT_DATE}, {"l_shipinstruct", CT_STR}, {"l_shipmode", CT_STR}};
static const ColDef ORDERS_COLS[O_NCOLS] = {{"o_orderkey", CT_I32}, {"o_custkey", CT_I32}, {"o_orderstatus", CT_CH}, {"o_totalprice", CT_DEC}, {"o_orderdate", CT_DATE}, {"o_orderpriority", CT_STR}, {"o_shippriority", CT_I32}, {"o_comment", CT_STR}};
static const ColDef CUSTOMER_COLS[C_NCOLS] = {{"c_custkey", CT_I32}, {"c_nationkey", CT_I32}, {"c_acctbal", CT_DEC}, {"c_mktsegment", CT_STR}, {"c_name", CT_STR}, {"c_phone", CT_STR}};
static const ColDef PART_COLS[P_NCOLS] = {{"p_partkey", CT_I32}, {"p_size", CT_I32}, {"p_retailprice", CT_DEC}, {"p_name", CT_STR}, {"p_type", CT_STR}, {"p_brand", CT_STR}, {"p_container", CT_STR}, {"p_mfgr", CT_STR}};
static const ColDef SUPPLIER_COLS[S_NCOLS] = {{"s_suppkey", CT_I32}, {"s_nationkey", CT_I32}, {"s_acctbal", CT_DEC}, {"s_name", CT_STR}, {"s_address", CT_STR}, {"s_phone", CT_STR}, {"s_comment", CT_STR}};
static const ColDef PARTSUPP_COLS[PS_NCOLS] = {{"ps_partkey", CT_I32}, {"ps_suppkey", CT_I32}, {"ps_availqty", CT_I32}, {"ps_supplycost", CT_DEC}};
static const ColDef NATION_COLS[N_NCOLS] = {{"n_nationkey", CT_I32}, {"n_regionkey", CT_I32}, {"n_name", CT_STR}};
static const ColDef REGION_COLS[R_NCOLS] = {{"r_regionkey", CT_I32}, {"r_name", CT_STR}};
static const ColDef PROBEKEYS_COLS[1] = {{"k_orderkey", CT_I32}};

static int table_def(int32_t table, const ColDef** cols, int* n_cols, const char** name) {
   switch (table) {
      case LDB_TPCH_LINEITEM: *cols = LINEITEM_COLS; *n_cols = L_NCOLS; *name = "lineitem"; return 0;
      case LDB_TPCH_ORDERS: *cols = ORDERS_COLS; *n_cols = O_NCOLS; *name = "orders"; return 0;
      case LDB_TPCH_CUSTOMER: *cols = CUSTOMER_COLS; *n_cols = C_NCOLS; *name = "customer"; return 0;
      case LDB_TPCH_PART: *cols = PART_COLS; *n_cols = P_NCOLS; *name = "part"; return 0;
      case LDB_TPCH_SUPPLIER: *cols = SUPPLIER_COLS; *n_cols = S_NCOLS; *name = "supplier"; return 0;
      case LDB_TPCH_PARTSUPP: *cols = PARTSUPP_COLS; *n_cols = PS_NCOLS; *name = "partsupp"; return 0;
      case LDB_TPCH_NATION: *cols = NATION_COLS; *n_cols = N_NCOLS; *name = "nation"; return 0;
      case LDB_TPCH_REGION: *cols = REGION_COLS; *n_cols = R_NCOLS; *name = "region"; return 0;
      case LDB_TPCH_PROBEKEYS: *cols = PROBEKEYS_COLS; *n_cols = 1; *name = "probekeys"; return 0;
      default: return -1;
   }
}

// rows [begin, end) of `table` owned by slice part/n_parts
static void table_slice(int32_t table, int64_t n_orders, int32_t part, int32_t n_parts, int64_t* begin, int64_t* end) {
   int64_t ob, oe;
   switch (table) {
      case LDB_TPCH_PROBEKEYS:
      case LDB_TPCH_LINEITEM:
         ldb_tpch_order_slice(n_orders, part, n_parts, &ob, &oe);
         *begin = ldb_tpch_line_offset(ob);
         *end = ldb_tpch_line_offset(oe);
         break;
      case LDB_TPCH_ORDERS: ldb_tpch_order_slice(n_orders, part, n_parts, begin, end); break;
      case LDB_TPCH_CUSTOMER: ldb_tpch_row_slice(ldb_tpch_n_customers(n_orders), part, n_parts, begin, end); break;
      case LDB_TPCH_PART: ldb_tpch_row_slice(ldb_tpch_n_parts(n_orders), part, n_parts, begin, end); break;
      case LDB_TPCH_SUPPLIER: ldb_tpch_row_slice(ldb_tpch_n_suppliers(n_orders), part, n_parts, begin, end); break;
      case LDB_TPCH_PARTSUPP: ldb_tpch_row_slice(ldb_tpch_n_parts(n_orders) * 4, part, n_parts, begin, end); break;
      case LDB_TPCH_NATION: *begin = 0; *end = 25; break; // small tables are replicated on every rank
      default: *begin = 0; *end = 5; break;
   }
}

// narrow level 2: bytes per value of a generated column, from the value ranges the generator's definition guarantees (include/ldb_tpchgen.h) — what
// ldb_gpu_table_register finds by scanning an imported column's values
static int32_t gen_width(int32_t table, int32_t c, const ldb_coltype& t, int32_t narrow) {
   const int32_t w = ldb_width_of(t, narrow);
   if (narrow < 2) return w;
   if (t.type == LDB_T_CHAR4) return 1; // l_returnflag, l_linestatus, o_orderstatus: one ASCII letter
   if (t.type != LDB_T_DECIMAL128) return w;
   switch (table) {
      case LDB_TPCH_LINEITEM:
         if (c == L_DISCOUNT || c == L_TAX) return 1; // 0.00 … 0.10 / 0.08
         if (c == L_QUANTITY) return 2; // 1.00 … 50.00
         return 4; // l_extendedprice <= 104 949.50
      case LDB_TPCH_ORDERS: return 4; // o_totalprice: at most seven lines
      case LDB_TPCH_CUSTOMER:
      case LDB_TPCH_SUPPLIER: return 4; // -999.99 … 9 999.99
      case LDB_TPCH_PART: return 4; // p_retailprice <= 2 100.00
      case LDB_TPCH_PARTSUPP: return 4; // ps_supplycost <= 1 000.00
      default: return w;
   }
}

extern "C" int32_t ldb_gpu_tpch_generate(ldb_ctx* ctx, int32_t table_id, int64_t n_orders, int32_t part, int32_t n_parts, uint64_t col_mask,
                                         int32_t narrow, ldb_table** out) {
   if (!ctx || !out || n_orders < 1 || n_parts < 1 || part < 0 || part >= n_parts) LDB_FAIL(LDB_ERR_INVALID, "tpch_generate: bad argument");
   const ColDef* defs;
   int n_all;
   const char* tname;
   if (table_def(table_id, &defs, &n_all, &tname)) LDB_FAIL(LDB_ERR_INVALID, "tpch_generate: unknown table %d", table_id);
   if (col_mask == 0) col_mask = (1ull << n_all) - 1;
   int64_t b, e;
   table_slice(table_id, n_orders, part, n_parts, &b, &e);
   const int64_t n = e - b;
   if (n >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "tpch_generate: slice of %ld rows exceeds uint32 row ids; use more partitions", (long) n);
   auto t = std::make_unique<ldb_table>();
   t->ctx = ctx;
   t->name = tname;
   t->n_rows = n;
   GenCols g;
   memset(&g, 0, sizeof(g));
   const int grid = ldb_grid_for(ctx, n, 256, 8);
   for (int c = 0; c < n_all; c++) {
      if (!((col_mask >> c) & 1)) continue;
      ldb_column col;
      col.name = defs[c].name;
      col.type = defs[c].type;
      col.width = gen_width(table_id, c, col.type, narrow);
      if (col.type.type == LDB_T_UTF8 && table_id == LDB_TPCH_CUSTOMER && c == C_NAME) {
         col.value_bytes = n * LDB_TPCH_CNAME_LEN;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) col.value_bytes));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.offsets, 8 * (size_t) (n + 1)));
         hipLaunchKernelGGL(k_gen_cname, dim3(grid), dim3(256), 0, ctx->stream, b, (uint64_t) n, col.offsets, (char*) col.values);
      } else if (col.type.type == LDB_T_UTF8 && ldb_tpch_is_text(table_id, c)) { // p_brand, s_comment, o_comment, c_phone …
         int64_t* lens;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &lens, 8 * (size_t) (n + 1)));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.offsets, 8 * (size_t) (n + 1)));
         if (n) hipLaunchKernelGGL(k_gen_text_lens, dim3(grid), dim3(256), 0, ctx->stream, table_id, (int32_t) c, b, (uint64_t) n, lens);
         LDB_TRY(ldb_exclusive_scan_i64(ctx, lens, col.offsets, n, col.offsets + n));
         uint64_t total = 0;
         LDB_TRY(ldb_read_u64(ctx, col.offsets + n, &total));
         ldb_dev_free(ctx, lens);
         col.value_bytes = (int64_t) total;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) total));
         if (n) hipLaunchKernelGGL(k_gen_text_fill, dim3(grid), dim3(256), 0, ctx->stream, table_id, (int32_t) c, b, (uint64_t) n, (const int64_t*) col.offsets, (char*) col.values);
      } else if (col.type.type == LDB_T_UTF8 && ldb_tpch_wordcol_words(table_id, c)) { // p_name, p_type
         const int words = ldb_tpch_wordcol_words(table_id, c);
         const char* const* vocab = c == P_NAME ? ldb_tpch_colors : ldb_tpch_typewords;
         const int n_vocab = c == P_NAME ? LDB_TPCH_NCOLORS : LDB_TPCH_NTYPEWORDS;
         ColorDomain dom;
         memset(&dom, 0, sizeof(dom));
         int pos = 0;
         for (int k = 0; k < n_vocab; k++) {
            dom.off[k] = pos;
            size_t len = strlen(vocab[k]);
            if (pos + len > sizeof(dom.blob)) LDB_FAIL(LDB_ERR_INVALID, "tpch_generate: vocabulary exceeds its buffer");
            memcpy(dom.blob + pos, vocab[k], len);
            pos += (int) len;
         }
         dom.off[n_vocab] = pos;
         int64_t* lens;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &lens, 8 * (size_t) (n + 1)));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.offsets, 8 * (size_t) (n + 1)));
         if (n) hipLaunchKernelGGL(k_gen_pname_lens, dim3(grid), dim3(256), 0, ctx->stream, dom, (int32_t) c, (int32_t) words, b, (uint64_t) n, lens);
         LDB_TRY(ldb_exclusive_scan_i64(ctx, lens, col.offsets, n, col.offsets + n));
         uint64_t total = 0;
         LDB_TRY(ldb_read_u64(ctx, col.offsets + n, &total));
         ldb_dev_free(ctx, lens);
         col.value_bytes = (int64_t) total;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) total));
         if (n) hipLaunchKernelGGL(k_gen_pname_fill, dim3(grid), dim3(256), 0, ctx->stream, dom, (int32_t) c, (int32_t) words, b, (uint64_t) n, (const int64_t*) col.offsets, (char*) col.values);
      } else if (col.type.type == LDB_T_UTF8) {
         const char* const* strs = domain_strings(table_id, c);
         StrDomain dom;
         memset(&dom, 0, sizeof(dom));
         dom.n = ldb_tpch_str_domain(table_id, c);
         int pos = 0;
         for (int k = 0; k < dom.n; k++) {
            dom.off[k] = pos;
            size_t len = strlen(strs[k]);
            memcpy(dom.blob + pos, strs[k], len);
            pos += (int) len;
         }
         dom.off[dom.n] = pos;
         int64_t* lens;
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &lens, 8 * (size_t) (n + 1)));
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &col.offsets, 8 * (size_t) (n + 1)));
         if (n) hipLaunchKernelGGL(k_gen_str_lens, dim3(grid), dim3(256), 0, ctx->stream, dom, table_id, c, b, (uint64_t) n, lens);
         LDB_TRY(ldb_exclusive_scan_i64(ctx, lens, col.offsets, n, col.offsets + n));
         uint64_t total = 0;
         LDB_TRY(ldb_read_u64(ctx, col.offsets + n, &total));
         ldb_dev_free(ctx, lens);
         col.value_bytes = (int64_t) total;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) total));
         if (n) hipLaunchKernelGGL(k_gen_str_fill, dim3(grid), dim3(256), 0, ctx->stream, dom, table_id, c, b, (uint64_t) n, (const int64_t*) col.offsets, (char*) col.values);
      } else {
         col.value_bytes = n * col.width;
         LDB_TRY(ldb_dev_alloc(ctx, &col.values, (size_t) col.value_bytes));
         g.values[c] = col.values;
         g.width[c] = col.width;
         g.mask |= 1ull << c;
      }
      t->cols.push_back(col);
   }
   if (n && g.mask) {
      switch (table_id) {
         case LDB_TPCH_LINEITEM: hipLaunchKernelGGL(k_gen_lineitem, dim3(grid), dim3(256), 0, ctx->stream, g, n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_ORDERS: hipLaunchKernelGGL(k_gen_orders, dim3(grid), dim3(256), 0, ctx->stream, g, n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_CUSTOMER: hipLaunchKernelGGL(k_gen_customer, dim3(grid), dim3(256), 0, ctx->stream, g, n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_PART: hipLaunchKernelGGL(k_gen_part, dim3(grid), dim3(256), 0, ctx->stream, g, n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_SUPPLIER: hipLaunchKernelGGL(k_gen_supplier, dim3(grid), dim3(256), 0, ctx->stream, g, n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_PARTSUPP: hipLaunchKernelGGL(k_gen_partsupp, dim3(grid), dim3(256), 0, ctx->stream, g, n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_PROBEKEYS: hipLaunchKernelGGL(k_gen_probekeys, dim3(grid), dim3(256), 0, ctx->stream, (int32_t*) g.values[0], n_orders, b, (uint64_t) n); break;
         case LDB_TPCH_NATION: hipLaunchKernelGGL(k_gen_nation, dim3(grid), dim3(256), 0, ctx->stream, g, b, (uint64_t) n); break;
         default: hipLaunchKernelGGL(k_gen_region, dim3(grid), dim3(256), 0, ctx->stream, g, b, (uint64_t) n); break;
      }
   }
   LDB_HIP(hipGetLastError());
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   LDB_TRY(ldb_table_dict_encode_all(ctx, t.get())); // as ldb_gpu_table_register does for imported tables
   *out = t.release();
   return LDB_OK;
}
