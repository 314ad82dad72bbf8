/*
 * tpchgen_host.c — host twin of the device generator (csrc/ldb_tpchgen.hip): same data
 * definition (include/ldb_tpchgen.h), plain host buffers out.  Used to feed the CPU baseline,
 * the parity tests (device generator == host generator) and ldb_gpu_table_register demos.
 */
#include "../../include/ldb_tpchgen.h"
#include <string.h>

static void slice(int32_t table, int64_t n_orders, int32_t part, int32_t n_parts, int64_t* begin, int64_t* end) {
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
      case LDB_TPCH_NATION:
         *begin = 0;
         *end = 25;
         break;
      default:
         *begin = 0;
         *end = 5;
         break;
   }
}

int64_t ldb_tpch_host_rows(int32_t table_id, int64_t n_orders, int32_t part, int32_t n_parts) {
   int64_t b, e;
   slice(table_id, n_orders, part, n_parts, &b, &e);
   return e - b;
}

static const char* const* domain(int32_t table, int32_t col) {
   if (table == LDB_TPCH_LINEITEM && col == L_SHIPINSTRUCT) return ldb_tpch_instructs;
   if (table == LDB_TPCH_LINEITEM && col == L_SHIPMODE) return ldb_tpch_shipmodes;
   if (table == LDB_TPCH_ORDERS && col == O_ORDERPRIORITY) return ldb_tpch_priorities;
   if (table == LDB_TPCH_CUSTOMER && col == C_MKTSEGMENT) return ldb_tpch_segments;
   if (table == LDB_TPCH_NATION && col == N_NAME) return ldb_tpch_nations;
   if (table == LDB_TPCH_REGION && col == R_NAME) return ldb_tpch_regions;
   return 0;
}

static void put_dec(void* out, int64_t i, int64_t v) {
   int64_t* p = (int64_t*) out + 2 * i;
   p[0] = v;
   p[1] = v >> 63;
}

/* kind of a column: 0 = int32-like (4 bytes), 1 = decimal128 (16 bytes), 2 = utf8 */
static int col_kind(int32_t table, int32_t col) {
   if (ldb_tpch_str_domain(table, col) || ldb_tpch_is_text(table, col)) return 2;
   if (table == LDB_TPCH_CUSTOMER && col == C_NAME) return 2;
   if (ldb_tpch_wordcol_words(table, col)) return 2;
   switch (table) {
      case LDB_TPCH_LINEITEM: return (col >= L_QUANTITY && col <= L_TAX) ? 1 : 0;
      case LDB_TPCH_ORDERS: return col == O_TOTALPRICE ? 1 : 0;
      case LDB_TPCH_CUSTOMER: return col == C_ACCTBAL ? 1 : 0;
      case LDB_TPCH_PART: return col == P_RETAILPRICE ? 1 : 0;
      case LDB_TPCH_SUPPLIER: return col == S_ACCTBAL ? 1 : 0;
      case LDB_TPCH_PARTSUPP: return col == PS_SUPPLYCOST ? 1 : 0;
      default: return 0;
   }
}

int64_t ldb_tpch_host_column(int32_t table, int32_t col, int64_t n_orders, int32_t part, int32_t n_parts, void* out, int64_t* offsets_out,
                             int64_t* bytes) {
   int64_t b, e;
   slice(table, n_orders, part, n_parts, &b, &e);
   const int64_t n = e - b;
   const int kind = col_kind(table, col);
   if (kind == 2 && table == LDB_TPCH_CUSTOMER && col == C_NAME) {
      for (int64_t i = 0; i < n; i++) {
         if (offsets_out) offsets_out[i] = i * LDB_TPCH_CNAME_LEN;
         if (out) ldb_tpch_c_name(b + i + 1, (char*) out + i * LDB_TPCH_CNAME_LEN);
      }
      if (offsets_out) offsets_out[n] = n * LDB_TPCH_CNAME_LEN;
      if (bytes) *bytes = n * LDB_TPCH_CNAME_LEN;
      return n;
   }
   if (kind == 2 && ldb_tpch_is_text(table, col)) { /* p_brand, s_comment, o_comment, c_phone … */
      int64_t pos = 0;
      char buf[LDB_TPCH_TEXT_MAX];
      for (int64_t i = 0; i < n; i++) {
         const int32_t len = ldb_tpch_text(table, col, b + i, buf);
         if (offsets_out) offsets_out[i] = pos;
         if (out) memcpy((char*) out + pos, buf, (size_t) len);
         pos += len;
      }
      if (offsets_out) offsets_out[n] = pos;
      if (bytes) *bytes = pos;
      return n;
   }
   if (kind == 2 && ldb_tpch_wordcol_words(table, col)) { /* p_name, p_type */
      const int words = ldb_tpch_wordcol_words(table, col);
      const char* const* vocab = col == P_NAME ? ldb_tpch_colors : ldb_tpch_typewords;
      int64_t pos = 0;
      for (int64_t i = 0; i < n; i++) {
         if (offsets_out) offsets_out[i] = pos;
         for (int j = 0; j < words; j++) {
            const char* w = vocab[ldb_tpch_wordcol_word(col, b + i, j)];
            size_t len = strlen(w);
            if (j) {
               if (out) ((char*) out)[pos] = ' ';
               pos++;
            }
            if (out) memcpy((char*) out + pos, w, len);
            pos += (int64_t) len;
         }
      }
      if (offsets_out) offsets_out[n] = pos;
      if (bytes) *bytes = pos;
      return n;
   }
   if (kind == 2) {
      const char* const* dom = domain(table, col);
      int64_t pos = 0;
      for (int64_t i = 0; i < n; i++) {
         const char* s = dom[ldb_tpch_str_idx(table, col, b + i)];
         size_t len = strlen(s);
         if (offsets_out) offsets_out[i] = pos;
         if (out) memcpy((char*) out + pos, s, len);
         pos += (int64_t) len;
      }
      if (offsets_out) offsets_out[n] = pos;
      if (bytes) *bytes = pos;
      return n;
   }
   if (!out) {
      if (bytes) *bytes = n * (kind == 1 ? 16 : 4);
      return n;
   }
   int32_t* o32 = (int32_t*) out;
   for (int64_t i = 0; i < n; i++) {
      const int64_t r = b + i;
      switch (table) {
         case LDB_TPCH_LINEITEM: {
            ldb_tpch_lineitem li;
            ldb_tpch_lineitem_row(r, n_orders, &li);
            switch (col) {
               case L_ORDERKEY: o32[i] = li.orderkey; break;
               case L_PARTKEY: o32[i] = li.partkey; break;
               case L_SUPPKEY: o32[i] = li.suppkey; break;
               case L_LINENUMBER: o32[i] = li.linenumber; break;
               case L_QUANTITY: put_dec(out, i, li.quantity); break;
               case L_EXTENDEDPRICE: put_dec(out, i, li.extendedprice); break;
               case L_DISCOUNT: put_dec(out, i, li.discount); break;
               case L_TAX: put_dec(out, i, li.tax); break;
               case L_RETURNFLAG: o32[i] = li.returnflag; break;
               case L_LINESTATUS: o32[i] = li.linestatus; break;
               case L_SHIPDATE: o32[i] = li.shipdate; break;
               case L_COMMITDATE: o32[i] = li.commitdate; break;
               default: o32[i] = li.receiptdate; break;
            }
            break;
         }
         case LDB_TPCH_ORDERS: {
            switch (col) {
               case O_ORDERKEY: o32[i] = ldb_tpch_orderkey(r); break;
               case O_CUSTKEY: o32[i] = ldb_tpch_o_custkey(r, n_orders); break;
               case O_ORDERSTATUS:
               case O_TOTALPRICE: {
                  int32_t st;
                  int64_t tp;
                  ldb_tpch_order_derived(r, n_orders, &st, &tp);
                  if (col == O_ORDERSTATUS) o32[i] = st;
                  else put_dec(out, i, tp);
                  break;
               }
               case O_ORDERDATE: o32[i] = ldb_tpch_o_orderdate(r); break;
               default: o32[i] = 0; break;
            }
            break;
         }
         case LDB_TPCH_CUSTOMER:
            if (col == C_CUSTKEY) o32[i] = (int32_t) (r + 1);
            else if (col == C_NATIONKEY) o32[i] = ldb_tpch_c_nationkey(r);
            else put_dec(out, i, ldb_tpch_c_acctbal(r));
            break;
         case LDB_TPCH_PART:
            if (col == P_PARTKEY) o32[i] = (int32_t) (r + 1);
            else if (col == P_SIZE) o32[i] = (int32_t) ldb_uniform(LDB_TPCH_PART, P_SIZE, (uint64_t) r, 1, 50);
            else put_dec(out, i, ldb_tpch_retailprice(r + 1));
            break;
         case LDB_TPCH_SUPPLIER:
            if (col == S_SUPPKEY) o32[i] = (int32_t) (r + 1);
            else if (col == S_NATIONKEY) o32[i] = (int32_t) (ldb_rnd(LDB_TPCH_SUPPLIER, S_NATIONKEY, (uint64_t) r) % 25);
            else put_dec(out, i, ldb_uniform(LDB_TPCH_SUPPLIER, S_ACCTBAL, (uint64_t) r, -99999, 999999));
            break;
         case LDB_TPCH_PARTSUPP: {
            int64_t pk = r / 4 + 1;
            if (col == PS_PARTKEY) o32[i] = (int32_t) pk;
            else if (col == PS_SUPPKEY) o32[i] = ldb_tpch_ps_suppkey(pk, (int32_t) (r % 4), n_orders);
            else if (col == PS_AVAILQTY) o32[i] = (int32_t) ldb_uniform(LDB_TPCH_PARTSUPP, PS_AVAILQTY, (uint64_t) r, 1, 9999);
            else put_dec(out, i, ldb_uniform(LDB_TPCH_PARTSUPP, PS_SUPPLYCOST, (uint64_t) r, 100, 100000));
            break;
         }
         case LDB_TPCH_PROBEKEYS: o32[i] = ldb_tpch_orderkey((int64_t) (ldb_rnd(LDB_TPCH_PROBEKEYS, 0, (uint64_t) r) % (uint64_t) n_orders)); break;
         case LDB_TPCH_NATION:
            if (col == N_NATIONKEY) o32[i] = (int32_t) r;
            else o32[i] = ldb_tpch_nation_region[r % 25];
            break;
         default: o32[i] = (int32_t) r; break;
      }
   }
   if (bytes) *bytes = n * (kind == 1 ? 16 : 4);
   return n;
}
