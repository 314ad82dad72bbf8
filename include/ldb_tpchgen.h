/*
 * ldb_tpchgen.h — deterministic, counter-based, TPC-H-*shaped* synthetic data.
 *
 * dbgen itself is not available offline (reference tools/generate/tpch.sh:6 downloads it), so
 * the benchmark and parity data come from this generator: every value is a pure function of
 * (seed, table, column, row), so the CPU (host generator), every GPU and every rank produce
 * identical data without communication.  Schema, physical types and value domains follow
 * resources/sql/tpch/initialize.sql and the TPC-H spec's distributions (uniform key domains,
 * 1..7 lines per order, date windows, flag rules), NOT dbgen's text grammar — golden answers
 * of test/sqlite-datasets/tpchSf1.test therefore do not apply.
 *
 * This header is the single definition of the data: LDB_HD functions are compiled both into
 * the device generator (csrc/ldb_tpchgen.hip) and the host generator (host/tpchgen_host.c).
 */
#ifndef LDB_TPCHGEN_H
#define LDB_TPCHGEN_H
#include <stdint.h>

#if defined(__HIPCC__)
#define LDB_HD __host__ __device__ static inline
#else
#define LDB_HD static inline
#endif

#define LDB_TPCH_SEED 20260925ULL

/* tables */
enum { LDB_TPCH_LINEITEM = 0,
       LDB_TPCH_ORDERS = 1,
       LDB_TPCH_CUSTOMER = 2,
       LDB_TPCH_PART = 3,
       LDB_TPCH_SUPPLIER = 4,
       LDB_TPCH_PARTSUPP = 5,
       LDB_TPCH_NATION = 6,
       LDB_TPCH_REGION = 7,
       /* bench support, not a TPC-H table: one int32 column `k_orderkey` with as many rows as lineitem,
        * each the key of a uniformly random order — an UNCLUSTERED foreign-key probe side with a 100 %
        * match rate (the case radix-partitioned joins exist for; l_orderkey itself is clustered) */
       LDB_TPCH_PROBEKEYS = 8 };

/* lineitem columns (index = column id in the generated table when all columns are requested) */
enum { L_ORDERKEY = 0,
       L_PARTKEY,
       L_SUPPKEY,
       L_LINENUMBER,
       L_QUANTITY,
       L_EXTENDEDPRICE,
       L_DISCOUNT,
       L_TAX,
       L_RETURNFLAG,
       L_LINESTATUS,
       L_SHIPDATE,
       L_COMMITDATE,
       L_RECEIPTDATE,
       L_SHIPINSTRUCT,
       L_SHIPMODE,
       L_NCOLS };
enum { O_ORDERKEY = 0,
       O_CUSTKEY,
       O_ORDERSTATUS,
       O_TOTALPRICE,
       O_ORDERDATE,
       O_ORDERPRIORITY,
       O_SHIPPRIORITY,
       O_COMMENT, /* pseudo text (vocabulary below); '%special%requests%' occurs naturally (Q13) */
       O_NCOLS };
enum { C_CUSTKEY = 0,
       C_NATIONKEY,
       C_ACCTBAL,
       C_MKTSEGMENT,
       C_NAME, /* "Customer#%09d" of the customer key: fixed 18 bytes (TPC-H spec 4.2.3) */
       C_PHONE, /* "CC-ddd-ddd-dddd", CC = c_nationkey + 10 (spec 4.2.2.9) */
       C_NCOLS };
#define LDB_TPCH_CNAME_LEN 18
/* writes the 18 bytes of c_name for customer key `custkey` */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline void ldb_tpch_c_name(int64_t custkey, char* out) {
   const char pre[9] = {'C', 'u', 's', 't', 'o', 'm', 'e', 'r', '#'};
   for (int i = 0; i < 9; i++) out[i] = pre[i];
   for (int i = 8; i >= 0; i--) {
      out[9 + i] = (char) ('0' + custkey % 10);
      custkey /= 10;
   }
}
enum { P_PARTKEY = 0,
       P_SIZE,
       P_RETAILPRICE,
       P_NAME, /* five colour words separated by blanks (TPC-H spec 4.2.3: P_NAME) */
       P_TYPE, /* three syllables, one from each of the TYPES lists (spec 4.2.2.13) */
       P_BRAND, /* "Brand#MN", M = manufacturer 1..5, N = 1..5 */
       P_CONTAINER, /* two syllables: {SM,LG,MED,JUMBO,WRAP} x {CASE,BOX,BAG,JAR,PKG,PACK,CAN,DRUM} */
       P_MFGR, /* "Manufacturer#M" */
       P_NCOLS };
enum { S_SUPPKEY = 0,
       S_NATIONKEY,
       S_ACCTBAL,
       S_NAME, /* "Supplier#%09d" */
       S_ADDRESS, /* 10..25 random letters */
       S_PHONE,
       S_COMMENT, /* pseudo text; one supplier in 200 carries "Customer ... Complaints" (Q16) */
       S_NCOLS };
enum { PS_PARTKEY = 0,
       PS_SUPPKEY,
       PS_AVAILQTY,
       PS_SUPPLYCOST,
       PS_NCOLS };
enum { N_NATIONKEY = 0,
       N_REGIONKEY,
       N_NAME,
       N_NCOLS };
enum { R_REGIONKEY = 0,
       R_NAME,
       R_NCOLS };

/* date32 constants (days since 1970-01-01) */
#define LDB_D_1992_01_01 8035
#define LDB_D_1998_08_02 10440
#define LDB_D_1995_06_17 9298

LDB_HD uint64_t ldb_mix64(uint64_t z) {
   z += 0x9E3779B97F4A7C15ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   return z ^ (z >> 31);
}
/* random 64-bit word for (table, stream, row) */
LDB_HD uint64_t ldb_rnd(uint32_t table, uint32_t stream, uint64_t row) {
   uint64_t k = ldb_mix64(LDB_TPCH_SEED ^ ((uint64_t) table << 56) ^ ((uint64_t) stream << 40));
   return ldb_mix64(k ^ (row * 0xD1342543DE82EF95ULL));
}
LDB_HD int64_t ldb_uniform(uint32_t table, uint32_t stream, uint64_t row, int64_t lo, int64_t hi) {
   return lo + (int64_t) (ldb_rnd(table, stream, row) % (uint64_t) (hi - lo + 1));
}

/* ---- cardinalities, all derived from the total number of orders (1 500 000 x SF) */
LDB_HD int64_t ldb_tpch_n_customers(int64_t n_orders) {
   int64_t n = n_orders / 10;
   return n < 3 ? 3 : n;
}
LDB_HD int64_t ldb_tpch_n_parts(int64_t n_orders) {
   int64_t n = n_orders * 2 / 15;
   return n < 1 ? 1 : n;
}
LDB_HD int64_t ldb_tpch_n_suppliers(int64_t n_orders) {
   int64_t n = n_orders / 150;
   return n < 4 ? 4 : n;
}

/* lines per order: fixed period-7 pattern (mean 4 → 6 000 000 x SF lineitems), so the
 * lineitem row offset of an order is closed-form (no scan needed on any device). */
LDB_HD int32_t ldb_tpch_lines_of(int64_t order_idx) {
   const int32_t pat[7] = {4, 1, 7, 3, 6, 2, 5};
   return pat[order_idx % 7];
}
LDB_HD int64_t ldb_tpch_line_offset(int64_t order_idx) { /* first lineitem row of order order_idx */
   const int32_t pre[8] = {0, 4, 5, 12, 15, 21, 23, 28};
   return (order_idx / 7) * 28 + pre[order_idx % 7];
}
LDB_HD int64_t ldb_tpch_n_lineitems(int64_t n_orders) {
   return ldb_tpch_line_offset(n_orders);
}
/* lineitem row → (order index, 1-based line number) */
LDB_HD void ldb_tpch_row_to_order(int64_t row, int64_t* order_idx, int32_t* line_no) {
   const int32_t pre[8] = {0, 4, 5, 12, 15, 21, 23, 28};
   int64_t blk = row / 28;
   int32_t r = (int32_t) (row % 28);
   int32_t k = 0;
   while (pre[k + 1] <= r) k++;
   *order_idx = blk * 7 + k;
   *line_no = r - pre[k] + 1;
}

/* ---- orders */
LDB_HD int32_t ldb_tpch_orderkey(int64_t order_idx) { /* dbgen-style sparse keys */
   int64_t i = order_idx + 1;
   return (int32_t) (((i >> 3) << 5) | (i & 7));
}
LDB_HD int32_t ldb_tpch_o_custkey(int64_t order_idx, int64_t n_orders) {
   int64_t nc = ldb_tpch_n_customers(n_orders);
   int64_t usable = (nc / 3) * 2; /* keys with key % 3 != 0 among 1..3*(nc/3) */
   if (usable < 1) usable = 1;
   int64_t r = (int64_t) (ldb_rnd(LDB_TPCH_ORDERS, O_CUSTKEY, (uint64_t) order_idx) % (uint64_t) usable);
   return (int32_t) ((r / 2) * 3 + (r % 2) + 1);
}
LDB_HD int32_t ldb_tpch_o_orderdate(int64_t order_idx) {
   return (int32_t) ldb_uniform(LDB_TPCH_ORDERS, O_ORDERDATE, (uint64_t) order_idx, LDB_D_1992_01_01, LDB_D_1998_08_02);
}
LDB_HD int32_t ldb_tpch_o_priority_idx(int64_t order_idx) {
   return (int32_t) (ldb_rnd(LDB_TPCH_ORDERS, O_ORDERPRIORITY, (uint64_t) order_idx) % 5);
}

/* ---- part / supplier */
LDB_HD int64_t ldb_tpch_retailprice(int64_t partkey) { /* cents */
   return 90000 + ((partkey / 10) % 20001) + 100 * (partkey % 1000);
}
LDB_HD int32_t ldb_tpch_ps_suppkey(int64_t partkey, int32_t j, int64_t n_orders) { /* j = 0..3 */
   int64_t s = ldb_tpch_n_suppliers(n_orders);
   return (int32_t) ((partkey + j * (s / 4 + (partkey - 1) / s)) % s + 1);
}

/* ---- lineitem (row = global lineitem row) */
typedef struct {
   int32_t orderkey, partkey, suppkey, linenumber;
   int64_t quantity, extendedprice, discount, tax; /* unscaled decimal(12,2) */
   int32_t returnflag, linestatus; /* char(1) as the 4 raw bytes of fixed_size_binary(4) */
   int32_t shipdate, commitdate, receiptdate;
   int32_t shipinstruct_idx, shipmode_idx;
} ldb_tpch_lineitem;

LDB_HD void ldb_tpch_lineitem_row(int64_t row, int64_t n_orders, ldb_tpch_lineitem* out) {
   int64_t oi;
   int32_t ln;
   ldb_tpch_row_to_order(row, &oi, &ln);
   out->orderkey = ldb_tpch_orderkey(oi);
   out->linenumber = ln;
   int64_t np = ldb_tpch_n_parts(n_orders);
   int64_t pk = ldb_uniform(LDB_TPCH_LINEITEM, L_PARTKEY, (uint64_t) row, 1, np);
   out->partkey = (int32_t) pk;
   out->suppkey = ldb_tpch_ps_suppkey(pk, (int32_t) (ldb_rnd(LDB_TPCH_LINEITEM, L_SUPPKEY, (uint64_t) row) & 3), n_orders);
   int64_t qty = ldb_uniform(LDB_TPCH_LINEITEM, L_QUANTITY, (uint64_t) row, 1, 50);
   out->quantity = qty * 100;
   out->extendedprice = qty * ldb_tpch_retailprice(pk);
   out->discount = ldb_uniform(LDB_TPCH_LINEITEM, L_DISCOUNT, (uint64_t) row, 0, 10);
   out->tax = ldb_uniform(LDB_TPCH_LINEITEM, L_TAX, (uint64_t) row, 0, 8);
   int32_t od = ldb_tpch_o_orderdate(oi);
   out->shipdate = od + (int32_t) ldb_uniform(LDB_TPCH_LINEITEM, L_SHIPDATE, (uint64_t) row, 1, 121);
   out->commitdate = od + (int32_t) ldb_uniform(LDB_TPCH_LINEITEM, L_COMMITDATE, (uint64_t) row, 30, 90);
   out->receiptdate = out->shipdate + (int32_t) ldb_uniform(LDB_TPCH_LINEITEM, L_RECEIPTDATE, (uint64_t) row, 1, 30);
   if (out->receiptdate <= LDB_D_1995_06_17) {
      out->returnflag = (ldb_rnd(LDB_TPCH_LINEITEM, L_RETURNFLAG, (uint64_t) row) & 1) ? 'R' : 'A';
   } else {
      out->returnflag = 'N';
   }
   out->linestatus = out->shipdate > LDB_D_1995_06_17 ? 'O' : 'F';
   out->shipinstruct_idx = (int32_t) (ldb_rnd(LDB_TPCH_LINEITEM, L_SHIPINSTRUCT, (uint64_t) row) & 3);
   out->shipmode_idx = (int32_t) (ldb_rnd(LDB_TPCH_LINEITEM, L_SHIPMODE, (uint64_t) row) % 7);
}

/* o_orderstatus / o_totalprice are functions of the order's lines */
LDB_HD void ldb_tpch_order_derived(int64_t order_idx, int64_t n_orders, int32_t* status, int64_t* totalprice) {
   int64_t base = ldb_tpch_line_offset(order_idx);
   int32_t n = ldb_tpch_lines_of(order_idx);
   int32_t nf = 0;
   int64_t tot = 0;
   for (int32_t l = 0; l < n; l++) {
      ldb_tpch_lineitem li;
      ldb_tpch_lineitem_row(base + l, n_orders, &li);
      nf += li.linestatus == 'F';
      tot += li.extendedprice * (100 - li.discount) * (100 + li.tax) / 10000;
   }
   *status = nf == n ? 'F' : (nf == 0 ? 'O' : 'P');
   *totalprice = tot;
}

/* small string domains (utf8 columns) */
#define LDB_TPCH_NSEG 5
#define LDB_TPCH_NPRIO 5
#define LDB_TPCH_NMODE 7
#define LDB_TPCH_NINSTR 4
static const char* const ldb_tpch_segments[LDB_TPCH_NSEG] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"};
static const char* const ldb_tpch_priorities[LDB_TPCH_NPRIO] = {"1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"};
static const char* const ldb_tpch_shipmodes[LDB_TPCH_NMODE] = {"REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"};
static const char* const ldb_tpch_instructs[LDB_TPCH_NINSTR] = {"DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"};
static const char* const ldb_tpch_nations[25] = {"ALGERIA", "ARGENTINA", "BRAZIL", "CANADA", "EGYPT", "ETHIOPIA", "FRANCE", "GERMANY", "INDIA", "INDONESIA", "IRAN", "IRAQ", "JAPAN", "JORDAN", "KENYA", "MOROCCO", "MOZAMBIQUE", "PERU", "CHINA", "ROMANIA", "SAUDI ARABIA", "VIETNAM", "RUSSIA", "UNITED KINGDOM", "UNITED STATES"};
static const int32_t ldb_tpch_nation_region[25] = {0, 1, 1, 1, 4, 0, 3, 3, 2, 2, 4, 4, 2, 4, 0, 0, 0, 1, 2, 3, 4, 2, 3, 3, 1};
static const char* const ldb_tpch_regions[5] = {"AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"};

/* p_name vocabulary: the 92 colour words of the TPC-H specification (clause 4.2.3, P_NAME) */
#define LDB_TPCH_NCOLORS 92
#define LDB_TPCH_PNAME_WORDS 5
static const char* const ldb_tpch_colors[LDB_TPCH_NCOLORS] = {
   "almond", "antique", "aquamarine", "azure", "beige", "bisque", "black", "blanched", "blue", "blush", "brown", "burlywood", "burnished", "chartreuse", "chiffon",
   "chocolate", "coral", "cornflower", "cornsilk", "cream", "cyan", "dark", "deep", "dim", "dodger", "drab", "firebrick", "floral", "forest", "frosted", "gainsboro",
   "ghost", "goldenrod", "green", "grey", "honeydew", "hot", "indian", "ivory", "khaki", "lace", "lavender", "lawn", "lemon", "light", "lime", "linen", "magenta",
   "maroon", "medium", "metallic", "midnight", "mint", "misty", "moccasin", "navajo", "navy", "olive", "orange", "orchid", "pale", "papaya", "peach", "peru", "pink",
   "plum", "powder", "puff", "purple", "red", "rose", "rosy", "royal", "saddle", "salmon", "sandy", "seashell", "sienna", "sky", "slate", "smoke", "snow", "spring",
   "steel", "tan", "thistle", "tomato", "turquoise", "violet", "wheat", "white", "yellow"};
/* colour index of word j (0..4) of the name of part row `part_idx` (independent draws: a colour may repeat) */
LDB_HD int32_t ldb_tpch_p_name_word(int64_t part_idx, int32_t j) {
   return (int32_t) (ldb_rnd(LDB_TPCH_PART, P_NAME * 8 + j, (uint64_t) part_idx) % LDB_TPCH_NCOLORS);
}

/* p_type vocabulary: syllable 1 (6 words), syllable 2 (5), syllable 3 (5) in one list */
#define LDB_TPCH_NTYPEWORDS 16
#define LDB_TPCH_PTYPE_WORDS 3
static const char* const ldb_tpch_typewords[LDB_TPCH_NTYPEWORDS] = {"STANDARD", "SMALL", "MEDIUM", "LARGE", "ECONOMY", "PROMO", "ANODIZED", "BURNISHED", "PLATED", "POLISHED",
                                                                    "BRUSHED", "TIN", "NICKEL", "BRASS", "STEEL", "COPPER"};
LDB_HD int32_t ldb_tpch_p_type_word(int64_t part_idx, int32_t j) {
   const uint64_t r = ldb_rnd(LDB_TPCH_PART, P_TYPE * 8 + j, (uint64_t) part_idx);
   return j == 0 ? (int32_t) (r % 6) : (j == 1 ? 6 + (int32_t) (r % 5) : 11 + (int32_t) (r % 5));
}
/* "word columns" (utf8 values made of blank-separated vocabulary words): words per value (0 = not
 * a word column) and the vocabulary index of word j */
LDB_HD int32_t ldb_tpch_wordcol_words(int32_t table, int32_t col) {
   if (table != LDB_TPCH_PART) return 0;
   return col == P_NAME ? LDB_TPCH_PNAME_WORDS : (col == P_TYPE ? LDB_TPCH_PTYPE_WORDS : 0);
}
LDB_HD int32_t ldb_tpch_wordcol_word(int32_t col, int64_t row, int32_t j) { return col == P_NAME ? ldb_tpch_p_name_word(row, j) : ldb_tpch_p_type_word(row, j); }

LDB_HD int32_t ldb_tpch_c_segment_idx(int64_t cust_idx) {
   return (int32_t) (ldb_rnd(LDB_TPCH_CUSTOMER, C_MKTSEGMENT, (uint64_t) cust_idx) % LDB_TPCH_NSEG);
}
LDB_HD int32_t ldb_tpch_c_nationkey(int64_t cust_idx) {
   return (int32_t) (ldb_rnd(LDB_TPCH_CUSTOMER, C_NATIONKEY, (uint64_t) cust_idx) % 25);
}
LDB_HD int64_t ldb_tpch_c_acctbal(int64_t cust_idx) {
   return ldb_uniform(LDB_TPCH_CUSTOMER, C_ACCTBAL, (uint64_t) cust_idx, -99999, 999999);
}

/* index into the column's string domain for utf8 columns */
LDB_HD int32_t ldb_tpch_str_idx(int32_t table, int32_t col, int64_t row) {
   switch (table) {
      case LDB_TPCH_LINEITEM:
         if (col == L_SHIPINSTRUCT) return (int32_t) (ldb_rnd(LDB_TPCH_LINEITEM, L_SHIPINSTRUCT, (uint64_t) row) & 3);
         return (int32_t) (ldb_rnd(LDB_TPCH_LINEITEM, L_SHIPMODE, (uint64_t) row) % 7);
      case LDB_TPCH_ORDERS: return ldb_tpch_o_priority_idx(row);
      case LDB_TPCH_CUSTOMER: return ldb_tpch_c_segment_idx(row);
      default: return (int32_t) row; /* nation / region names */
   }
}
/* number of strings in the domain of a utf8 column (0 = not a string column) */
LDB_HD int32_t ldb_tpch_str_domain(int32_t table, int32_t col) {
   if (table == LDB_TPCH_LINEITEM && col == L_SHIPINSTRUCT) return LDB_TPCH_NINSTR;
   if (table == LDB_TPCH_LINEITEM && col == L_SHIPMODE) return LDB_TPCH_NMODE;
   if (table == LDB_TPCH_ORDERS && col == O_ORDERPRIORITY) return LDB_TPCH_NPRIO;
   if (table == LDB_TPCH_CUSTOMER && col == C_MKTSEGMENT) return LDB_TPCH_NSEG;
   if (table == LDB_TPCH_NATION && col == N_NAME) return 25;
   if (table == LDB_TPCH_REGION && col == R_NAME) return 5;
   return 0;
}

/* ---- generated text columns (p_brand, p_container, p_mfgr, s_name, s_address, s_phone, s_comment,
 * o_comment, c_phone): ldb_tpch_text writes the value of (table, col, row) to `out` (capacity
 * LDB_TPCH_TEXT_MAX) and returns its length; ldb_tpch_is_text tells whether a column is one. */
#define LDB_TPCH_TEXT_MAX 104
LDB_HD int32_t ldb_tpch_is_text(int32_t table, int32_t col) {
   if (table == LDB_TPCH_PART) return col == P_BRAND || col == P_CONTAINER || col == P_MFGR;
   if (table == LDB_TPCH_SUPPLIER) return col == S_NAME || col == S_ADDRESS || col == S_PHONE || col == S_COMMENT;
   if (table == LDB_TPCH_ORDERS) return col == O_COMMENT;
   if (table == LDB_TPCH_CUSTOMER) return col == C_PHONE;
   return 0;
}
LDB_HD int32_t ldb_tpch_put(char* out, int32_t pos, const char* s) {
   while (*s) out[pos++] = *s++;
   return pos;
}
LDB_HD int32_t ldb_tpch_put_num(char* out, int32_t pos, int64_t v, int32_t digits) { /* zero padded */
   for (int32_t i = digits - 1; i >= 0; i--) {
      out[pos + i] = (char) ('0' + v % 10);
      v /= 10;
   }
   return pos + digits;
}
LDB_HD int32_t ldb_tpch_phone(char* out, int32_t nationkey, uint64_t r) {
   int32_t p = ldb_tpch_put_num(out, 0, nationkey + 10, 2);
   out[p++] = '-';
   p = ldb_tpch_put_num(out, p, (int64_t) (100 + r % 900), 3);
   out[p++] = '-';
   p = ldb_tpch_put_num(out, p, (int64_t) (100 + (r >> 16) % 900), 3);
   out[p++] = '-';
   return ldb_tpch_put_num(out, p, (int64_t) (1000 + (r >> 32) % 9000), 4);
}
/* word k (0..31) of the comment vocabulary appended at out[pos] */
LDB_HD int32_t ldb_tpch_put_word(char* out, int32_t pos, int32_t k) {
   const char blob[] = "furiously\0quickly\0carefully\0blithely\0slyly\0regular\0final\0ironic\0even\0bold\0silent\0pending\0express\0unusual\0special\0requests\0"
                       "deposits\0packages\0accounts\0instructions\0foxes\0ideas\0theodolites\0pinto\0beans\0platelets\0asymptotes\0courts\0dolphins\0excuses\0sleep\0wake";
   int32_t at = 0;
   for (int32_t w = 0; w < (k & 31); w++) {
      while (blob[at]) at++;
      at++;
   }
   return ldb_tpch_put(out, pos, blob + at);
}
/* blank-separated vocabulary words up to `cap` bytes; `inject` != 0 puts "Customer" … "Complaints" in */
LDB_HD int32_t ldb_tpch_comment(char* out, int32_t table, int32_t col, int64_t row, int32_t cap, int32_t inject) {
   const uint64_t r0 = ldb_rnd((uint32_t) table, (uint32_t) col * 8u, (uint64_t) row);
   const int32_t n_words = 4 + (int32_t) (r0 % 6); /* 4..9 words */
   int32_t pos = 0;
   for (int32_t j = 0; j < n_words; j++) {
      const uint64_t r = ldb_rnd((uint32_t) table, (uint32_t) col * 8u + 1u, (uint64_t) row * 16u + (uint64_t) j);
      if (pos + 14 > cap) break; /* the longest word has 12 bytes */
      if (j) out[pos++] = ' ';
      if (inject && j == 1) pos = ldb_tpch_put(out, pos, "Customer");
      else if (inject && j == 3) pos = ldb_tpch_put(out, pos, "Complaints");
      else pos = ldb_tpch_put_word(out, pos, (int32_t) (r & 31));
   }
   return pos;
}
LDB_HD int32_t ldb_tpch_s_nationkey(int64_t supp_idx) { return (int32_t) (ldb_rnd(LDB_TPCH_SUPPLIER, S_NATIONKEY, (uint64_t) supp_idx) % 25); }
LDB_HD int32_t ldb_tpch_p_mfgr(int64_t part_idx) { return 1 + (int32_t) (ldb_rnd(LDB_TPCH_PART, P_MFGR, (uint64_t) part_idx) % 5); }
LDB_HD int32_t ldb_tpch_text(int32_t table, int32_t col, int64_t row, char* out) {
   int32_t p = 0;
   if (table == LDB_TPCH_PART) {
      if (col == P_BRAND) {
         p = ldb_tpch_put(out, 0, "Brand#");
         out[p++] = (char) ('0' + ldb_tpch_p_mfgr(row));
         out[p++] = (char) ('1' + ldb_rnd(LDB_TPCH_PART, P_BRAND, (uint64_t) row) % 5);
         return p;
      }
      if (col == P_MFGR) {
         p = ldb_tpch_put(out, 0, "Manufacturer#");
         out[p++] = (char) ('0' + ldb_tpch_p_mfgr(row));
         return p;
      }
      /* P_CONTAINER */
      const uint64_t r = ldb_rnd(LDB_TPCH_PART, P_CONTAINER, (uint64_t) row);
      switch (r % 5) {
         case 0: p = ldb_tpch_put(out, 0, "SM "); break;
         case 1: p = ldb_tpch_put(out, 0, "LG "); break;
         case 2: p = ldb_tpch_put(out, 0, "MED "); break;
         case 3: p = ldb_tpch_put(out, 0, "JUMBO "); break;
         default: p = ldb_tpch_put(out, 0, "WRAP "); break;
      }
      switch ((r >> 8) % 8) {
         case 0: return ldb_tpch_put(out, p, "CASE");
         case 1: return ldb_tpch_put(out, p, "BOX");
         case 2: return ldb_tpch_put(out, p, "BAG");
         case 3: return ldb_tpch_put(out, p, "JAR");
         case 4: return ldb_tpch_put(out, p, "PKG");
         case 5: return ldb_tpch_put(out, p, "PACK");
         case 6: return ldb_tpch_put(out, p, "CAN");
         default: return ldb_tpch_put(out, p, "DRUM");
      }
   }
   if (table == LDB_TPCH_SUPPLIER) {
      if (col == S_NAME) {
         p = ldb_tpch_put(out, 0, "Supplier#");
         return ldb_tpch_put_num(out, p, row + 1, 9);
      }
      if (col == S_PHONE) return ldb_tpch_phone(out, ldb_tpch_s_nationkey(row), ldb_rnd(LDB_TPCH_SUPPLIER, S_PHONE, (uint64_t) row));
      if (col == S_ADDRESS) {
         const uint64_t r0 = ldb_rnd(LDB_TPCH_SUPPLIER, S_ADDRESS, (uint64_t) row);
         const int32_t len = 10 + (int32_t) (r0 % 16);
         for (int32_t i = 0; i < len; i++) {
            const uint64_t r = ldb_rnd(LDB_TPCH_SUPPLIER, S_ADDRESS * 8 + 1, (uint64_t) row * 32u + (uint64_t) i);
            out[i] = (char) ((r % 2 ? 'a' : 'A') + (r >> 8) % 26);
         }
         return len;
      }
      /* S_COMMENT */
      return ldb_tpch_comment(out, table, col, row, 101, ldb_rnd(LDB_TPCH_SUPPLIER, S_COMMENT * 8 + 2, (uint64_t) row) % 200 == 0);
   }
   if (table == LDB_TPCH_ORDERS) return ldb_tpch_comment(out, table, col, row, 79, 0);
   /* customer.c_phone */
   return ldb_tpch_phone(out, ldb_tpch_c_nationkey(row), ldb_rnd(LDB_TPCH_CUSTOMER, C_PHONE, (uint64_t) row));
}

/* ---- slices: rank `part` of `n_parts` owns a contiguous block of each table.  Orders (and
 * their lineitems) are split on 7-order boundaries so lineitem slices are closed-form. */
LDB_HD void ldb_tpch_order_slice(int64_t n_orders, int32_t part, int32_t n_parts, int64_t* begin, int64_t* end) {
   int64_t blocks = (n_orders + 6) / 7;
   int64_t b0 = blocks * part / n_parts, b1 = blocks * (part + 1) / n_parts;
   *begin = b0 * 7 < n_orders ? b0 * 7 : n_orders;
   *end = b1 * 7 < n_orders ? b1 * 7 : n_orders;
}
LDB_HD void ldb_tpch_row_slice(int64_t n_rows, int32_t part, int32_t n_parts, int64_t* begin, int64_t* end) {
   *begin = n_rows * part / n_parts;
   *end = n_rows * (part + 1) / n_parts;
}

#ifdef __cplusplus
extern "C" {
#endif
struct ldb_ctx;
struct ldb_table;
/* Device generator (liblingodb_gpu.so): generates slice `part`/`n_parts` of `table_id` for a
 * database of `n_orders` orders straight into HBM.  col_mask bit i = generate column i
 * (0 = all columns).  Benchmark/test support — not part of the operator path. */
int32_t ldb_gpu_tpch_generate(struct ldb_ctx* ctx, int32_t table_id, int64_t n_orders, int32_t part, int32_t n_parts,
                              uint64_t col_mask, int32_t narrow_decimals, struct ldb_table** out);
/* Host generator (libldb_host.so): fills `out` with column `col` of the same slice.
 * Fixed-width: out = values (decimals as 16-byte little-endian).  utf8: out = bytes,
 * offsets_out = int64[n+1].  Returns the number of rows (or bytes written for utf8 in *bytes). */
int64_t ldb_tpch_host_rows(int32_t table_id, int64_t n_orders, int32_t part, int32_t n_parts);
int64_t ldb_tpch_host_column(int32_t table_id, int32_t col, int64_t n_orders, int32_t part, int32_t n_parts, void* out,
                             int64_t* offsets_out, int64_t* bytes);
#ifdef __cplusplus
}
#endif
#endif
