"""pyarrow tables of the synthetic TPC-H-shaped database from the HOST generator
(lingo-db_amd/host/tpchgen_host.c — same definition as the device generator, include/ldb_tpchgen.h),
typed with the reference's physical Arrow types (LingoDBTable.cpp:122-195)."""
import ctypes as C

import numpy as np
import pyarrow as pa

from lingodb_amd import capi

LINEITEM, ORDERS, CUSTOMER, PART, SUPPLIER, PARTSUPP, NATION, REGION, PROBEKEYS = range(9)
DEC = pa.decimal128(12, 2)
CH = pa.binary(4)
SCHEMAS = {
    LINEITEM: [("l_orderkey", pa.int32()), ("l_partkey", pa.int32()), ("l_suppkey", pa.int32()), ("l_linenumber", pa.int32()),
               ("l_quantity", DEC), ("l_extendedprice", DEC), ("l_discount", DEC), ("l_tax", DEC),
               ("l_returnflag", CH), ("l_linestatus", CH), ("l_shipdate", pa.date32()), ("l_commitdate", pa.date32()),
               ("l_receiptdate", pa.date32()), ("l_shipinstruct", pa.string()), ("l_shipmode", pa.string())],
    ORDERS: [("o_orderkey", pa.int32()), ("o_custkey", pa.int32()), ("o_orderstatus", CH), ("o_totalprice", DEC),
             ("o_orderdate", pa.date32()), ("o_orderpriority", pa.string()), ("o_shippriority", pa.int32()), ("o_comment", pa.string())],
    CUSTOMER: [("c_custkey", pa.int32()), ("c_nationkey", pa.int32()), ("c_acctbal", DEC), ("c_mktsegment", pa.string()), ("c_name", pa.string()), ("c_phone", pa.string())],
    PART: [("p_partkey", pa.int32()), ("p_size", pa.int32()), ("p_retailprice", DEC), ("p_name", pa.string()), ("p_type", pa.string()),
           ("p_brand", pa.string()), ("p_container", pa.string()), ("p_mfgr", pa.string())],
    SUPPLIER: [("s_suppkey", pa.int32()), ("s_nationkey", pa.int32()), ("s_acctbal", DEC), ("s_name", pa.string()), ("s_address", pa.string()),
               ("s_phone", pa.string()), ("s_comment", pa.string())],
    PARTSUPP: [("ps_partkey", pa.int32()), ("ps_suppkey", pa.int32()), ("ps_availqty", pa.int32()), ("ps_supplycost", DEC)],
    NATION: [("n_nationkey", pa.int32()), ("n_regionkey", pa.int32()), ("n_name", pa.string())],
    REGION: [("r_regionkey", pa.int32()), ("r_name", pa.string())],
    PROBEKEYS: [("k_orderkey", pa.int32())],
}


def host_column(table_id, col, n_orders, part=0, n_parts=1):
    lib = capi.host_lib()
    name, typ = SCHEMAS[table_id][col]
    n = lib.ldb_tpch_host_rows(table_id, n_orders, part, n_parts)
    nbytes = C.c_int64()
    if pa.types.is_string(typ):
        offs = np.zeros(n + 1, dtype=np.int64)
        lib.ldb_tpch_host_column(table_id, col, n_orders, part, n_parts, None, offs.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nbytes))
        data = np.zeros(max(nbytes.value, 1), dtype=np.uint8)
        lib.ldb_tpch_host_column(table_id, col, n_orders, part, n_parts, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nbytes))
        o32 = offs.astype(np.int32)
        return pa.Array.from_buffers(pa.string(), n, [None, pa.py_buffer(o32.tobytes()), pa.py_buffer(data.tobytes())])
    width = 16 if pa.types.is_decimal(typ) else 4
    buf = np.zeros(max(n * width, 1), dtype=np.uint8)
    lib.ldb_tpch_host_column(table_id, col, n_orders, part, n_parts, buf.ctypes.data_as(C.c_void_p), None, C.byref(nbytes))
    return pa.Array.from_buffers(typ, n, [None, pa.py_buffer(buf[: n * width].tobytes())])


def host_table(table_id, n_orders, part=0, n_parts=1, cols=None):
    fields = SCHEMAS[table_id]
    idx = list(range(len(fields))) if cols is None else list(cols)
    arrays = [host_column(table_id, c, n_orders, part, n_parts) for c in idx]
    return pa.Table.from_arrays(arrays, names=[fields[c][0] for c in idx])
