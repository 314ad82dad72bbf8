"""Transcribes the reference's SQL-level join tests (test/sqlite-small/join.test, 22 of its 23
live cases — the `x = all(... where y <= x)` case at :134 needs a non-equi correlated residual and
is left out) into operator-level fixtures: the VALUES lists parsed from the SQL text, the join kind
the reference's plan uses, and the expected rows parsed from the lines under `----`.  What the SQL
frontend does before the join (casting mixed int/decimal keys to the common decimal type,
`sql_analyzer.cpp:3083-3159`) is applied here: every value is stored as an integer at `scale`
decimal digits.  Writes sqlite_small_join.json.  Needs /root/reference."""
import json
import os
import re
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/test/sqlite-small/join.test"


def values_lists(sql):
    """[(alias, [value text …])] of the `(values(a),(b),…) alias(col)` items of a query, in order"""
    out = []
    for m in re.finditer(r"\(values\s*((?:\([^()]*\),?)+)\)\s*(\w)\(", sql):
        out.append((m.group(2), re.findall(r"\(([^()]*)\)", m.group(1))))
    return out


def kind_of(sql):
    if "left outer join" in sql:
        return "left_outer"
    if "right outer join" in sql:
        return "right_outer"
    if "full outer join" in sql:
        return "full_outer"
    if "not exists" in sql:
        return "anti"
    if "exists" in sql:
        return "semi"
    if "=some" in sql:
        return "mark"
    return "inner"


def main():
    lines = open(SRC).read().split("\n")
    cases = []
    i = 0
    while i < len(lines):
        if lines[i].startswith("query "):
            sql, line_no = lines[i + 1], i + 2
            j = i + 3  # after ----
            rows = []
            while j < len(lines) and lines[j].strip():
                rows.append(lines[j].split("\t"))
                j += 1
            i = j
            if "=all" in sql:
                continue
            vl = dict(values_lists(sql))
            s, t = vl["s"], vl["t"]
            m = re.search(r"where y>(\d+)\)", sql)  # the filter under the mark join of :157
            if m:
                t = [v for v in t if Fraction(v) > int(m.group(1))]
            texts = [v for v in s + t if v != "NULL"] + [c for r in rows for c in r if c not in ("NULL", "t", "f")]
            scale = max([len(v.split(".")[1]) if "." in v else 0 for v in texts] + [0])
            conv = lambda v: None if v == "NULL" else int(Fraction(v) * 10 ** scale)  # noqa: E731
            exp = [[{"t": True, "f": False}[c] if c in ("t", "f") else conv(c) for c in r] for r in rows]
            cases.append({"source": f"test/sqlite-small/join.test:{line_no}", "sql": sql, "kind": kind_of(sql), "scale": scale,
                          "s": [conv(v) for v in s], "t": [conv(v) for v in t], "expected": exp})
        else:
            i += 1
    with open(os.path.join(HERE, "sqlite_small_join.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
