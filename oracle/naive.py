"""naive.py - TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product.

Pure-Python, loop-for-loop restatement of the reference's WHOLE SELECT pipeline
(reference src/engine/executor_select.c:1655-1744) for small inputs, taking the same
input the reference executor takes: the parser's RPN token queue (reference
src/parser/midorisql.y:517-528) plus the tables.  Phases and their reference lines:

    build_cols_hashtable + build_table_scafold   :267-322   -> result column set and djb2 order
    proc_from_clause_table (scan)                :1282-1343
    _join_nested_loop_tbl2tbl                    :1076-1149 -> nested loops, left-major
    (recursive join)                             :1151-1280 -> restated with the INTENDED semantics
                                                               ((A x B) x C); the reference's own
                                                               tbl2mat is defective
    proc_where_clause / eval_row_cond            :1435-1463, 1027-1074
    proc_groupby_clause                          :1526-1588 -> quadratic, first occurrence survives
                                                               (without the per-datablock restart of
                                                               defect D1)
    proc_select_clause                           :1369-1433 -> projection
    handle_countonly_case                        :1590-1653
    table_vacuum                                 :1726      -> survivors keep their order

It is pinned against golden vectors produced by the real reference (tests/test_oracle_pinning.py)
and is the whole-query checker for shapes the per-operator oracles (np_oracle.py) do not cover.
Values: Python ints (INT columns) or floats (DOUBLE); NULL = None.  Results are reported the way
query_column_int64() would return them: NULL cells read 0, DOUBLE cells as their raw 64-bit pattern.
"""
import struct


# ---- RPN -> tiny AST (same stack machine as reference src/parser/ast_select.c:1021-1140) -----------

def parse_rpn(rpn):
    st = []
    out = {}
    for t in [x for x in rpn.strip().split("\n") if x]:
        head, _, rest = t.partition(" ")
        if head == "NAME":
            st.append(("name", rest))
        elif head == "FIELDNAME":
            tbl, col = rest.split(".")
            st.append(("field", tbl, col))
        elif head == "NUMBER":
            st.append(("int", int(rest)))
        elif head == "FLOAT":
            st.append(("float", float(rest)))
        elif head == "NULL":
            st.append(("null",))
        elif head == "CMP":
            r, l = st.pop(), st.pop()
            st.append(("cmp", int(rest), l, r))
        elif head in ("AND", "OR", "XOR"):
            r, l = st.pop(), st.pop()
            st.append(("logop", head, l, r))
        elif head in ("ISNULL", "ISNOTNULL"):
            st.append(("isnull", head == "ISNOTNULL", st.pop()))
        elif head in ("ISIN", "ISNOTIN"):
            k = int(rest)
            vals = st[-k:]
            del st[-k:]
            st.append(("isin", head == "ISNOTIN", st.pop(), vals))
        elif head == "COUNTALL":
            st.append(("count",))
        elif head == "COUNTFIELD":
            st.pop()
            st.append(("count",))
        elif head == "SELECTALL":
            st.append(("selectall",))
        elif head == "TABLE":
            st.append(("table", rest, None))
        elif head == "ALIAS":
            x = st.pop()
            st.append(("table", x[1], rest) if x[0] == "table" else ("alias", rest, x))
        elif head == "ONEXPR":
            st.append(("onexpr", st.pop()))
        elif head == "JOIN":
            on, right, left = st.pop(), st.pop(), st.pop()
            st.append(("join", int(rest), left, right, on[1]))
        elif head == "WHERE":
            st.append(("where", st.pop()))
        elif head == "GROUPBYLIST":
            k = int(rest)
            g = st[-k:]
            del st[-k:]
            st.append(("groupby", g))
        elif head == "HAVING":
            st.append(("having", st.pop()))
        elif head == "ORDERBYITEM":
            st.append(("orderitem", int(rest), st.pop()))
        elif head == "ORDERBYLIST":
            k = int(rest)
            items = st[-k:]
            del st[-k:]
            st.append(("orderby", items))
        elif head == "LIMIT":
            k = int(rest)
            vals = st[-k:]
            del st[-k:]
            st.append(("limit", [v[1] for v in vals]))
        elif head == "SELECT":
            d, n = rest.split()
            kids = st[-int(n):]
            del st[-int(n):]
            out = {"sel": [], "from": [], "where": None, "group": [], "distinct": bool(int(d) & 2), "having": None, "order": [],
                   "limit": None}
            for k in kids:
                if k[0] in ("table", "join"):
                    out["from"].append(k)
                elif k[0] == "where":
                    out["where"] = k[1]
                elif k[0] == "groupby":
                    out["group"] = k[1]
                elif k[0] == "having":
                    out["having"] = k[1]
                elif k[0] == "orderby":
                    out["order"] = k[1]
                elif k[0] == "limit":
                    out["limit"] = k[1]
                else:
                    out["sel"].append(k)
        elif head == "ASSIGN":
            st.append(("assign", rest, st.pop()))
        elif head == "DELETEONE":
            out = {"kind": "delete", "table": rest, "where": st.pop()[1] if st else None}
        elif head == "UPDATE":
            tbl, na, hw = rest.split()
            where = st.pop()[1] if int(hw) else None
            out = {"kind": "update", "table": tbl, "assign": st[-int(na):], "where": where}
            del st[-int(na):]
        elif head == "STMT":
            break
        else:
            raise ValueError(f"naive oracle: unsupported token {t!r}")
    return out


# ---- result column order: the reference's chained hash table, replayed ---------------

def djb2(key):
    h = 5381
    for c in key.encode() + b"\0":		# strlen + 1 bytes: the NUL is hashed too (hashtable.c:269-281)
        h = (h * 33 + c) & 0xFFFFFFFFFFFFFFFF
    return h


def reference_column_order(keys):
    cap, buckets, count = 16, [[] for _ in range(16)], 0
    for k in keys:
        buckets[djb2(k) % cap].insert(0, k)			# list_add: new entry becomes the chain head
        count += 1
        if count / cap >= 0.5:					# hashtable_resize (hashtable.c:84-129)
            ncap = cap * 2
            nb = [[] for _ in range(ncap)]
            for b in buckets:
                for e in b:
                    nb[djb2(e) % ncap].insert(0, e)
            cap, buckets = ncap, nb
    return [e for b in buckets for e in b]


# ---- executor ------------------------------------------------------------------------------------------

def _cmp(op, a, b):
    return {1: a < b, 2: a > b, 3: a != b, 4: a == b, 5: a <= b, 6: a >= b}[op]


class Naive:
    def __init__(self, tables):
        """tables: {name: (column names, rows)}; rows = list of lists, None = NULL."""
        self.tables = tables

    def _tables_of(self, ref, out):
        if ref[0] == "table":
            out.append(ref)
        else:
            self._tables_of(ref[2], out)
            self._tables_of(ref[3], out)

    def _resolve(self, e, tabs):
        """NAME -> fully qualified field, alias -> table name (reference optimiser_select.c:114-183)."""
        if e[0] == "name":
            hits = [(t[1], e[1]) for t in tabs if e[1] in self.tables[t[1]][0]]
            assert len(hits) == 1, f"column {e[1]!r} not unique / unknown"
            return ("field",) + hits[0]
        if e[0] == "field":
            for t in tabs:
                if e[1] == t[2] or e[1] == t[1]:
                    return ("field", t[1], e[2])
            raise AssertionError(f"table {e[1]!r} not in FROM")
        if e[0] == "cmp":
            return ("cmp", e[1], self._resolve(e[2], tabs), self._resolve(e[3], tabs))
        if e[0] == "logop":
            return ("logop", e[1], self._resolve(e[2], tabs), self._resolve(e[3], tabs))
        if e[0] == "isnull":
            return ("isnull", e[1], self._resolve(e[2], tabs))
        if e[0] == "isin":
            return ("isin", e[1], self._resolve(e[2], tabs), e[3])
        if e[0] == "orderitem":
            return ("orderitem", e[1], self._resolve(e[2], tabs))
        return e

    def _value(self, e, row):
        if e[0] == "field":
            return row[f"{e[1]}.{e[2]}"]
        if e[0] in ("int", "float"):
            return e[1]
        if e[0] == "null":
            return None
        if e[0] == "count":			# HAVING COUNT(*) <op> n
            return row["COUNT(*)"]
        raise AssertionError(e)

    def _cond(self, e, row):
        if e[0] == "cmp":					# eval_cmp :865-921; NULL operand => false (:557-579)
            a, b = self._value(e[2], row), self._value(e[3], row)
            return a is not None and b is not None and _cmp(e[1], a, b)
        if e[0] == "logop":					# :1041-1058
            l, r = self._cond(e[2], row), self._cond(e[3], row)
            return (l and r) if e[1] == "AND" else ((l or r) if e[1] == "OR" else (l != r))
        if e[0] == "isnull":					# :965
            return (self._value(e[2], row) is None) != e[1]
        if e[0] == "isin":					# SQL semantics; the reference's conjunction is defect D3
            v = self._value(e[2], row)
            hits = [v is not None and x[0] != "null" and v == x[1] for x in e[3]]
            return (not any(hits) and v is not None and all(x[0] != "null" for x in e[3])) if e[1] else any(hits)
        raise AssertionError(e)

    def _from(self, ref, tabs):
        if ref[0] == "table":					# proc_from_clause_table :1282-1343
            cols, rows = self.tables[ref[1]]
            return [{f"{ref[1]}.{c}": v for c, v in zip(cols, r)} for r in rows]
        left, right = self._from(ref[2], tabs), self._from(ref[3], tabs)
        on = self._resolve(ref[4], tabs)
        out = []
        for a in left:						# outer = left rows in order (:1096-1106)
            for b in right:					# inner = right rows (:1108-1119)
                m = dict(a)
                m.update(b)
                if self._cond(on, m):				# ON evaluated on the merged row (:1128)
                    out.append(m)
        return out

    def run(self, rpn):
        q = parse_rpn(rpn)
        tabs = []
        for ref in q["from"]:
            self._tables_of(ref, tabs)
        sel = [s if s[0] in ("count", "selectall") else self._resolve(s, tabs) for s in q["sel"]]
        has_count = any(s[0] == "count" for s in sel)
        # comma-separated FROM -> synthetic JOIN ... ON 1=1 (optimiser_select.c:395-464)
        ref = q["from"][0]
        for nxt in q["from"][1:]:
            ref = ("join", 1, ref, nxt, ("cmp", 4, ("int", 1), ("int", 1)))
        rows = self._from(ref, tabs)
        for r in rows:
            r["COUNT(*)"] = 1					# init_count_cols :324-338
        if q["where"] is not None:				# proc_where_clause :1435-1463
            w = self._resolve(q["where"], tabs)
            rows = [r for r in rows if self._cond(w, r)]
        if q["group"]:						# proc_groupby_clause :1526-1588
            # One field: the reference's loop, restated (quadratic, first occurrence survives).  Several fields: the
            # reference runs that loop once per field, one after the other (:1537-1541), which collapses
            # `GROUP BY a, b` over {(1,1),(1,2),(2,1),...} into ONE row - not a grouping in any sense (verified
            # against oracle/_ref).  Like D1/D2 the intended semantics are restated instead: rows are equal when
            # they agree on EVERY group field (NULL equal to NULL, as in the single-field loop :1477-1478).
            gks = ["{}.{}".format(*self._resolve(g, tabs)[1:]) for g in q["group"]]
            alive = [True] * len(rows)
            for i in range(len(rows)):
                if not alive[i]:
                    continue
                for j in range(i + 1, len(rows)):
                    if alive[j] and all(rows[i][gk] == rows[j][gk] for gk in gks):
                        alive[j] = False
                        rows[i]["COUNT(*)"] += 1
            rows = [r for r, a in zip(rows, alive) if a]
        keys = (["COUNT(*)"] if has_count else []) + [f"{t[1]}.{c}" for t in tabs for c in self.tables[t[1]][0]]
        order = reference_column_order(keys)
        if any(s[0] == "selectall" for s in sel):
            wanted = set(keys)
        else:
            wanted = {"COUNT(*)" if s[0] == "count" else f"{s[1]}.{s[2]}" for s in sel}
        names = [k for k in order if k in wanted]		# proc_select_clause :1369-1433
        if names == ["COUNT(*)"] and not q["group"]:		# handle_countonly_case :1590-1653
            rows = [{"COUNT(*)": len(rows)}] if rows else []
        # ---- HAVING / DISTINCT / ORDER BY / LIMIT: the reference parses and checks these clauses but its executor
        #      never reads them, so there is no reference behaviour to pin: plain SQL semantics, the
        #      specification of the 8f row 4 extension (NULL sorts lowest, sorting is stable, DISTINCT keeps first
        #      occurrences, LIMIT off, cnt as in the grammar midorisql.y:193-196)
        if q.get("having") is not None:
            h = self._resolve(q["having"], tabs)
            rows = [r for r in rows if self._cond(h, r)]
        if q.get("distinct"):
            seen, keep = set(), []
            for r in rows:
                k = tuple((r[n] is None, self._raw(r[n])) for n in names)
                if k not in seen:
                    seen.add(k)
                    keep.append(r)
            rows = keep
        for it in reversed(q.get("order") or []):
            _, desc, f = self._resolve(it, tabs)
            col = f"{f[1]}.{f[2]}"
            # sorted() is stable, also with reverse=True (ties keep their order)
            rows = sorted(rows, key=lambda r: (r[col] is not None, self._order_image(r[col])), reverse=bool(desc))
        if q.get("limit"):
            off, cnt = (0, q["limit"][0]) if len(q["limit"]) == 1 else q["limit"]
            rows = rows[off:off + cnt]
        return names, [tuple(self._raw(r[n]) for n in names) for r in rows]

    @staticmethod
    def _order_image(v):
        """order-preserving integer image of a cell (IEEE total order for DOUBLE: -0.0 < +0.0), 0 for NULL"""
        if v is None:
            return 0
        if isinstance(v, float):
            b = struct.unpack("<Q", struct.pack("<d", v))[0]
            return (~b & 0xFFFFFFFFFFFFFFFF) if b >> 63 else (b | (1 << 63))
        return int(v) + (1 << 63)

    # ---- DELETE / UPDATE (reference src/engine/executor_delete.c:412-440, executor_update.c:460-484) ------
    def run_dml(self, rpn):
        """Executes one DELETE or UPDATE statement on self.tables; returns the rows affected
        (query_output.n_rows_aff).  Row predicate = eval_delete_row / should_update_row
        (executor_delete.c:354-410, executor_update.c:318-392): same comparison rules as SELECT's WHERE
        (NULL operand -> false), evaluated on the row's OLD values; UPDATE then applies every assignment
        (set_field_to_value, executor_update.c:394-433).  DELETE keeps the survivors in scan order:
        the reference only flags the row (table_delete_row) and every later scan skips flagged rows."""
        q = parse_rpn(rpn)
        cols, rows = self.tables[q["table"]]
        tabs = [("table", q["table"], None)]
        where = self._resolve(q["where"], tabs) if q["where"] is not None else None

        def hit(r):
            if where is None:
                return True
            return self._cond(where, {f"{q['table']}.{c}": v for c, v in zip(cols, r)})

        if q["kind"] == "delete":
            keep = [r for r in rows if not hit(r)]
            n = len(rows) - len(keep)
            rows[:] = keep
            return n
        n = 0
        for r in rows:
            if hit(r):
                for _, col, val in q["assign"]:
                    r[cols.index(col)] = None if val[0] == "null" else val[1]
                n += 1
        return n

    @staticmethod
    def _raw(v):
        if v is None:
            return 0						# cpy_cols skips the copy into the zeroed row (:384-387)
        if isinstance(v, float):
            return struct.unpack("<q", struct.pack("<d", v))[0]	# query_column_int64 returns the raw bits (query.c:162-166)
        return int(v)
