"""np_oracle.py - TEST INFRASTRUCTURE ONLY (oracle).  Never imported by the product.

Vectorised numpy restatement of what the reference's SELECT executor computes on the
hot path (reference src/engine/executor_select.c), at the granularity of the device
operators of include/mdb_dev.h, for inputs far larger than the reference itself can
process (its join is O(nA*nB): ~1.86 M row pairs/s).  Small cases of every
function here are pinned against the real reference (oracle/_ref) and against the
pure-Python line-by-line restatement (oracle/naive.py) in tests/test_oracle_pinning.py.

Semantics restated, with the reference lines they come from:
  * join_pairs: for every live left row (outer loop, table order) and every live right row
    (inner loop) keep the pair when ON l = r holds (:1096-1141); a NULL operand makes the
    comparison false (:557-579) -> output sorted by (left position, right position).
  * group_count: each live row deletes every later row with an equal key and increments its
    own COUNT column once per deleted row (:1542-1583, :1501-1524); survivors keep table
    order = first-occurrence order; NULL keys compare equal to each other (:1477-1482).
  * join_group_count = group_count over the key column of join_pairs' output (the
    north-star query, reference tests/engine/executor_select.c:348-378).
  * filter: comparison with a NULL operand is false (:629-631), IS [NOT] NULL reads the NULL
    bit (:965), AND/OR/XOR on plain booleans (:1041-1058).
All integer comparisons here are full 64-bit; the reference truncates to 32 bits (SURVEY 8a
D5), so parity inputs stay inside [-2^31, 2^31).
"""
import numpy as np


def _valid_index(n, nulls):
    if nulls is None:
        return np.arange(n, dtype=np.int64)
    return np.flatnonzero(~np.asarray(nulls, dtype=bool)).astype(np.int64)


def join_pairs(keys_l, nulls_l, keys_r, nulls_r):
    """-> (pos_l, pos_r) int64 arrays sorted by (pos_l, pos_r)."""
    keys_l = np.asarray(keys_l, dtype=np.int64)
    keys_r = np.asarray(keys_r, dtype=np.int64)
    vl = _valid_index(len(keys_l), nulls_l)
    vr = _valid_index(len(keys_r), nulls_r)
    kr = keys_r[vr]
    order = np.argsort(kr, kind="stable")		# right rows grouped by key, ascending position inside a key
    kr_sorted = kr[order]
    pr_sorted = vr[order]
    kl = keys_l[vl]
    lo = np.searchsorted(kr_sorted, kl, side="left")
    hi = np.searchsorted(kr_sorted, kl, side="right")
    m = hi - lo
    total = int(m.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    pos_l = np.repeat(vl, m)
    starts = np.repeat(lo, m)
    first_out = np.cumsum(m) - m
    within = np.arange(total, dtype=np.int64) - np.repeat(first_out, m)
    pos_r = pr_sorted[starts + within]
    return pos_l, pos_r


def group_count(keys, nulls):
    """-> (first_pos, count) in first-occurrence order; NULL keys form one group."""
    keys = np.asarray(keys, dtype=np.int64)
    n = len(keys)
    v = _valid_index(n, nulls)
    firsts, counts = [], []
    if len(v):
        _, idx, cnt = np.unique(keys[v], return_index=True, return_counts=True)
        firsts.append(v[idx])
        counts.append(cnt.astype(np.int64))
    if nulls is not None:
        nz = np.flatnonzero(np.asarray(nulls, dtype=bool))
        if len(nz):
            firsts.append(nz[:1].astype(np.int64))
            counts.append(np.array([len(nz)], dtype=np.int64))
    if not firsts:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    first = np.concatenate(firsts)
    count = np.concatenate(counts)
    o = np.argsort(first, kind="stable")
    return first[o], count[o]


def join_group_count(keys_l, nulls_l, keys_r, nulls_r):
    """-> (key, count, first_left_pos, joined_rows) in the reference's group order.

    Equivalent to group_count over the left key of join_pairs' output without building
    the pairs: COUNT(*) of key k = (#left rows with k) * (#right rows with k); groups are
    ordered by their first joined row, i.e. by the first left position holding the key.
    """
    keys_l = np.asarray(keys_l, dtype=np.int64)
    keys_r = np.asarray(keys_r, dtype=np.int64)
    vl = _valid_index(len(keys_l), nulls_l)
    vr = _valid_index(len(keys_r), nulls_r)
    if len(vl) == 0 or len(vr) == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z, z, 0
    ul, idx_l, cnt_l = np.unique(keys_l[vl], return_index=True, return_counts=True)
    ur, cnt_r = np.unique(keys_r[vr], return_counts=True)
    pos = np.searchsorted(ur, ul)
    pos_c = np.minimum(pos, len(ur) - 1)
    hit = ur[pos_c] == ul
    key = ul[hit]
    count = cnt_l[hit].astype(np.int64) * cnt_r[pos_c[hit]].astype(np.int64)
    first = vl[idx_l[hit]]
    o = np.argsort(first, kind="stable")
    return key[o], count[o], first[o], int(count.sum())


# ---- predicates ---------------------------------------------------------------------------------
_CMP = {
    1: lambda a, b: a < b,
    2: lambda a, b: a > b,
    3: lambda a, b: a != b,
    4: lambda a, b: a == b,
    5: lambda a, b: a <= b,
    6: lambda a, b: a >= b,
}


def filter_positions(prog, cols, n):
    """Evaluate a postfix predicate program (same encoding as mdb_pred_insn) -> passing positions.

    cols: list of (values ndarray (int64 or float64), nulls bool ndarray or None, rid ndarray or None).
    """
    def load(slot):
        vals, nulls, rid = cols[slot]
        vals = np.asarray(vals)
        if rid is not None:
            rid = np.asarray(rid, dtype=np.int64)
            v = vals[rid]
            isnull = np.asarray(nulls, dtype=bool)[rid] if nulls is not None else np.zeros(n, dtype=bool)
        else:
            v = vals[:n]
            isnull = np.asarray(nulls, dtype=bool)[:n] if nulls is not None else np.zeros(n, dtype=bool)
        return v, isnull

    def const(imm, typ):
        return np.float64(imm) if typ == 1 else np.int64(imm)

    st = []
    for (op, cmp_, typ, a, b, imm) in prog:
        if op == 1:
            v, isnull = load(a)
            st.append(~isnull & _CMP[cmp_](v, const(imm, typ)))
        elif op == 2:
            v, isnull = load(a)
            st.append(~isnull & _CMP[cmp_](const(imm, typ), v))
        elif op == 3:
            va, na = load(a)
            vb, nb = load(b)
            st.append(~na & ~nb & _CMP[cmp_](va, vb))
        elif op == 4:
            _, isnull = load(a)
            st.append(isnull ^ bool(cmp_))
        elif op == 5:
            st.append(np.full(n, bool(imm)))
        else:
            r = st.pop()
            l = st.pop()
            st.append((l & r) if op == 6 else ((l | r) if op == 7 else (l ^ r)))
    assert len(st) == 1
    return np.flatnonzero(st[0]).astype(np.int64)


# ---- multi-GPU shuffle ---------------------------------------------------------------------------
_M = (1 << 64) - 1


def fmix64(k):
    """murmur3 finaliser on uint64 arrays (the device's hash; used only to mirror the
    destination-GPU assignment hash(key) mod nGPU in the gloo tests)."""
    k = np.asarray(k).astype(np.uint64)
    k = k ^ (k >> np.uint64(33))
    k = (k * np.uint64(0xff51afd7ed558ccd)) & np.uint64(_M)
    k = k ^ (k >> np.uint64(33))
    k = (k * np.uint64(0xc4ceb9fe1a85ec53)) & np.uint64(_M)
    k = k ^ (k >> np.uint64(33))
    return k


def dest_of(keys, n_dest):
    hv = fmix64(np.asarray(keys, dtype=np.int64).view(np.uint64))
    return ((hv & np.uint64(0xFFFFFFFF)) % np.uint64(n_dest)).astype(np.int64)


def mixk(x, k):
    """include csrc/mdb_dev_common.h mdb_mixk: the murmur3 finaliser on k-bit words (a bijection of [0, 2^k))"""
    x = np.asarray(x, dtype=np.uint64)
    mask = np.uint64((1 << k) - 1)
    s = np.uint64((k + 1) >> 1)
    x = x ^ (x >> s)
    x = (x * np.uint64(0x85EBCA6B)) & mask
    x = x ^ (x >> s)
    x = (x * np.uint64(0xC2B2AE35)) & mask
    x = x ^ (x >> s)
    return x


def dest_of_fused(keys, n_dest, r_lo, r_span):
    """destination rank of a key when the sharded operator ships first-level regions (mdb_dev_shard.hip): the top log2(n_dest)
    bits of the k-bit hash of key - r_lo, k = the bits of the right table's global key range (at least 13)"""
    k = max(int(np.ceil(np.log2(max(int(r_span), 1)))), 13)
    while (1 << k) < r_span:
        k += 1
    with np.errstate(over="ignore"):
        rel = (np.asarray(keys, dtype=np.int64) - np.int64(r_lo)).astype(np.uint64)
    h = mixk(rel, k)
    return (h >> np.uint64(k - 9)).astype(np.int64) // (512 // n_dest)


def partition_by_dest(keys, nulls, n_dest):
    """-> (keys grouped by destination, stable inside a destination; counts per destination)."""
    keys = np.asarray(keys, dtype=np.int64)
    v = _valid_index(len(keys), nulls)
    k = keys[v]
    d = dest_of(k, n_dest)
    o = np.argsort(d, kind="stable")
    return k[o], np.bincount(d, minlength=n_dest).astype(np.int64)


# ---- synthetic data (include/mdb_gen.h) -----------------------------------------------------------
def splitmix64(state):
    state = (state + 0x9e3779b97f4a7c15) & _M
    z = state
    z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & _M
    z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & _M
    return state, z ^ (z >> 31)


def _is_prime(x):
    if x < 2:
        return False
    if x % 2 == 0:
        return x == 2
    d = 3
    while d * d <= x:
        if x % d == 0:
            return False
        d += 2
    return True


def perm_make(n, seed):
    p = max(n, 2)
    while not _is_prime(p):
        p += 1
    lim = min(p - 1, 1 << 30)
    s, r1 = splitmix64(seed)
    s, r2 = splitmix64(s)
    return dict(n=n, p=p, a=1 + r1 % lim, b=r2 % p)


def gen_keys(n, first_index, domain, seed, modulus=0):
    """numpy twin of mdb_dev_gen_keys / mdb_perm_apply."""
    pm = perm_make(domain, seed)
    a, b, p = np.uint64(pm["a"]), np.uint64(pm["b"]), np.uint64(pm["p"])
    x = np.arange(first_index, first_index + n, dtype=np.uint64)
    x = (a * x + b) % p
    while True:
        bad = x >= np.uint64(domain)
        if not bad.any():
            break
        x[bad] = (a * x[bad] + b) % p
    if modulus:
        x = x % np.uint64(modulus)
    return x.astype(np.int64)


def gen_payload(n, first_index, seed, kind):
    """include/mdb_gen.h mdb_splitmix64_at: cell i = output first_index + i of SplitMix64(seed); kind 0: INT64 = z >> 33,
    kind 1: DOUBLE = (z >> 11) * 2^-53"""
    with np.errstate(over="ignore"):
        i = np.arange(first_index, first_index + n, dtype=np.uint64)
        z = np.uint64(seed) + (i + np.uint64(1)) * np.uint64(0x9e3779b97f4a7c15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
        z = z ^ (z >> np.uint64(31))
    if kind == 1:
        return (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return (z >> np.uint64(33)).astype(np.int64)


def sort_perm(keys, n):
    """Oracle of mdb_dev_sort_perm (ORDER BY; an extension, the reference never executes ORDER BY - SURVEY 8a D7):
    keys = [(values, nulls bool or None, rid or None, is_double, desc)], row i of the stream reads values[rid[i]].
    Stable; NULL is the smallest value (first for ASC, last for DESC); DOUBLE in IEEE order with -0.0 < +0.0.
    Plain numpy: one stable argsort per key, last key first."""
    perm = np.arange(n, dtype=np.int64)
    for values, nulls, rid, is_double, desc in reversed(keys):
        v = np.asarray(values)
        rows = perm if rid is None else np.asarray(rid, dtype=np.int64)[perm]
        bits = v.view(np.uint64)[rows] if v.dtype != np.uint64 else v[rows]
        if is_double:
            neg = (bits >> np.uint64(63)).astype(bool)
            img = np.where(neg, ~bits, bits ^ np.uint64(1 << 63))
        else:
            img = bits ^ np.uint64(1 << 63)
        if desc:
            img = ~img
        isnull = np.zeros(n, dtype=bool) if nulls is None else np.asarray(nulls, dtype=bool)[rows]
        img = np.where(isnull, np.uint64(0), img)		# NULL rows tie with each other: they keep their order
        order = np.argsort(img, kind="stable")
        perm, isnull = perm[order], isnull[order]
        flag = (isnull == bool(desc)).astype(np.uint8)	# ASC: NULL -> 0 (first); DESC: NULL -> 1 (last)
        perm = perm[np.argsort(flag, kind="stable")]
    return perm.astype(np.uint32)


def distinct_sel(keys, n):
    """Oracle of mdb_dev_distinct_sel: ascending stream positions of the first occurrence of every distinct key
    combination (NULL equals NULL, values compared by their 64 bits)."""
    seen, out = set(), []
    cols = []
    for values, nulls, rid, _is_double, _desc in keys:
        v = np.asarray(values)
        rows = np.arange(n) if rid is None else np.asarray(rid, dtype=np.int64)
        bits = v.view(np.uint64)[rows]
        isnull = np.zeros(n, dtype=bool) if nulls is None else np.asarray(nulls, dtype=bool)[rows]
        cols.append((np.where(isnull, np.uint64(0), bits), isnull))
    for i in range(n):
        k = tuple((int(b[i]), bool(z[i])) for b, z in cols)
        if k not in seen:
            seen.add(k)
            out.append(i)
    return np.array(out, dtype=np.uint32)


def group_count_multi(keys, n):
    """Oracle of mdb_dev_group_count_multi: rows are one group when they agree on every key column (NULL = NULL,
    values by their 64 bits); -> (first positions ascending, counts)."""
    cols = []
    for values, nulls, rid, _is_double, _desc in keys:
        v = np.asarray(values)
        rows = np.arange(n) if rid is None else np.asarray(rid, dtype=np.int64)
        bits = v.view(np.uint64)[rows]
        isnull = np.zeros(n, dtype=bool) if nulls is None else np.asarray(nulls, dtype=bool)[rows]
        cols.append((np.where(isnull, np.uint64(0), bits), isnull))
    first, count = {}, {}
    for i in range(n):
        k = tuple((int(b[i]), bool(z[i])) for b, z in cols)
        if k not in first:
            first[k] = i
            count[k] = 0
        count[k] += 1
    order = sorted(first, key=lambda k: first[k])
    return np.array([first[k] for k in order], dtype=np.int64), np.array([count[k] for k in order], dtype=np.int64)
