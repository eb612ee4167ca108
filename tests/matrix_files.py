"""Test-side reader of the reference's matrix files (tests/golden/tcov = CHOLMOD/Tcov/Matrix, tests/golden/demo =
CHOLMOD/Demo/Matrix), written from the format notes of CHOLMOD/Check/cholmod_read.c:9-140 -- independent of the product's
reader (csrc/host/io.c), which the tests compare against it.

read_file (path) -> dict(kind="sparse", nrow, ncol, stype, xtype in {"real","complex","pattern"}, Ap, Ai, Ax)   CSC, sorted,
                    duplicates summed, only the triangle `stype` names kept
                  | dict(kind="dense", nrow, ncol, X)        Matrix Market "array"
                  | dict(kind="invalid", why)
"""
import numpy as np


def _tokens(path):
    mm = None
    lines = []
    with open(path, "rb") as f:
        raw = f.read().decode("latin-1")
    first = True
    for ln in raw.splitlines():
        t = ln.strip()
        if first and t.lower().startswith("%%matrixmarket"):
            w = t.lower().split()
            mm = dict(fmt=w[2] if len(w) > 2 else "coordinate", typ=w[3] if len(w) > 3 else "real",
                      sto=w[4] if len(w) > 4 else "general")
        if t:
            first = False
        if not t or t.startswith("%"):
            continue
        lines.append(t.split())
    return mm, lines


def _num(tok):
    t = tok.lower()
    if t in ("inf", "+inf"):
        return np.inf
    if t == "-inf":
        return -np.inf
    return float(tok)


def read_file(path, prefer_binary=False):
    try:
        mm, lines = _tokens(path)
    except Exception as e:           # pragma: no cover
        return dict(kind="invalid", why=repr(e))
    if not lines:
        return dict(kind="invalid", why="no data line")
    try:
        head = [float(v) for v in lines[0]]
    except ValueError:
        return dict(kind="invalid", why="header is not numeric")
    if any(h != int(h) for h in head):
        return dict(kind="invalid", why="header is not integral")
    head = [int(h) for h in head]
    sto = (mm or {}).get("sto", None)
    if len(head) == 2 or (mm and mm["fmt"].startswith("a")):
        if len(head) < 2:
            return dict(kind="invalid", why="dense header")
        nrow, ncol = head[:2]
        if nrow < 0 or ncol < 0:
            return dict(kind="invalid", why="negative dimension")
        ent = lines[1:]
        sym = sto[0] if sto else "g"
        skew = bool(sto) and sto.startswith("sk")
        cx = bool(ent) and len(ent[0]) >= 2
        X = np.zeros((nrow, ncol), dtype=np.complex128 if cx else np.float64)
        k = 0
        try:
            for j in range(ncol):
                i0 = 0 if sym == "g" else (j + 1 if skew else j)
                for i in range(i0, nrow):
                    v = _num(ent[k][0]) + (1j * _num(ent[k][1]) if cx else 0)
                    k += 1
                    X[i, j] = v
                    if sym != "g" and i != j:
                        X[j, i] = -v if skew else (np.conj(v) if sym == "h" else v)
        except (IndexError, ValueError):
            return dict(kind="invalid", why="dense body")
        return dict(kind="dense", nrow=nrow, ncol=ncol, X=X)
    if len(head) < 3:
        return dict(kind="invalid", why="header")
    nrow, ncol, nnz = head[:3]
    if nrow < 0 or ncol < 0 or nnz < 0:
        return dict(kind="invalid", why="negative header")
    stype = None
    if sto is not None:
        stype = -1 if (sto[0] in "sh" and not sto.startswith("sk")) else 0
    if len(head) >= 4:
        stype = head[3]
        if stype not in (-1, 0, 1):
            # (the reference takes any fourth integer: < 0 lower, > 0 upper)
            stype = -1 if stype < 0 else 1
    if nrow == 0 or ncol == 0 or nnz == 0:
        # an empty matrix comes back unsymmetric whatever the header says, the rest of the file unread (cholmod_read.c:519-525)
        return dict(kind="sparse", nrow=nrow, ncol=ncol, stype=0, xtype="real", Ap=np.zeros(ncol + 1 if ncol < 10**7 else 1, dtype=np.int64),
                    Ai=np.zeros(0, dtype=np.int64), Ax=np.zeros(0))
    if max(nrow, ncol) > 10 ** 8:
        # (dimensions no test machine allocates: shape and entry count only)
        return dict(kind="sparse", nrow=nrow, ncol=ncol, stype=0, xtype="real", Ap=np.array([0, nnz], dtype=np.int64),
                    Ai=np.zeros(0, dtype=np.int64), Ax=np.zeros(0))
    ent = lines[1:1 + nnz]
    if len(ent) < nnz:
        return dict(kind="invalid", why="premature end of file")
    ntok = len(ent[0]) if ent else 3
    if any(len(e) != ntok for e in ent) or ntok < 2 or ntok > 4:
        return dict(kind="invalid", why="ragged entries")
    try:
        ii = np.array([int(float(e[0])) for e in ent], dtype=np.int64)
        jj = np.array([int(float(e[1])) for e in ent], dtype=np.int64)
        if ntok == 2:
            vv = np.ones(nnz)
        elif ntok == 3:
            vv = np.array([_num(e[2]) for e in ent], dtype=np.float64)
        else:
            vv = np.array([_num(e[2]) + 1j * _num(e[3]) for e in ent], dtype=np.complex128)
    except ValueError:
        return dict(kind="invalid", why="entry is not numeric")
    xtype = {2: "pattern", 3: "real", 4: "complex"}[ntok]
    if nnz and ii.min() > 0 and jj.min() > 0:
        ii -= 1
        jj -= 1
    if nnz and (ii.min() < 0 or jj.min() < 0 or ii.max() >= nrow or jj.max() >= ncol):
        return dict(kind="invalid", why="index out of range")
    if stype is None:
        # one triangle only: symmetric with that triangle stored; a diagonal matrix counts as upper (cholmod_read.c:737-760)
        lo, up = bool(np.any(ii > jj)), bool(np.any(ii < jj))
        stype = 0 if (nrow != ncol or (lo and up)) else (-1 if lo else 1)
    skew = sto is not None and sto.startswith("sk")
    csym = sto is not None and sto[0] == "s" and not skew and xtype == "complex"
    if skew or csym:
        # returned with both triangles, stype 0 (cholmod_read.c:44-49)
        off = ii != jj
        ii, jj, vv = (np.concatenate([ii, jj[off]]), np.concatenate([jj, ii[off]]),
                      np.concatenate([vv, (-vv[off]) if skew else vv[off]]))
        if skew and xtype == "pattern":
            vv = np.ones(ii.size)
        stype = 0
    if nrow != ncol:
        stype = 0
    if xtype == "pattern" and stype != 0 and not prefer_binary:
        # diagonal = 1 + degree (entries of the stored triangle, as they stand in the file), off-diagonals -1 (:819-857)
        deg = np.zeros(max(nrow, 1), dtype=np.int64)
        tri = (ii > jj) if stype < 0 else (ii < jj)
        np.add.at(deg, ii[tri], 1)
        np.add.at(deg, jj[tri], 1)
        vv = np.where(ii == jj, deg[ii] + 1.0, -1.0)
    # entries in the other triangle are moved across the diagonal as they are (Core/t_cholmod_triplet.c:62-100)
    if stype < 0:
        sw = ii < jj
    elif stype > 0:
        sw = ii > jj
    else:
        sw = np.zeros(ii.size, dtype=bool)
    ii, jj = np.where(sw, jj, ii), np.where(sw, ii, jj)
    key = jj * max(nrow, 1) + ii
    order = np.argsort(key, kind="stable")
    key, ii, jj, vv = key[order], ii[order], jj[order], vv[order]
    if key.size:
        first = np.concatenate(([True], key[1:] != key[:-1]))
        grp = np.cumsum(first) - 1
        vs = np.zeros(int(grp[-1]) + 1, dtype=vv.dtype)
        np.add.at(vs, grp, vv)
        ii, jj, vv = ii[first], jj[first], vs
    Ap = np.zeros(ncol + 1, dtype=np.int64)
    np.add.at(Ap, jj + 1, 1)
    Ap = np.cumsum(Ap)
    return dict(kind="sparse", nrow=nrow, ncol=ncol, stype=int(stype), xtype=xtype, Ap=Ap, Ai=ii.astype(np.int64),
                Ax=vv if xtype == "complex" else vv.astype(np.float64))


def to_lower(m):
    """A symmetric / Hermitian file as lower-stored CSC (stype -1): the upper-stored ones are (conjugate-)transposed."""
    assert m["kind"] == "sparse" and m["stype"] != 0
    if m["stype"] < 0:
        return m["nrow"], m["Ap"], m["Ai"], m["Ax"]
    n = m["nrow"]
    jj = np.repeat(np.arange(n), np.diff(m["Ap"]))
    ii = m["Ai"]
    vv = np.conj(m["Ax"])
    # entry (i, j), i <= j, becomes (j, i): column i, row j
    order = np.lexsort((jj, ii))
    col, row, val = ii[order], jj[order], vv[order]
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.add.at(Ap, col + 1, 1)
    return n, np.cumsum(Ap), row.astype(np.int64), val


def dense_of(m):
    """The full dense matrix a sparse file stands for."""
    A = np.zeros((m["nrow"], m["ncol"]), dtype=np.complex128 if m["xtype"] == "complex" else np.float64)
    jj = np.repeat(np.arange(m["ncol"]), np.diff(m["Ap"]))
    A[m["Ai"], jj] = m["Ax"]
    if m["stype"] != 0:
        A = A + np.conj(A.T) - np.diag(np.real(np.diag(A)) if m["xtype"] == "complex" else np.diag(A))
        if m["xtype"] == "complex":
            # (the imaginary part of a Hermitian file's diagonal is ignored by the factorization)
            pass
    return A
