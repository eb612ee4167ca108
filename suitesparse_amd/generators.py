"""Synthetic SPD inputs and orderings for the CHOLMOD supernodal path.

Definitions follow SURVEY.md section 8d / appendix D (which record how the
reference was probed):

* 3D 7-point Poisson on an m^3 grid, lower triangle stored (stype -1),
  p = x + m*(y + m*z): A(p,p)=6, A(p+1,p)=A(p+m,p)=A(p+m^2,p)=-1.
* 2D 5-point Poisson m^2: A(p,p)=4, A(p+1,p)=A(p+m,p)=-1.
* box stencil m^3 radius r ("nd24k stand-in"): A(q,p)=-1 for every grid point
  q>p with |dx|,|dy|,|dz|<=r, A(p,p)=(#neighbours of p)+1.
* geometric nested dissection: recursive coordinate bisection, separator plane
  last, leaf boxes (all extents <= leaf) in lexicographic order.
* the demo right-hand side b(i) = 1 + i/n (reference
  CHOLMOD/Demo/cholmod_l_demo.c:231-239).

All matrices are returned as packed CSC (Ap, Ai, Ax) with int64 indices and
sorted columns -- the layout of the reference's cholmod_sparse with
itype=CHOLMOD_LONG (CHOLMOD/Include/cholmod_core.h:1243).
"""
from __future__ import annotations

import numpy as np


def _csc_from_lower_coo(n, rows, cols, vals):
    order = np.lexsort((rows, cols))
    rows = rows[order].astype(np.int64)
    cols = cols[order]
    vals = vals[order].astype(np.float64)
    Ap = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(cols, minlength=n), out=Ap[1:])
    return Ap, rows, vals


def poisson3d(m: int, my: int | None = None, mz: int | None = None):
    """Lower-stored 7-point Poisson on an m x my x mz grid (stype = -1)."""
    mx = m
    my = m if my is None else my
    mz = m if mz is None else mz
    n = mx * my * mz
    p = np.arange(n, dtype=np.int64)
    x = p % mx
    y = (p // mx) % my
    z = p // (mx * my)
    # column p holds rows p, p + 1, p + mx, p + mx my (those that exist), already in ascending order: the CSC arrays are
    # written in place, without the sort of the general path (8 M columns: seconds instead of minutes)
    if mx >= 2 and my >= 2:
        masks = (x + 1 < mx, y + 1 < my, z + 1 < mz)
        offs = (1, mx, mx * my)
        cnt = np.ones(n, dtype=np.int64)
        for mk in masks:
            cnt += mk
        Ap = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(cnt, out=Ap[1:])
        Ai = np.empty(int(Ap[-1]), dtype=np.int64)
        Ax = np.full(int(Ap[-1]), -1.0)
        pos = Ap[:-1].copy()
        Ai[pos] = p
        Ax[pos] = 6.0
        pos += 1
        for mk, off in zip(masks, offs):
            Ai[pos[mk]] = p[mk] + off
            pos += mk
        return n, Ap, Ai, Ax
    rows = [p]
    cols = [p]
    vals = [np.full(n, 6.0)]
    for mask, off in ((x + 1 < mx, 1), (y + 1 < my, mx), (z + 1 < mz, mx * my)):
        q = p[mask]
        rows.append(q + off)
        cols.append(q)
        vals.append(np.full(q.size, -1.0))
    return (n,) + _csc_from_lower_coo(n, np.concatenate(rows), np.concatenate(cols),
                                      np.concatenate(vals))


def poisson2d(m: int, my: int | None = None):
    """Lower-stored 5-point Poisson on an m x my grid (stype = -1)."""
    mx = m
    my = m if my is None else my
    n = mx * my
    p = np.arange(n, dtype=np.int64)
    x = p % mx
    y = p // mx
    rows = [p]
    cols = [p]
    vals = [np.full(n, 4.0)]
    for mask, off in ((x + 1 < mx, 1), (y + 1 < my, mx)):
        q = p[mask]
        rows.append(q + off)
        cols.append(q)
        vals.append(np.full(q.size, -1.0))
    return (n,) + _csc_from_lower_coo(n, np.concatenate(rows), np.concatenate(cols),
                                      np.concatenate(vals))


def box_stencil3d(m: int, r: int):
    """Lower-stored radius-r box stencil on m^3 (fat supernodes; nd24k stand-in)."""
    n = m * m * m
    p = np.arange(n, dtype=np.int64)
    x = p % m
    y = (p // m) % m
    z = p // (m * m)
    deg = np.zeros(n, dtype=np.int64)
    rows, cols = [], []
    for dz in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                if dx == 0 and dy == 0 and dz == 0:
                    continue
                ok = ((x + dx >= 0) & (x + dx < m) & (y + dy >= 0) & (y + dy < m)
                      & (z + dz >= 0) & (z + dz < m))
                deg += ok
                off = dx + m * (dy + m * dz)
                if off > 0:
                    q = p[ok]
                    rows.append(q + off)
                    cols.append(q)
    rows.append(p)
    cols.append(p)
    nb = sum(a.size for a in rows) - n
    vals = np.concatenate([np.full(nb, -1.0), (deg + 1).astype(np.float64)])
    return (n,) + _csc_from_lower_coo(n, np.concatenate(rows), np.concatenate(cols), vals)


def geometric_nd(mx: int, my: int = 1, mz: int = 1, leaf: int = 4, width: int = 1) -> np.ndarray:
    """Nested-dissection permutation of an mx*my*mz grid.

    nd(box): if all extents <= leaf emit the points lexicographically (x
    fastest); else cut the longest extent (ties x, y, z) at c = lo + extent//2,
    recurse on [lo,c), then [c+width,hi), then emit the slab [c,c+width) last.
    width = 1 is the survey's definition (7-point / 5-point stencils); a radius-r
    box stencil needs width = r for the slab to be a separator.
    Returns Perm with Perm[k] = original index of the k-th pivot.
    """
    out = np.empty(mx * my * mz, dtype=np.int64)
    pos = 0
    # explicit stack of ("box", bounds) / ("emit", bounds) work items
    stack = [(0, (0, mx, 0, my, 0, mz))]
    while stack:
        kind, (x0, x1, y0, y1, z0, z1) = stack.pop()
        ex, ey, ez = x1 - x0, y1 - y0, z1 - z0
        if ex <= 0 or ey <= 0 or ez <= 0:
            continue
        if kind == 1 or (ex <= leaf and ey <= leaf and ez <= leaf):
            zz, yy, xx = np.meshgrid(np.arange(z0, z1), np.arange(y0, y1),
                                     np.arange(x0, x1), indexing="ij")
            idx = (xx + mx * (yy + my * zz)).ravel()
            out[pos:pos + idx.size] = idx
            pos += idx.size
            continue
        w = width
        if ex >= ey and ex >= ez:
            c = x0 + ex // 2
            parts = [(0, (x0, c, y0, y1, z0, z1)), (0, (c + w, x1, y0, y1, z0, z1)),
                     (1, (c, min(c + w, x1), y0, y1, z0, z1))]
        elif ey >= ez:
            c = y0 + ey // 2
            parts = [(0, (x0, x1, y0, c, z0, z1)), (0, (x0, x1, c + w, y1, z0, z1)),
                     (1, (x0, x1, c, min(c + w, y1), z0, z1))]
        else:
            c = z0 + ez // 2
            parts = [(0, (x0, x1, y0, y1, z0, c)), (0, (x0, x1, y0, y1, c + w, z1)),
                     (1, (x0, x1, y0, y1, c, min(c + w, z1)))]
        stack.extend(reversed(parts))
    assert pos == out.size
    return out


def poisson_logdet(*dims: int) -> float:
    """log det of the Dirichlet 5-/7-point Poisson matrix on a grid with the given
    extents (poisson2d / poisson3d above), in closed form: the eigenvalues are
    2*len(dims) - 2*sum_d cos(i_d*pi/(m_d+1)), i_d = 1..m_d.  For a Cholesky factor
    L of any symmetric permutation of A, sum_j 2*log L(j,j) equals this number --
    a whole-factor checksum at sizes no CPU oracle reaches."""
    import math
    lam = np.zeros((1,) * len(dims))
    for ax, m in enumerate(dims):
        c = 2.0 - 2.0 * np.cos(np.arange(1, m + 1) * np.pi / (m + 1))
        shape = [1] * len(dims)
        shape[ax] = m
        lam = lam + c.reshape(shape)
    return math.fsum(np.log(lam).ravel().tolist()) if lam.size <= 2_000_000 else float(np.sum(np.log(lam), dtype=np.longdouble))


def demo_rhs(n: int) -> np.ndarray:
    """b(i) = 1 + i/n, the demo's right-hand side."""
    return 1.0 + np.arange(n, dtype=np.float64) / n


def read_triplet(path):
    """Parse a CHOLMOD triplet / Matrix-Market coordinate file (real, symmetric
    or general) following the format notes of the reference reader
    (CHOLMOD/Check/cholmod_read.c:14-110): '%' comment lines, a header line
    `nrow ncol nnz [stype]`, 1-based unless a zero index appears, duplicates
    summed, Matrix-Market "symmetric" => stype -1.  Symmetric inputs are
    returned with the stored triangle as given (no prefer_upper conversion).
    Returns (n, Ap, Ai, Ax, stype).
    """
    stype = None
    mm_sym = None
    header = None
    ent = []
    with open(path) as f:
        for ln in f:
            t = ln.strip()
            if not t:
                continue
            if t.startswith("%"):
                if t.lower().startswith("%%matrixmarket"):
                    tok = t.lower().split()
                    mm_sym = tok[4] if len(tok) > 4 else "general"
                continue
            tok = t.split()
            if header is None:
                header = [int(float(v)) for v in tok]
                continue
            ent.append(tok)
    nrow, ncol, nnz = header[:3]
    if len(header) > 3:
        stype = header[3]
    elif mm_sym is not None:
        stype = -1 if mm_sym[0] in "sh" and not mm_sym.startswith("sk") else 0
    ii = np.array([int(e[0]) for e in ent[:nnz]], dtype=np.int64)
    jj = np.array([int(e[1]) for e in ent[:nnz]], dtype=np.int64)
    if ent and len(ent[0]) >= 3:
        vv = np.array([float(e[2]) for e in ent[:nnz]], dtype=np.float64)
    else:
        vv = np.ones(ii.size)
    if ii.size and ii.min() > 0 and jj.min() > 0:
        ii -= 1
        jj -= 1
    if stype is None:
        lo = bool(np.any(ii > jj))
        up = bool(np.any(ii < jj))
        stype = 0 if (nrow != ncol or (lo and up)) else (-1 if lo else (1 if up else -1))
    if stype < 0:
        keep = ii >= jj
    elif stype > 0:
        keep = ii <= jj
    else:
        keep = np.ones(ii.size, dtype=bool)
    ii, jj, vv = ii[keep], jj[keep], vv[keep]
    # sum duplicates
    key = jj * max(nrow, 1) + ii
    order = np.argsort(key, kind="stable")
    key, ii, jj, vv = key[order], ii[order], jj[order], vv[order]
    if key.size:
        first = np.concatenate(([True], key[1:] != key[:-1]))
        grp = np.cumsum(first) - 1
        vsum = np.zeros(int(grp[-1]) + 1)
        np.add.at(vsum, grp, vv)
        ii, jj, vv = ii[first], jj[first], vsum
    Ap = np.zeros(ncol + 1, dtype=np.int64)
    np.cumsum(np.bincount(jj, minlength=ncol), out=Ap[1:])
    return nrow, Ap, ii.astype(np.int64), vv, stype


def sym_matvec(n, Ap, Ai, Ax, stype, x):
    """y = A x for a symmetric matrix with one triangle stored."""
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
    y = np.zeros_like(x, dtype=np.float64)
    np.add.at(y, Ai, Ax * x[cols])
    off = Ai != cols
    if stype != 0:
        np.add.at(y, cols[off], Ax[off] * x[Ai[off]])
    return y


def hermitian_phases(n, Ap, Ai, Ax, seed=0, scale=0.9):
    """A Hermitian positive definite matrix on the pattern of a real, diagonally dominant SPD one
    (lower-stored CSC): every off-diagonal entry is turned by a random phase and shrunk, the
    diagonal keeps dominating.  Returns the complex value array (same Ap, Ai)."""
    rng = np.random.default_rng(seed)
    vals = np.asarray(Ax, dtype=np.complex128).copy()
    cols = np.repeat(np.arange(n), np.diff(Ap))
    off = np.asarray(Ai) != cols
    vals[off] *= scale * np.exp(1j * rng.uniform(0, 2 * np.pi, int(off.sum())))
    return vals


def herm_matvec(n, Ap, Ai, Ax, x):
    """y = A x for a Hermitian matrix given by its lower triangle (CSC)."""
    cols = np.repeat(np.arange(n), np.diff(Ap))
    rows = np.asarray(Ai)
    y = np.zeros(n, dtype=np.complex128)
    np.add.at(y, rows, Ax * x[cols])
    off = rows != cols
    np.add.at(y, cols[off], np.conj(Ax[off]) * x[rows[off]])
    return y
