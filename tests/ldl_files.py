"""Reader for the matrices and the recorded output of the reference's LDL demo (LDL/Matrix/A01 .. A30,
LDL/Demo/ldlmain.out; format: LDL/Demo/ldlmain.c:12-24).  Test infrastructure: these are the only files in the reference
that hold EXPECTED symbolic results -- the number of entries of L and the flop count of a factorization under a permutation
stored in the file and under the natural order -- for matrices the supernodal path can take (tests/test_ldl_recorded.py)."""
import re

import numpy as np


def read_ldl_matrix(path):
    """-> dict(name, n, jumbled, Ap, Ai, Ax, P); P is None when the file ends before it."""
    with open(path) as f:
        title = f.readline().strip()
        tok = f.read().split()
    n, jumbled = int(tok[0]), int(tok[1])
    pos = 2
    Ap = np.array([int(t) for t in tok[pos:pos + n + 1]], dtype=np.int64)
    pos += n + 1
    nz = int(Ap[-1]) if n >= 0 and len(Ap) == n + 1 else 0
    nz = max(nz, 0)
    Ai = np.array([int(t) for t in tok[pos:pos + nz]], dtype=np.int64)
    pos += nz
    Ax = np.array([float(t) for t in tok[pos:pos + nz]], dtype=np.float64)
    pos += nz
    P = np.array([int(t) for t in tok[pos:pos + n]], dtype=np.int64) if len(tok) >= pos + n else None
    return {"name": title, "n": n, "jumbled": jumbled, "Ap": Ap, "Ai": Ai, "Ax": Ax, "P": P}


def upper_csc(m):
    """The upper triangle (what LDL_symbolic / LDL_numeric read: entries with i <= k of column k, LDL/Source/ldl.c), duplicates
    summed and columns sorted -- a jumbled file describes the same matrix as its clean twin."""
    import scipy.sparse as sp
    n = m["n"]
    A = sp.csc_matrix((m["Ax"], m["Ai"], m["Ap"]), shape=(n, n))
    A.sum_duplicates()
    A = sp.triu(A, format="csc")
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)


def recorded(path):
    """ldlmain.out -> {"A13": {"name": ..., "given": (nz, flops), "natural": (nz, flops)} | {"invalid": True}}"""
    out, cur = {}, None
    with open(path) as f:
        for line in f:
            m = re.match(r"Input file: \.\./Matrix/(A\d+)", line)
            if m:
                cur = m.group(1)
                out[cur] = {"pairs": []}
                continue
            if cur is None:
                continue
            if line.startswith("name:"):
                out[cur]["name"] = line[5:].strip()
            m = re.match(r"Nz in L: (\d+)\s+Flop count: (\S+)", line)
            if m:
                out[cur]["pairs"].append((int(m.group(1)), float(m.group(2))))
            if "invalid matrix and/or permutation" in line:
                out[cur]["invalid"] = True
    for k, v in out.items():
        p = v.pop("pairs")
        if p:
            v["given"], v["natural"] = p[0], p[1]
    return out
