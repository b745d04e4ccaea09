#!/usr/bin/env python3
"""Extracts the literal test fixtures the reference's unit tests hold for the spmv/spgemm
path into .npz files (runs only in the build container; outputs are committed):
  matrix_issue402.npz  <- sparse/unit_test/matrixIssue402.hpp (1813x1813, 11156 nnz; used by
                          test_issue402, sparse/unit_test/Test_Sparse_spgemm.hpp:372-442)
  crs_10x10.npz        <- sparse/unit_test/Test_Sparse_CrsMatrix.hpp:67-76 (10x10, 21 nnz)
"""
import os, re
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/sparse/unit_test/"

def c_array(text, name):
    m = re.search(r"%s\s*\[\s*\d*\s*\]\s*=\s*\{(.*?)\}" % re.escape(name), text, re.S)
    body = re.sub(r"//.*", "", m.group(1))
    return [t for t in re.split(r"[\s,]+", body) if t]

t = open(REF + "matrixIssue402.hpp").read()
vals = np.array([float(v) for v in c_array(t, "values")])
rm = np.array([int(v) for v in c_array(t, "rowmap")], dtype=np.int64)
ent = np.array([int(v) for v in c_array(t, "entries")], dtype=np.int32)
assert len(vals) == 11156 and len(rm) == 1814 and len(ent) == 11156 and rm[-1] == 11156
np.savez_compressed(os.path.join(HERE, "matrix_issue402.npz"), row_map=rm, entries=ent, values=vals)
print("matrix_issue402", len(rm) - 1, len(ent))

t = open(REF + "Test_Sparse_CrsMatrix.hpp").read()
ptr = np.array([int(v) for v in c_array(t, "ptrRaw")], dtype=np.int64)
ind = np.array([int(v) for v in c_array(t, "indRaw")], dtype=np.int32)
val = np.array([float(v) for v in c_array(t, "valRaw")])
assert ptr[-1] == len(ind) == len(val)
np.savez_compressed(os.path.join(HERE, "crs_10x10.npz"), row_map=ptr, entries=ind, values=val)
print("crs_10x10", len(ptr) - 1, len(ind))
