"""CPU oracle for the KokkosSparse spmv/spgemm hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/kk_oracle.h for what it restates and how it is pinned.
"""
from .kk_oracle import *  # noqa: F401,F403
