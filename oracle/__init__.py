"""CPU oracle for the quorum hot path — TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package (see quorum_oracle.c).  The product (raft_amd/) never does.
"""
from .pyoracle import *  # noqa: F401,F403
