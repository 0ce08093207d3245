"""CPU oracle for the Snappy block codec hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(see the header of snappy_oracle.c).  The product package snappier_amd never does.
"""
from .pyoracle import *  # noqa: F401,F403
