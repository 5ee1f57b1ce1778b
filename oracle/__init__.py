"""CPU oracle for the SpinTorque-v0 step path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(see oracle/stg_oracle.h).  The product package never does.
"""
from .oracle import *  # noqa: F401,F403
