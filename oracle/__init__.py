"""CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/aof_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package, and only as the checker."""
