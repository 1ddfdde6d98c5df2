"""CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/oracle.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package."""
from .binding import *  # noqa: F401,F403
