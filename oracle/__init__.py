"""ORACLE (test infrastructure, NOT product code).  CPU restatement of the reference's LDE + BLAKE3 Merkle
commitment path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
