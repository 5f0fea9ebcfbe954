"""CPU oracle: TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import anything from this package; the product package (fpc_diffrend_amd) never does."""
