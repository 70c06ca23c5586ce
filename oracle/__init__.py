"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference forward, pinned on reference-generated goldens (tests/golden/).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package s2m2_amd."""
