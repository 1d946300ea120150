"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the corenav-GP slip-GP hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (corenav_gp_amd/) never imports it and fails loudly without its HIP library.
"""
