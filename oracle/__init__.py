"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference hot path (see oracle/tpnet_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (tpnet_amd/) never imports it and has no CPU fallback.
"""
