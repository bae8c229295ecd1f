"""CPU restatements of the reference's hot path — TEST INFRASTRUCTURE ONLY.

Nothing under proqa_amd/ imports this package.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may use it, and only as the checker / the CPU baseline.
"""
