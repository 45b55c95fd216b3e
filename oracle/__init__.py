"""CPU oracle for the gym_fishing hot path -- test infrastructure, not product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from this package.
"""
