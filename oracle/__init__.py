"""CPU oracle for the MVSNet cost-volume path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product (mvs_amd) never does.
"""
