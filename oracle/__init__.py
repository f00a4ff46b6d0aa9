"""CPU oracle for the k-SLAM alignment hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (k-slam_amd/) never does.  See kslam_oracle.h.
"""
from .binding import *  # noqa: F401,F403
