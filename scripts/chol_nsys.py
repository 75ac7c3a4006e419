"""Factorisation time of the metric shape (n = 1194, band 6) against the number of systems in the launch and the workers per system
(MCP_BA_CHOL_WORKERS): where does the 4-system launch lose its 20 us?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from mcptam_amd import chain_bundle as cb
from gpu_chol2 import spd
A, b = spd(1194, band=6)
for nsys in (1, 2, 3, 4):
    tf, tb, x = cb.chol_time(np.tril(A), b, nsys=nsys, reps=30, band=6)
    print("workers %s nsys %d: factor %.1f us  back %.1f us" % (os.environ.get("MCP_BA_CHOL_WORKERS", "auto"), nsys, tf*1e3, tb*1e3), flush=True)
