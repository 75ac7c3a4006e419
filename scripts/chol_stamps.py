"""Stamps of the one-launch factorisation at the metric map's shape (n = 1194, banded + bordered plan): needs a library built with
-DMCP_CP_PROF (scripts/build_variants.sh cpprof4 / cpabl1), selected through MCP_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from mcptam_amd import chain_bundle as cb
from gpu_chol2 import spd
A, b = spd(1194, band=6)
tf, tb, x = cb.chol_time(np.tril(A), b, nsys=1, reps=30, band=6)
print("n=1194 band=6: factor %.1f us  back %.1f us" % (tf * 1e3, tb * 1e3))
