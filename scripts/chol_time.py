import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcptam_amd.chain_bundle import dense_spd_solve
n = 1194
rng = np.random.default_rng(0)
B = rng.normal(size=(n, n)); A = B @ B.T + n*np.eye(n); b = rng.normal(size=n)
for i in range(3):
    try: dense_spd_solve(A, b)
    except Exception as e: pass
