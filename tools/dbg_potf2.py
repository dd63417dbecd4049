import sys, numpy as np
sys.path.insert(0, '/root/repo')
import cugp_amd.gp as gp
rng = np.random.default_rng(128)
n = 128
M = rng.standard_normal((n, n)); K = M @ M.T + n * np.eye(n)
L = gp.potrf(K); Lo = np.linalg.cholesky(K)
E = np.abs(L - Lo)
np.set_printoptions(precision=1, linewidth=250)
print("max err per column of rows 16..127, cols 0..15:", E[16:, :16].max(axis=0))
print("max err per row (rows 16..63), cols 0..15:", E[16:64, :16].max(axis=1))
A = K[16:32, :16]
print("row 16 got :", L[16, :16]); print("row 16 want:", Lo[16, :16])
