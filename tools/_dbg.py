import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cugp_amd.gp as gp
n = 129
rng = np.random.default_rng(n)
M = rng.standard_normal((n, n)); K = M @ M.T + n * np.eye(n)
L = gp.potrf(K)
Lref = np.linalg.cholesky(K)
L00 = Lref[:128, :128]
a = K[128, :128]
xg = L[128, :128]                      # GPU: a L00^-T with its own inverses
# solve for the implied inverse action: x = a T^T ; with T = [[T00,0],[T10',T11]] (trsm uses X0 = a0 T00^T, Z1 = a1 - X0 L10^T, X1 = Z1 T11^T)
T00 = np.linalg.inv(L00[:64, :64]); T11 = np.linalg.inv(L00[64:, 64:]); L10 = L00[64:, :64]
X0 = a[:64] @ T00.T
Z1 = a[64:] - X0 @ L10.T
print("X0 err", np.abs(X0 - xg[:64]).max())
X1 = Z1 @ T11.T
print("X1 err by micro column:", [float(np.abs(X1[16*i:16*i+16] - xg[64+16*i:64+16*i+16]).max()) for i in range(4)])
# hypotheses for T11 micro tiles
def mt(T, i, j): return T[16*i:16*i+16, 16*j:16*j+16]
for name, (i, j) in {"T(5,4)": (1, 0), "T(5,5)": (1, 1), "T(6,4)": (2, 0), "T(6,5)": (2, 1), "T(7,6)": (3, 2)}.items():
    for hyp in ("zero", "L"):
        Th = T11.copy()
        mt(Th, i, j)[:] = 0.0 if hyp == "zero" else mt(L00[64:, 64:], i, j)
        Xh = Z1 @ Th.T
        print(name, hyp, "-> micro col err", [float(np.abs(Xh[16*k:16*k+16] - xg[64+16*k:64+16*k+16]).max()) for k in range(4)])
