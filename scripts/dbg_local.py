import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, logging
from scipy import sparse
from conftest import load_golden, make_adata
from spatialcore_amd.spatial import local_morans_i
logging.getLogger("spatialcore_amd").setLevel(logging.ERROR)
f32 = np.float32
g = load_golden("ref_local_morans.npz")
ci = 1
X = g[f"c{ci}_X"]
ad = make_adata(g[f"c{ci}_coords"], X)
local_morans_i(ad, n_neighbors=4, n_permutations=0)
z = ad.obsm["local_morans_z"]
Xs = sparse.csc_matrix(X)
m = np.asarray(Xs.mean(axis=0)).ravel(); q = np.asarray(Xs.power(2).mean(axis=0)).ravel()
sd = np.sqrt(q - m**2)
def ulps(v, k):
    out = [v]
    a = b = v
    for _ in range(k):
        a = np.nextafter(a, f32(np.inf)); b = np.nextafter(b, f32(-np.inf)); out += [a, b]
    return out
for j in range(3):
    col = X[:, j]; lv = np.unique(col)
    idx = [np.where(col == v)[0][0] for v in lv]
    found = []
    for mm in ulps(m[j], 4):
        for ss in ulps(sd[j], 4):
            if all(f32(f32(f32(v) - mm) / ss) == z[i, j] for v, i in zip(lv, idx)):
                found.append((mm, ss))
    print("gene", j, "ref m, sd", m[j], sd[j], "q", q[j], "device-consistent (m, sd):", found[:4])
    for qq in ulps(q[j], 3):
        for mm in ulps(m[j], 3):
            s2 = np.sqrt(f32(qq - f32(mm * mm)))
            if (mm, s2) in found:
                print("    explained by q =", qq, "(ref", q[j], ") m =", mm, "(ref", m[j], ")")
