import sys, os; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
import numpy as np, scipy.sparse as sp, slc_amd as slc
from conftest import flat_phi
g = np.load('/root/repo/tests/golden/coupled_group_phi.npz')
Nx = int(g["Nx"]); Pc = slc.workloads.chain_plant(Nx); Nu = Pc.Nu
W = sp.csc_matrix((g["dense_W_data"], g["dense_W_indices"], g["dense_W_indptr"]), shape=(Nx + Nu, Nx + Nu))
B1 = sp.csc_matrix((g["B1_data"], g["B1_indices"], g["B1_indptr"]), shape=(Nx, Nx))
D11 = sp.csc_matrix((g["D11_data"], g["D11_indices"], g["D11_indptr"]), shape=(Nx + Nu, Nx))
P = slc.Plant(Pc.A, B1, Pc.B2, W[:, :Nx], D11, W[:, Nx:])
S = list(slc.workloads.localization_masks(P.A, P.B2, int(g["d"]), int(g["T"]), float(g["alpha"])))
gp = g["group_ptr"]; gc = g["group_cols"]
groups = [[int(c) for c in gc[gp[i]:gp[i + 1]]] for i in range(len(gp) - 1)]
want = np.concatenate([g["dense_vals_x"], g["dense_vals_u"]])
for devs in ([0], [0, 0], [0, 0, 0]):
    ctx = slc.Context(devs)
    Px, Pu, info = slc.SLS_H2(P, S, groups, ctx=ctx, return_info=True, dropzeros=False)
    got = np.concatenate([flat_phi(Px, S[0]), flat_phi(Pu, S[1])])
    print(devs, "coupled groups: max err %.1e" % np.abs(got - want).max(), "status", np.bincount(info["col_status"]).tolist())
    Px, Pu, info = slc.SLS_H2(Pc, S, ctx=ctx, return_info=True, dropzeros=False, objective="sum_of_norms")
    g1 = np.concatenate([flat_phi(Px, S[0]), flat_phi(Pu, S[1])])
    if len(devs) == 1: ref = g1
    print(devs, "sum of norms: max diff vs 1 device %.1e" % np.abs(g1 - ref).max(), "status", np.bincount(info["col_status"]).tolist())
    ctx.close()
