"""CPU study behind the Anderson acceleration of the sum-of-norms ADMM (csrc/sls_wave_kernel.hip): plain vs AA(5) vs AA(8) step counts with the
kernel's safeguards (restart on a ρ change or a tenfold rise of ‖g‖, regularised normal equations), on chain columns of three sizes."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, slc_amd as slc, sls_oracle as o, sls_son_oracle as son

def setup(Nx, d, T, c):
    P = slc.workloads.chain_plant(Nx)
    S = list(slc.workloads.localization_masks(P.A, P.B2, d, T, 1.5))
    Po = o.OraclePlant(P.A, P.B1, P.B2)
    E, f, w, tslice, info = son._column_problem(Po, c, S[0], S[1])
    Ew = E / w; pinv = np.linalg.pinv(Ew, rcond=1e-11)
    return Ew, f, pinv, tslice

def run(Ew, f, pinv, tslice, tol=1e-9, m=0, maxit=20000, start=20, alpha=1.8, reg=1e-10, guard=10.0):
    n = Ew.shape[1]
    x0 = pinv @ f
    big = max(np.linalg.norm(x0[idx]) for idx in tslice)
    rho = 8.0 / big
    y = x0.copy(); u = np.zeros(n)
    dF = []; dG = []; Fprev = None; gprev = None; gmin = np.inf
    for it in range(1, maxit + 1):
        v0 = y - u
        x = v0 - pinv @ (Ew @ v0 - f)
        xh = alpha * x + (1 - alpha) * y
        v = xh + u
        yn = np.zeros(n)
        for idx in tslice:
            nv = np.linalg.norm(v[idx])
            if nv > 1 / rho: yn[idx] = (1 - 1 / (rho * nv)) * v[idx]
        un = v - yn
        rp = np.linalg.norm(x - yn); rd = rho * np.linalg.norm(yn - y)
        if max(rp, rd) <= tol * max(1, np.linalg.norm(x)): return it, obj(x, tslice)
        Fs = np.concatenate([yn, un]); s = np.concatenate([y, u]); g = Fs - s
        gn = np.linalg.norm(g)
        if it % 10 == 0:
            sc = 2.0 if rp > 10 * rd else (0.5 if rd > 10 * rp else 1.0)
            if sc != 1.0:
                rho *= sc; un = un / sc
                dF, dG, Fprev, gprev, gmin = [], [], None, None, np.inf
                y, u = yn, un; continue
        if m > 0 and it > start:
            if gn > guard * gmin:                     # safeguard: restart
                dF, dG, Fprev, gprev, gmin = [], [], None, None, np.inf
                y, u = yn, un; continue
            gmin = min(gmin, gn)
            if Fprev is not None:
                dF.append(Fs - Fprev); dG.append(g - gprev)
                if len(dF) > m: dF.pop(0); dG.pop(0)
            Fprev, gprev = Fs.copy(), g.copy()
            if dG:
                G = np.array(dG)                       # k × 2n
                A = G @ G.T; b = G @ g
                A += reg * np.trace(A) / len(dG) * np.eye(len(dG))
                gam = np.linalg.solve(A, b)
                sn = Fs - np.array(dF).T @ gam
                y, u = sn[:n], sn[n:]
                continue
        y, u = yn, un
    return maxit, obj(x, tslice)

def obj(x, tslice): return sum(np.linalg.norm(x[idx]) for idx in tslice)

cases = [(80, 12, 40, c) for c in (0, 1, 5, 13, 26, 40, 79)] + [(59, 9, 29, c) for c in (0, 10, 30, 58)] + [(23, 6, 18, c) for c in (0, 11, 22)]
for (Nx, d, T, c) in cases:
    Ew, f, pinv, tslice = setup(Nx, d, T, c)
    r0 = run(Ew, f, pinv, tslice, m=0)
    r5 = run(Ew, f, pinv, tslice, m=5)
    r8 = run(Ew, f, pinv, tslice, m=8)
    print(Nx, d, T, c, "plain", r0[0], "AA5", r5[0], "AA8", r8[0], "obj diff %.1e %.1e" % (r5[1] - r0[1], r8[1] - r0[1]))
