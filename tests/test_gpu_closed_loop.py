"""GPU tests of the closed-loop simulator (sls_closed_loop_*): the README's simulation script
(reference README.md:62-72) run on the device with the Φ the solve left there, against the oracle's restatement of the
same recursion.  Same arithmetic (FP64 FMA chains), different summation order: tolerance 1e-11 relative to max|x|."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RTOL = 1e-11


def _solve_on_device(slc, ctx, P, S):
    plan = slc.Plan(ctx, P, S)
    d = plan.alloc_values()
    plan.execute(d); plan.synchronize()
    st, _, _ = plan.fetch_status()
    assert np.all(st == 0)
    vx, vu = plan.download(d)
    Phix, Phiu = slc.assemble_phi(S[0], S[1], vx, vu, dropzeros=False)
    return plan, d, Phix, Phiu


def test_readme_impulse_response_matches_oracle(slc, gpu_ctx, readme, oracle):
    """README.md:60-76: w(t) = δ(t−50)·e₃₀, 250 steps; Φ never leaves the device between the solve and the simulation."""
    P, S, _ = readme
    plan, d, Phix, Phiu = _solve_on_device(slc, gpu_ctx, P, S)
    loop = slc.ClosedLoop(gpu_ctx, P, S)
    assert loop.n_entries == sum(m.nnz for m in S[0][1:]) + sum(m.nnz for m in S[1])
    w = np.zeros((250, P.Nw)); w[49, 29] = 1.0                  # row t−1 is w(t)
    x, u = loop.simulate(d, w, steps=250)
    xo, uo = oracle.closed_loop(P.A, P.B1, P.B2, Phix, Phiu)
    assert x.shape == (250, 59) and u.shape == (250, 20)
    # committed fixture (oracle recursion on the golden Φ): tests/golden/readme_closed_loop.npz
    g = np.load(os.path.join(GOLDEN, "readme_closed_loop.npz"))
    t0, w_ = int(g["t0"]), g["x"].shape[1]
    assert np.abs(x[t0:t0 + w_].T - g["x"]).max() < 1e-8 and np.abs(u[t0:t0 + w_].T - g["u"]).max() < 1e-8
    assert np.abs(x[:t0]).max() == 0 and t0 + w_ == 250
    assert np.abs(x.T - xo).max() < RTOL * max(1.0, np.abs(xo).max())
    assert np.abs(u.T - uo).max() < RTOL * max(1.0, np.abs(uo).max())
    # localized in space (|i−30| ≤ 9) and dead T = 29 steps after the impulse
    r, c = np.nonzero(np.abs(x.T) > 1e-9)
    assert r.min() + 1 >= 21 and r.max() + 1 <= 39 and c.min() + 1 == 51 and c.max() + 1 <= 79
    loop.close(); plan.close()


@pytest.mark.parametrize("nscen", [1, 5, 70])
def test_random_disturbances_many_scenarios(slc, gpu_ctx, oracle, nscen):
    """Dense random w on every step, several scenarios in one run (lane layouts SCN = 1, 8, 64 with a ragged last chunk):
    every scenario equals the oracle run on its own w."""
    P = slc.workloads.chain_plant(23)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 6, 18, 1.5))
    plan, d, Phix, Phiu = _solve_on_device(slc, gpu_ctx, P, S)
    loop = slc.ClosedLoop(gpu_ctx, P, S)
    steps = 60
    w = np.random.default_rng(nscen).standard_normal((steps, P.Nw, nscen))
    x, u = loop.simulate(d, w, steps=steps)
    assert x.shape == (steps, P.Nx, nscen) and u.shape == (steps, P.Nu, nscen)
    for s in sorted({0, nscen // 2, nscen - 1}):
        xo, uo = oracle.closed_loop(P.A, P.B1, P.B2, Phix, Phiu, steps=steps, w=w[:, :, s])
        assert np.abs(x[:, :, s].T - xo).max() < RTOL * max(1.0, np.abs(xo).max())
        assert np.abs(u[:, :, s].T - uo).max() < RTOL * max(1.0, np.abs(uo).max())
    loop.close(); plan.close()


def test_arbitrary_operator_shared_and_idle_actuators(slc, gpu_ctx, oracle):
    """The simulator does not assume Φ came from a solve: random masks (stored-false entries included), random values,
    non-square B₁, an actuator that drives two states (evaluated by both row owners) and one that drives none
    (still reported in u).  Device-pointer entry point, run twice so that the second run replays the cached hipGraph."""
    import torch
    rng = np.random.default_rng(7)
    Nx, Nu, Nw, T, steps, nscen = 17, 5, 17, 6, 40, 3
    A = sp.random(Nx, Nx, 0.25, random_state=1, format="csc") * 0.3
    B1 = sp.identity(Nx, format="csc")
    B2 = sp.lil_matrix((Nx, Nu))
    B2[2, 0] = 1.0; B2[9, 0] = -0.5                                # actuator 0 drives two states
    B2[4, 1] = 0.7; B2[11, 2] = 1.2; B2[15, 3] = 0.9               # actuator 4 drives none
    B2 = B2.tocsc()
    P = slc.Plant(A, B1, B2)
    Sx, Su = [], []
    for t in range(T):
        mx = sp.random(Nx, Nx, 0.3, random_state=10 + t, format="csc"); mx.data[:] = 1
        mu = sp.random(Nu, Nx, 0.4, random_state=30 + t, format="csc"); mu.data[:] = 1
        mx = mx.astype(bool).tocsc(); mu = mu.astype(bool).tocsc()
        mx.sort_indices(); mu.sort_indices()
        Sx.append(mx); Su.append(mu)
    Sx[2].data[3] = False                                         # stored false: not part of the operator
    vals_x = [rng.uniform(-0.2, 0.2, m.nnz) * m.data for m in Sx]
    vals_u = [rng.uniform(-0.2, 0.2, m.nnz) * m.data for m in Su]
    Phix, Phiu = slc.assemble_phi(Sx, Su, vals_x, vals_u, dropzeros=False)
    w = rng.standard_normal((steps, Nw, nscen))
    dev = torch.device("cuda:0")
    d_vals = torch.from_numpy(np.concatenate(vals_x + vals_u)).to(dev)
    d_w = torch.from_numpy(w).to(dev)
    d_x = torch.full((steps, Nx, nscen), np.nan, dtype=torch.float64, device=dev)
    d_u = torch.full((steps, Nu, nscen), np.nan, dtype=torch.float64, device=dev)
    loop = slc.ClosedLoop(gpu_ctx, P, [Sx, Su])
    stream = torch.cuda.current_stream(dev).cuda_stream
    for rep in range(2):
        loop.run(d_vals.data_ptr(), d_w.data_ptr(), steps, nscen, d_x.data_ptr(), d_u.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        x, u = d_x.cpu().numpy(), d_u.cpu().numpy()
        assert np.isfinite(x).all() and np.isfinite(u).all()
        for s in range(nscen):
            xo, uo = oracle.closed_loop(P.A, P.B1, P.B2, Phix, Phiu, steps=steps, w=w[:, :, s])
            assert np.abs(x[:, :, s].T - xo).max() < RTOL * max(1.0, np.abs(xo).max())
            assert np.abs(u[:, :, s].T - uo).max() < RTOL * max(1.0, np.abs(uo).max())
        assert np.abs(u[:, 4, :]).max() > 0                        # the idle actuator's command is still computed
        d_x.fill_(float("nan")); d_u.fill_(float("nan"))
    assert loop.last_ms() > 0
    loop.close()


def test_chain1024_disturbance_rejection_full_size(slc, gpu_ctx):
    """Full-size property (1024 states, T = 40, 1.2 M stored Φ entries): impulses on 16 states at once; by linearity and
    shift invariance of the interior of the chain every response is the same localized pulse, gone after T steps."""
    P, S, meta = slc.workloads.make_workload("chain1024")
    plan = slc.Plan(gpu_ctx, P, S)
    d = plan.alloc_values(); plan.execute(d); plan.synchronize()
    loop = slc.ClosedLoop(gpu_ctx, P, S)
    steps, T, dloc = 100, meta["T"], meta["d"]
    centers = 200 + 6 * np.arange(16)                              # same position in the actuator pattern
    w = np.zeros((steps, P.Nw, len(centers)))
    for s, c in enumerate(centers):
        w[9, c, s] = 1.0                                           # w(10) = e_c
    x, u = loop.simulate(d, w, steps=steps)
    assert np.abs(x[10, centers, np.arange(16)] - 1.0).max() < 1e-12   # x(11) = B₁w(10)
    for s, c in enumerate(centers):
        r, i = np.nonzero(np.abs(x[:, :, s]) > 1e-9)
        assert i.min() >= c - dloc and i.max() <= c + dloc and r.min() == 10 and r.max() <= 10 + T - 1
        shift = c - centers[0]
        assert np.abs(x[:, shift:, s] - x[:, : P.Nx - shift, 0]).max() < 1e-10
    # the same run through the host recursion of SciPy mat-vecs on the downloaded Φ (scenario 0 only)
    vx, vu = plan.download(d)
    Phix, Phiu = slc.assemble_phi(S[0], S[1], vx, vu, dropzeros=False)
    what = np.zeros((steps, P.Nx)); xh = np.zeros((steps, P.Nx))
    for k in range(1, steps):
        beta = sum(Phix[tau] @ what[k - tau] for tau in range(1, min(k, T - 1) + 1))
        uk = sum(Phiu[tau - 1] @ what[k - tau] for tau in range(1, min(k, T) + 1))
        xh[k] = P.A @ xh[k - 1] + P.B1 @ w[k - 1, :, 0] + P.B2 @ uk
        what[k] = xh[k] - beta
    assert np.abs(xh - x[:, :, 0]).max() < RTOL * max(1.0, np.abs(xh).max())
    loop.close(); plan.close()
