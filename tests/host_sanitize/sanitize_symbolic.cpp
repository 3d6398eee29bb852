// Host symbolic pass (csrc/sls_symbolic.cpp — no HIP in it) under AddressSanitizer + UndefinedBehaviorSanitizer: test infrastructure,
// built and run by tests/test_host.py::test_symbolic_pass_under_sanitizers with g++ -fsanitize=address,undefined.  Drives every
// host-only entry the C ABI forwards to — mask recipe, validation, index sets, the full symbolic pass in its four table layouts
// (explicit / compact × packed / mask order), shard ranges, caller groups (decoupled and coupled), cost model, the inputs of
// the two device passes, the closed-loop FIR operator — on a chain, a 2-D grid and a random plant, in both index bases, plus
// malformed inputs that must be refused without touching memory out of bounds.  Exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../systemlevelcontrol.jl_amd/csrc/sls_symbolic.h"

namespace {

struct Csc {
  int64_t nr = 0, nc = 0;
  std::vector<int64_t> cp, ri;
  std::vector<double> v;
  std::vector<uint8_t> b;
  sls_csc_f64 f64() const { return sls_csc_f64{nr, nc, cp.data(), ri.data(), v.data()}; }
  sls_csc_bool boolean() const { return sls_csc_bool{nr, nc, cp.data(), ri.data(), b.empty() ? nullptr : b.data()}; }
};

// dense pattern → CSC in the given index base
Csc from_dense(const std::vector<std::vector<double>>& M, int64_t nr, int64_t nc, int base) {
  Csc m; m.nr = nr; m.nc = nc; m.cp.assign(nc + 1, base);
  for (int64_t c = 0; c < nc; ++c) {
    for (int64_t r = 0; r < nr; ++r)
      if (M[r][c] != 0.0) { m.ri.push_back(r + base); m.v.push_back(M[r][c]); }
    m.cp[c + 1] = (int64_t)m.ri.size() + base;
  }
  return m;
}

struct Plant { int64_t Nx, Nu; Csc A, B1, B2; };

Plant chain(int64_t Nx, int base, bool coupled_b1) {
  std::vector<std::vector<double>> A(Nx, std::vector<double>(Nx, 0.0)), B1(Nx, std::vector<double>(Nx, 0.0));
  for (int64_t i = 0; i < Nx; ++i) {
    A[i][i] = 1.0; B1[i][i] = 1.0 + 0.01 * i;
    if (i + 1 < Nx) { A[i][i + 1] = 0.2; A[i + 1][i] = -0.2; if (coupled_b1 && i % 4 == 0) B1[i][i + 1] = 0.3; }
  }
  const int64_t Nu = (Nx + 2) / 3;
  std::vector<std::vector<double>> B2(Nx, std::vector<double>(Nu, 0.0));
  for (int64_t j = 0; j < Nu; ++j) B2[std::min(Nx - 1, 3 * j)][j] = 1.0;
  return Plant{Nx, Nu, from_dense(A, Nx, Nx, base), from_dense(B1, Nx, Nx, base), from_dense(B2, Nx, Nu, base)};
}

Plant grid(int64_t n, int base) {
  const int64_t Nx = n * n;
  std::vector<std::vector<double>> A(Nx, std::vector<double>(Nx, 0.0)), B1(Nx, std::vector<double>(Nx, 0.0));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) {
      const int64_t k = i * n + j;
      A[k][k] = 0.9; B1[k][k] = 1.0;
      if (i + 1 < n) { A[k][k + n] = 0.1; A[k + n][k] = 0.1; }
      if (j + 1 < n) { A[k][k + 1] = 0.1; A[k + 1][k] = 0.1; }
    }
  const int64_t Nu = (Nx + 1) / 2;
  std::vector<std::vector<double>> B2(Nx, std::vector<double>(Nu, 0.0));
  for (int64_t j = 0; j < Nu; ++j) B2[std::min(Nx - 1, 2 * j)][j] = 1.0;
  return Plant{Nx, Nu, from_dense(A, Nx, Nx, base), from_dense(B1, Nx, Nx, base), from_dense(B2, Nx, Nu, base)};
}

Plant random_plant(int64_t Nx, int base, unsigned seed) {
  std::mt19937 g(seed);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::vector<std::vector<double>> A(Nx, std::vector<double>(Nx, 0.0)), B1(Nx, std::vector<double>(Nx, 0.0));
  for (int64_t i = 0; i < Nx; ++i) {
    A[i][i] = 1.0; B1[i][i] = 0.5 + u(g);
    for (int64_t j = 0; j < Nx; ++j) if (i != j && u(g) < 0.05) A[i][j] = u(g) - 0.5;
  }
  Plant P{Nx, 0, from_dense(A, Nx, Nx, base), from_dense(B1, Nx, Nx, base), {}};
  // stored zeros in A: (A .≠ 0) is by value
  for (size_t k = 0; k < P.A.v.size(); k += 7) if (P.A.ri[k] - base != (int64_t)0) P.A.v[k] = (k % 14 == 0) ? 0.0 : P.A.v[k];
  const int64_t Nu = Nx / 2;
  std::vector<std::vector<double>> B2(Nx, std::vector<double>(Nu, 0.0));
  for (int64_t j = 0; j < Nu; ++j) { B2[2 * j][j] = 1.0; if (u(g) < 0.3) B2[(2 * j + 5) % Nx][j] = 0.5; }
  P.Nu = Nu; P.B2 = from_dense(B2, Nx, Nu, base);
  return P;
}

int fails = 0;
#define EXPECT(cond, what) do { if (!(cond)) { std::fprintf(stderr, "FAILED: %s (%s:%d)\n", what, __FILE__, __LINE__); ++fails; } } while (0)

struct Masks { std::vector<Csc> x, u; std::vector<sls_csc_bool> bx, bu; };

Masks make_masks(const Plant& P, int64_t d, int64_t T, double alpha, int base) {
  sls_dims dims{P.Nx, P.Nu, P.Nx + P.Nu, P.Nx, T, base, 0};
  sls_csc_f64 A = P.A.f64(), B2 = P.B2.f64();
  std::vector<int64_t> nx(T), nu(T);
  std::string msg;
  int rc = sls::localization_masks(&dims, &A, &B2, d, alpha, nx.data(), nu.data(), nullptr, nullptr, nullptr, nullptr, msg);
  EXPECT(rc == 0, "mask recipe (count)");
  Masks M; M.x.resize(T); M.u.resize(T);
  std::vector<int64_t*> cpx(T), rvx(T), cpu(T), rvu(T);
  for (int64_t t = 0; t < T; ++t) {
    M.x[t].nr = P.Nx; M.x[t].nc = P.Nx; M.x[t].cp.assign(P.Nx + 1, 0); M.x[t].ri.assign(nx[t], 0);
    M.u[t].nr = P.Nu; M.u[t].nc = P.Nx; M.u[t].cp.assign(P.Nx + 1, 0); M.u[t].ri.assign(nu[t], 0);
    cpx[t] = M.x[t].cp.data(); rvx[t] = M.x[t].ri.data(); cpu[t] = M.u[t].cp.data(); rvu[t] = M.u[t].ri.data();
  }
  rc = sls::localization_masks(&dims, &A, &B2, d, alpha, nx.data(), nu.data(), cpx.data(), rvx.data(), cpu.data(), rvu.data(), msg);
  EXPECT(rc == 0, "mask recipe (fill)");
  for (int64_t t = 0; t < T; ++t) { M.bx.push_back(M.x[t].boolean()); M.bu.push_back(M.u[t].boolean()); }
  return M;
}

void exercise(const Plant& P, int64_t d, int64_t T, int base, bool irregular, const char* name) {
  Masks M = make_masks(P, d, T, 1.5, base);
  if (irregular) {                                    // a stored-false entry: forces the explicit tables (no compact layout)
    Csc& m = M.x[T - 1];
    m.b.assign(m.ri.size(), 1);
    if (m.b.size() > 3) m.b[3] = 0;
    M.bx[T - 1] = m.boolean();
  }
  sls_dims dims{P.Nx, P.Nu, P.Nx + P.Nu, P.Nx, T, base, 0};
  sls_csc_f64 A = P.A.f64(), B1 = P.B1.f64(), B2 = P.B2.f64();
  sls_plant plant{&A, &B1, &B2, nullptr, nullptr, nullptr};
  std::string msg;
  // caller groups: pairs of neighbours, a singleton, a triple
  std::vector<int64_t> gptr{0}, gcols;
  for (int64_t c = 0; c + 1 < P.Nx; c += 5) {
    gcols.push_back(c + base); if (c % 10 == 0) gcols.push_back(c + 1 + base);
    if (c % 15 == 0 && c + 2 < P.Nx) gcols.push_back(c + 2 + base);
    gptr.push_back((int64_t)gcols.size());
  }
  const int64_t ng = (int64_t)gptr.size() - 1;
  for (int variant = 0; variant < 2; ++variant) {
    sls::Inputs in{&dims, &plant, M.bx.data(), M.bu.data(), variant ? ng : 0, variant ? gptr.data() : nullptr, variant ? gcols.data() : nullptr};
    int rc = sls::validate_inputs(in, msg);
    EXPECT(rc == 0, "validate_inputs");
    if (rc) { std::fprintf(stderr, "  %s: %s\n", name, msg.c_str()); return; }
    const int64_t n_groups = variant ? ng : P.Nx;
    std::vector<double> cost;
    EXPECT(sls::group_costs(in, cost, msg) == 0 && (int64_t)cost.size() == n_groups, "group_costs");
    for (int packed = 0; packed < 2; ++packed)
      for (int compact = 0; compact < 2; ++compact) {
        const int64_t cuts[4] = {0, n_groups / 3, n_groups / 3, n_groups};      // an empty shard in the middle
        int64_t total_packed = 0, nvals = -1;
        for (int s = 0; s < 3; ++s) {
          sls::Symbolic S; S.want_packed = packed; S.compact = compact && !packed;
          rc = sls::build_symbolic(in, cuts[s], cuts[s + 1], S, msg);
          if (rc == SLS_EUNSUPPORTED) continue;                                  // e.g. a coupled group beyond the kernels' limits
          EXPECT(rc == 0, "build_symbolic");
          if (rc) { std::fprintf(stderr, "  %s: %s\n", name, msg.c_str()); continue; }
          total_packed += S.n_packed;
          EXPECT(nvals < 0 || nvals == S.n_values, "n_values differs between shards");
          nvals = S.n_values;
          if (packed) {
            EXPECT((int64_t)S.packed_to_final.size() == S.n_packed, "packed_to_final length");
            for (int64_t k = 0; k < S.n_packed; ++k) EXPECT(S.packed_to_final[k] >= 0 && S.packed_to_final[k] < S.n_values, "packed_to_final range");
          }
          if (!S.compact)
            for (size_t k = 0; k < S.dest_pool.size(); ++k) EXPECT(S.dest_pool[k] >= -1 && S.dest_pool[k] < S.n_values, "dest range");
        }
        (void)total_packed;
      }
  }
  // inputs of the device passes
  std::vector<int32_t> kx, ku, a_cp, a_ri, b_rp, b_ci, sx_cp, sx_ri, su_cp, su_ri;
  int kmax = 0;
  EXPECT(sls::mask_recipe_inputs(&dims, &A, &B2, d, 1.5, kx, ku, kmax, a_cp, a_ri, b_rp, b_ci, msg) == 0, "mask_recipe_inputs");
  EXPECT(sls::index_set_inputs(&dims, &A, &M.bx[T - 1], &M.bu[T - 1], a_cp, a_ri, sx_cp, sx_ri, su_cp, su_ri, msg) == 0, "index_set_inputs");
  // closed-loop operator
  sls::FirOperator F;
  EXPECT(sls::build_fir_operator(&dims, &A, &B1, &B2, M.bx.data(), M.bu.data(), F, msg) == 0, "build_fir_operator");
  for (int32_t p : F.perm) EXPECT(p >= 0 && p < F.n_values, "FIR perm range");
  // malformed inputs must be refused, not read out of bounds
  {
    Csc bad = P.A; if (!bad.ri.empty()) bad.ri[0] = P.Nx + 5 + base;             // row out of range
    sls_csc_f64 Ab = bad.f64(); sls_plant pb{&Ab, &B1, &B2, nullptr, nullptr, nullptr};
    sls::Inputs in{&dims, &pb, M.bx.data(), M.bu.data(), 0, nullptr, nullptr};
    EXPECT(sls::validate_inputs(in, msg) != 0, "row index out of range accepted");
    Csc bad2 = P.A; bad2.cp[P.Nx] = (int64_t)bad2.ri.size() + base + 3;           // colptr past the arrays: caught by monotone/nnz checks?
    bad2.cp[P.Nx] = bad2.cp[P.Nx - 1] - 1;                                        // non-monotone
    sls_csc_f64 Ab2 = bad2.f64(); sls_plant pb2{&Ab2, &B1, &B2, nullptr, nullptr, nullptr};
    sls::Inputs in2{&dims, &pb2, M.bx.data(), M.bu.data(), 0, nullptr, nullptr};
    EXPECT(sls::validate_inputs(in2, msg) != 0, "non-monotone colptr accepted");
    std::vector<int64_t> gp{0, 2}, gc{(int64_t)base, (int64_t)base};             // a column twice in a group
    sls::Inputs in3{&dims, &plant, M.bx.data(), M.bu.data(), 1, gp.data(), gc.data()};
    EXPECT(sls::validate_inputs(in3, msg) != 0, "duplicate column accepted");
    std::vector<int64_t> gc2{(int64_t)(P.Nx + base), (int64_t)(P.Nx + 1 + base)};   // columns out of range
    sls::Inputs in4{&dims, &plant, M.bx.data(), M.bu.data(), 1, gp.data(), gc2.data()};
    EXPECT(sls::validate_inputs(in4, msg) != 0, "column out of range accepted");
  }
  std::printf("%s base=%d irregular=%d: done\n", name, base, (int)irregular);
}

}  // namespace

int main() {
  for (int base = 0; base < 2; ++base) {
    exercise(chain(59, base, false), 9, 29, base, false, "chain59");
    exercise(chain(40, base, true), 4, 10, base, false, "chain40_coupled_B1");
    exercise(chain(30, base, false), 3, 8, base, true, "chain30_irregular");
    exercise(grid(9, base), 2, 6, base, false, "grid9");
    exercise(random_plant(70, base, 11u + base), 2, 7, base, false, "random70");
  }
  // worker pool: a pass large enough for several threads (Nx/256 ≥ 2)
  exercise(chain(600, 0, false), 6, 12, 0, false, "chain600");
  if (fails) { std::fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  std::printf("sanitize_symbolic: clean\n");
  return 0;
}
