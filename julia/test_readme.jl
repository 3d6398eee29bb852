# test_readme.jl — what a maintainer runs on a box with Julia ≥ 1.8, SystemLevelControl.jl and one MI355X:
#     SLS_MI355X_LIB=/path/to/libsls_mi355x.so julia --project julia/test_readme.jl
# The README plant (README.md:43-54) through the MI355X engine.  Checks, in order: the default context, the drop-in call,
# the pattern (Φ ⊆ mask), the cost Σ‖Φ‖² = 893.3262819770 (the oracle's value, SURVEY.md §8c; tests/golden/readme_chain_phi.npz
# holds every value), the closed-loop localisation of README.md:62-72, and — when Ipopt is installed — Φ against the
# reference's own SLS_𝓗₂ to 1e-6.
# NOT EXECUTED in the build container (no Julia there: SURVEY.md §0 F5).
using SparseArrays, LinearAlgebra, Test
using SystemLevelControl
include(joinpath(@__DIR__, "SLSMI355X.jl")); using .SLSMI355X

Nx, Nu = 59, 20
d, T, α = 9, 29, 1.5
A = I + spdiagm(1 => 0.2ones(Nx-1)) - spdiagm(-1 => 0.2ones(Nx-1))
B₁ = sparse(I, Nx, Nx)
B₂ = spdiagm(0 => ones(Nx))[:, vec((1:2) .+ 6*(0:9)')]
P = Plant(A, B₁, B₂)
𝓢ₓ = [           (A .≠ 0)^min(d,  floor(Int, α*(t-1))) .≠ 0 for t = 1:T]
𝓢ᵤ = [(B₂' .≠ 0)*(A .≠ 0)^min(d+1,floor(Int, α*(t-1))) .≠ 0 for t = 1:T]

@testset "README chain on the MI355X engine" begin
    ctx = default_ctx()
    @test ctx != C_NULL && ctx === default_ctx()
    status = Int32[]
    Φₓ, Φᵤ = SLS_𝓗₂_mi355x(ctx, P, [𝓢ₓ, 𝓢ᵤ]; status = status)
    @test length(Φₓ) == T && length(Φᵤ) == T && all(==(0), status) && length(status) == Nx
    @test all(t -> all(findall(!iszero, Φₓ[t]) .∈ Ref(Set(findall(𝓢ₓ[t])))), 1:T)          # pattern ⊆ mask
    cost = sum(sum(abs2, Φₓ[t]) + sum(abs2, Φᵤ[t]) for t = 1:T)
    @test cost ≈ 893.3262819770 atol = 1e-7
    @test sum(sum(abs2, Φₓ[t][:, 1]) + sum(abs2, Φᵤ[t][:, 1]) for t = 1:T) ≈ 1.739859 atol = 1e-5
    # achievability (README.md:31): Φx[1] = I, Φx[t+1] = AΦx[t] + B₂Φu[t], AΦx[T] + B₂Φu[T] = 0
    @test norm(Φₓ[1] - I, Inf) < 1e-12
    @test maximum(norm(Φₓ[t+1] - (A*Φₓ[t] + B₂*Φᵤ[t]), Inf) for t = 1:T-1) < 1e-11
    @test norm(A*Φₓ[T] + B₂*Φᵤ[T], Inf) < 1e-11
    # closed loop (README.md:62-72): impulse at state 30, t = 50: response confined to |i−30| ≤ 9 and dead after T steps
    w(t) = (t == 50) * I(Nx)[:, 30]
    x = spzeros(Nx, 250); β = similar(x); u = spzeros(Nu, 250)
    for t = 1:249
        β[:, t+1] = sum([Φₓ[τ+1]*(x[:, t+1-τ] - β[:, t+1-τ]) for τ = 1:min(t, T-1)])
        u[:, t]   = sum([Φᵤ[τ]  *(x[:, t+1-τ] - β[:, t+1-τ]) for τ = 1:min(t, T)])
        x[:, t+1] = A*x[:, t] + B₁*w(t) + B₂*u[:, t]
    end
    r, c = findnz(abs.(x) .> 1e-9)[1:2]
    @test minimum(r) ≥ 21 && maximum(r) ≤ 39 && minimum(c) == 51 && maximum(c) ≤ 79
    # the one-argument form uses the default context
    Φ2ₓ, _ = SLS_𝓗₂_mi355x(P, [𝓢ₓ, 𝓢ᵤ])
    @test all(Φ2ₓ[t] == Φₓ[t] for t = 1:T)
    # against the reference itself (needs Ipopt): the north-star contract ‖Φ − Φ_ref‖∞ < 1e-6
    if Base.find_package("Ipopt") !== nothing
        Rₓ, Rᵤ = SLS_𝓗₂(P, [𝓢ₓ, 𝓢ᵤ])
        @test maximum(norm(Φₓ[t] - Rₓ[t], Inf) for t = 1:T) < 1e-6
        @test maximum(norm(Φᵤ[t] - Rᵤ[t], Inf) for t = 1:T) < 1e-6
    end
end
