# SLSMI355X.jl — the reference-side binding for libsls_mi355x.so (include/sls_mi355x.h).
#
# What a maintainer of aaltoKEPO/SystemLevelControl.jl adds to route `SLS_𝓗₂` through the MI355X engine:
# this file is pure `ccall` marshalling (no algorithm), ≈100 lines.  It replaces the body of
# src/synthesis.jl:11-32 (the @distributed loop over _SLS_𝓗₂) and nothing else; `Plant`, the masks and the
# returned `Φₓ,Φᵤ` (vectors of SparseMatrixCSC{Float64,Int}) are unchanged, so README.md:43-72 runs as is.
#
# NOT EXECUTED in the build container (no Julia there: SURVEY.md §0 F5).  The identical marshalling — same struct
# layouts, same argument order, index_base = 0 instead of 1 — is exercised by the Python ctypes binding
# (systemlevelcontrol.jl_amd/_capi.py) in tests/.
module SLSMI355X

using SparseArrays
export SLS_𝓗₂_mi355x, SLS_𝓗₂_mi355x_localized, sls_context, sls_close, default_ctx, sls_ridge!

const LIB = get(ENV, "SLS_MI355X_LIB", "libsls_mi355x.so")

struct CscF64            # sls_csc_f64
    nrows::Int64; ncols::Int64
    colptr::Ptr{Int64}; rowval::Ptr{Int64}; nzval::Ptr{Float64}
end
struct CscBool           # sls_csc_bool  (Julia Bool is one byte: 0x00 / 0x01)
    nrows::Int64; ncols::Int64
    colptr::Ptr{Int64}; rowval::Ptr{Int64}; nzval::Ptr{UInt8}
end
struct Dims              # sls_dims
    Nx::Int64; Nu::Int64; Nz::Int64; Nw::Int64; T::Int64
    index_base::Int32; flags::UInt32
end
struct PlantPtrs         # sls_plant
    A::Ptr{CscF64}; B1::Ptr{CscF64}; B2::Ptr{CscF64}; C1::Ptr{CscF64}; D11::Ptr{CscF64}; D12::Ptr{CscF64}
end

csc(M::SparseMatrixCSC{Float64,Int}) = CscF64(size(M,1), size(M,2), pointer(M.colptr), pointer(M.rowval), pointer(M.nzval))
csc(M::SparseMatrixCSC{Bool,Int})    = CscBool(size(M,1), size(M,2), pointer(M.colptr), pointer(M.rowval),
                                               Ptr{UInt8}(pointer(M.nzval)))

"One context = the GPUs used, the analogue of the `julia -p N` worker pool (src/synthesis.jl:16)."
function sls_context(devices::Vector{<:Integer}=[0])
    devs = Int32.(devices)
    ctx = ccall((:sls_create, LIB), Ptr{Cvoid}, (Ptr{Int32}, Cint, UInt32), devs, length(devs), 0)
    ctx == C_NULL && error(unsafe_string(ccall((:sls_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
    return ctx
end
sls_close(ctx) = ccall((:sls_destroy, LIB), Cvoid, (Ptr{Cvoid},), ctx)

# Lazily created process-wide context (what the one-line patch of src/synthesis.jl:11 in INTEGRATION.md calls): the devices
# come from ENV["SLS_MI355X_DEVICES"] ("0,1,2,3"; default "0"); it is destroyed at exit.
const _default_ctx = Ref{Ptr{Cvoid}}(C_NULL)
function default_ctx()
    if _default_ctx[] == C_NULL
        devs = parse.(Int, split(get(ENV, "SLS_MI355X_DEVICES", "0"), ","))
        _default_ctx[] = sls_context(devs)
        atexit() do
            _default_ctx[] != C_NULL && (sls_close(_default_ctx[]); _default_ctx[] = C_NULL)
        end
    end
    return _default_ctx[]
end

"""
    sls_ridge!(ctx, rₓ, rᵤ)

The diagonal quadratic instance of the reference's `L⁺` hook (src/synthesis.jl:21,52): every later solve on `ctx` adds
`Σₜ Σᵢ rₓ[i]·Φₓ[t][i,c]² + Σⱼ rᵤ[j]·Φᵤ[t][j,c]²` to each column's cost.  `sls_ridge!(ctx, Float64[], Float64[])` clears it.
"""
function sls_ridge!(ctx::Ptr{Cvoid}, rₓ::Vector{Float64}, rᵤ::Vector{Float64})
    rc = ccall((:sls_set_ridge, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ptr{Float64}),
               ctx, length(rₓ), rₓ, length(rᵤ), rᵤ)
    rc < 0 && error(unsafe_string(ccall((:sls_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)))
    return nothing
end

"""
    Φₓ,Φᵤ = SLS_𝓗₂_mi355x(ctx, P, [𝓢ₓ,𝓢ᵤ]; 𝓘=nothing, objective=:h2)

`objective = :sum_of_norms` minimises, per column, `Σₜ‖[C̃₁ D̃₁₂]Φ̃[t]B̃₁‖₂` instead (the column-separable bound of the 𝓗∞ norm;
`SLS_SOLVE_SUM_OF_NORMS` — not in the reference, which has no 𝓗∞ synthesis).

Drop-in for `SLS_𝓗₂(P, 𝓢; 𝓘)` (src/synthesis.jl:11).  `P` is any state-feedback plant exposing the reference's
fields (`A,B₁,B₂,C₁,D₁₁,D₁₂,Nx,Nu,Nz,Nw`); anything else returns `nothing`, like the reference (src/synthesis.jl:13,30).
"""
SLS_𝓗₂_mi355x(P, 𝓢::AbstractVector; kw...) = SLS_𝓗₂_mi355x(default_ctx(), P, 𝓢; kw...)
function SLS_𝓗₂_mi355x(ctx::Ptr{Cvoid}, P, 𝓢::AbstractVector; 𝓘=nothing, status::Union{Nothing,Vector{Int32}}=nothing,
                        objective::Symbol=:h2)
    hasproperty(P, :C₂) && size(P.D₂₁, 1) == 0 || return nothing          # StateFeedback only
    𝓢ₓ, 𝓢ᵤ = 𝓢
    T = length(𝓢ₓ)
    f64(M) = SparseMatrixCSC{Float64,Int}(M)
    A, B1, B2, C1, D11, D12 = f64(P.A), f64(P.B₁), f64(P.B₂), f64(P.C₁), f64(P.D₁₁), f64(P.D₁₂)
    Sx = [SparseMatrixCSC{Bool,Int}(S) for S in 𝓢ₓ];  Su = [SparseMatrixCSC{Bool,Int}(S) for S in 𝓢ᵤ]
    mats = [csc(A), csc(B1), csc(B2), csc(C1), csc(D11), csc(D12)]
    sx = [csc(S) for S in Sx];  su = [csc(S) for S in Su]
    flags = objective === :sum_of_norms ? UInt32(1) : UInt32(0)               # SLS_SOLVE_SUM_OF_NORMS
    dims = Ref(Dims(P.Nx, P.Nu, P.Nz, P.Nw, T, 1, flags))                  # index_base = 1: Julia's own arrays, no copy
    if 𝓘 === nothing
        ng, gptr, gcols = 0, Ptr{Int64}(C_NULL), Ptr{Int64}(C_NULL);  nsub = P.Nx
        keep = nothing
    else
        gp = Int64[0; cumsum(length.(𝓘))];  gc = Int64.(reduce(vcat, 𝓘))
        ng, gptr, gcols, nsub, keep = length(𝓘), pointer(gp), pointer(gc), length(gc), (gp, gc)
    end
    vx = [zeros(Float64, nnz(S)) for S in Sx];  vu = [zeros(Float64, nnz(S)) for S in Su]
    px = [pointer(v) for v in vx];  pu = [pointer(v) for v in vu]
    st = status === nothing ? zeros(Int32, nsub) : resize!(status, nsub)
    GC.@preserve A B1 B2 C1 D11 D12 Sx Su mats sx su vx vu px pu st keep begin
        pm = pointer(mats)
        plant = Ref(PlantPtrs(pm, pm + sizeof(CscF64), pm + 2sizeof(CscF64), pm + 3sizeof(CscF64),
                              pm + 4sizeof(CscF64), pm + 5sizeof(CscF64)))
        rc = ccall((:sls_h2_sf_solve, LIB), Cint,
                   (Ptr{Cvoid}, Ref{Dims}, Ref{PlantPtrs}, Ptr{CscBool}, Ptr{CscBool}, Int64, Ptr{Int64}, Ptr{Int64},
                    Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int32}, Ptr{Cvoid}),
                   ctx, dims, plant, sx, su, ng, gptr, gcols, px, pu, st, C_NULL)
        rc < 0 && error(unsafe_string(ccall((:sls_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)))
        rc > 0 && @warn "SLS_𝓗₂: $rc column(s) not solved to tolerance (see `status`; the reference never checks Ipopt's status)"
    end
    # values arrive in the masks' own CSC order ⇒ the pattern is the mask's, bit for bit; dropzeros! reproduces what
    # the reference's sparse `+` accumulation does to numerical zeros (src/synthesis.jl:65-67)
    Φₓ = [dropzeros!(SparseMatrixCSC(P.Nx, P.Nx, copy(S.colptr), copy(S.rowval), v)) for (S, v) in zip(Sx, vx)]
    Φᵤ = [dropzeros!(SparseMatrixCSC(P.Nu, P.Nx, copy(S.colptr), copy(S.rowval), v)) for (S, v) in zip(Su, vu)]
    return Φₓ, Φᵤ
end

"""
    Φₓ,Φᵤ = SLS_𝓗₂_mi355x_localized(ctx, P, d, α, T)

The same solve for the README's own masks (README.md:52-54) given as `(d, α, T)`: `sls_h2_sf_solve_localized` builds index sets,
mask slices and destinations on the device — no mask array crosses PCIe.  The patterns of the returned matrices are the README
recipe's, computed here in Julia only to wrap the value arrays (a caller that keeps Φ on the device does not need them).
Default plant weights and default groups only (the library says so otherwise).
"""
function SLS_𝓗₂_mi355x_localized(ctx::Ptr{Cvoid}, P, d::Integer, α::Real, T::Integer)
    hasproperty(P, :C₂) && size(P.D₂₁, 1) == 0 || return nothing
    f64(M) = SparseMatrixCSC{Float64,Int}(M)
    A, B1, B2 = f64(P.A), f64(P.B₁), f64(P.B₂)
    Ab, Bb = (A .≠ 0), (SparseMatrixCSC(B2') .≠ 0)
    Sx = [SparseMatrixCSC{Bool,Int}((Ab^min(d, floor(Int, α * (t - 1)))) .≠ 0) for t in 1:T]
    Su = [SparseMatrixCSC{Bool,Int}((Bb * Ab^min(d + 1, floor(Int, α * (t - 1)))) .≠ 0) for t in 1:T]
    mats = [csc(A), csc(B1), csc(B2)]
    dims = Ref(Dims(P.Nx, P.Nu, P.Nz, P.Nw, T, 1, UInt32(0)))
    vx = [zeros(Float64, nnz(S)) for S in Sx];  vu = [zeros(Float64, nnz(S)) for S in Su]
    px = [pointer(v) for v in vx];  pu = [pointer(v) for v in vu]
    st = zeros(Int32, P.Nx)
    GC.@preserve A B1 B2 mats vx vu px pu st begin
        pm = pointer(mats)
        plant = Ref(PlantPtrs(pm, pm + sizeof(CscF64), pm + 2sizeof(CscF64), Ptr{CscF64}(C_NULL), Ptr{CscF64}(C_NULL), Ptr{CscF64}(C_NULL)))
        rc = ccall((:sls_h2_sf_solve_localized, LIB), Cint,
                   (Ptr{Cvoid}, Ref{Dims}, Ref{PlantPtrs}, Int64, Cdouble, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int32}, Ptr{Cvoid}),
                   ctx, dims, plant, d, α, px, pu, st, C_NULL)
        rc < 0 && error(unsafe_string(ccall((:sls_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx)))
        rc > 0 && @warn "SLS_𝓗₂: $rc column(s) not solved to tolerance"
    end
    Φₓ = [dropzeros!(SparseMatrixCSC(P.Nx, P.Nx, copy(S.colptr), copy(S.rowval), v)) for (S, v) in zip(Sx, vx)]
    Φᵤ = [dropzeros!(SparseMatrixCSC(P.Nu, P.Nx, copy(S.colptr), copy(S.rowval), v)) for (S, v) in zip(Su, vu)]
    return Φₓ, Φᵤ
end

end # module
