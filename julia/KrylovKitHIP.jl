# KrylovKitHIP.jl -- the reference-side binding of libkrylov_hip.so.
#
# This is the shim a KrylovKit.jl maintainer (or user) adds next to KrylovKit: a device vector
# type + a device operator type that satisfy the L1 protocol (VectorInterface verbs, `apply`),
# plus method overloads of the L2/L3 entry points for that type which forward to the FUSED
# C-ABI calls.  KrylovKit's own eigsolve / linsolve / svdsolve (L5), its algorithm structs
# (L4) and its small dense LAPACK work (L0) run unchanged on the host.
#
# NOTE: the build image has no Julia toolchain, so this file has not been executed here; the
# identical ccall sequence is exercised through the ctypes mirror
# (krylovkit.jl_amd/krylovkit_hip) by tests/test_gpu_parity.py.  Signatures are exactly those of
# include/krylov_hip.h.
module KrylovKitHIP

using KrylovKit, VectorInterface, LinearAlgebra, SparseArrays
import KrylovKit: apply, apply_normal, apply_adjoint, expand!, initialize, shrink!, basis,
                  OrthonormalBasis, LanczosIterator, LanczosFactorization, ArnoldiIterator,
                  ArnoldiFactorization, GKLIterator, GKLFactorization, orthogonalize!!,
                  project!!, unproject!!, rank1update!, basistransform!,
                  ClassicalGramSchmidt, ModifiedGramSchmidt, ClassicalGramSchmidt2,
                  ModifiedGramSchmidt2, ClassicalGramSchmidtIR, ModifiedGramSchmidtIR

const lib = "libkrylov_hip"

# ---------------------------------------------------------------- error handling
# mirrors chklapackerror (src/dense/linalg.jl:447): integer status -> Julia exception
function chk(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:kk_last_error, lib), Cstring, ()))
    status == -2 && throw(DimensionMismatch(msg))
    status == -5 && throw(ArgumentError(msg))          # "initial vector should not have norm zero"
    error("libkrylov_hip error $status: $msg")
end

# ---------------------------------------------------------------- handles
mutable struct HipContext
    h::Ptr{Cvoid}
    function HipContext(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}()
        chk(ccall((:kk_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        finalizer(c -> ccall((:kk_ctx_destroy, lib), Cint, (Ptr{Cvoid},), c.h), new(r[]))
    end
end

"Contiguous HBM slab of `capacity` columns: replaces the Vector{T} inside OrthonormalBasis{T}."
mutable struct HipSlab
    h::Ptr{Cvoid}
    n::Int
    capacity::Int
    ctx::HipContext
    function HipSlab(ctx::HipContext, n::Integer, capacity::Integer)
        r = Ref{Ptr{Cvoid}}()
        chk(ccall((:kk_basis_create, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ref{Ptr{Cvoid}}), ctx.h, n, capacity, r))
        finalizer(s -> ccall((:kk_basis_free, lib), Cint, (Ptr{Cvoid},), s.h), new(r[], n, capacity, ctx))
    end
end

"The device vector type T of OrthonormalBasis{T}: one column of a slab."
struct HipVec
    slab::HipSlab
    col::Cint            # 0-based column
end

"Device sparse operator built from Julia's SparseMatrixCSC{Float64,Int64} arrays as they are."
mutable struct HipOperator
    h::Ptr{Cvoid}
    size::Tuple{Int,Int}
    function HipOperator(ctx::HipContext, A::SparseMatrixCSC{Float64,Int64}; symmetric::Bool = issymmetric(A))
        r = Ref{Ptr{Cvoid}}()
        chk(ccall((:kk_csc_create, lib), Cint,
                  (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Ref{Ptr{Cvoid}}),
                  ctx.h, size(A, 1), size(A, 2), nnz(A), A.colptr, A.rowval, A.nzval, 1, symmetric ? 1 : 0, r))
        finalizer(o -> ccall((:kk_op_free, lib), Cint, (Ptr{Cvoid},), o.h), new(r[], size(A)))
    end
end

orthcode(::ClassicalGramSchmidt) = (Cint(0), 0.0)
orthcode(::ModifiedGramSchmidt) = (Cint(1), 0.0)
orthcode(::ClassicalGramSchmidt2) = (Cint(2), 0.0)
orthcode(::ModifiedGramSchmidt2) = (Cint(3), 0.0)
orthcode(o::ClassicalGramSchmidtIR) = (Cint(4), Float64(o.η))
orthcode(o::ModifiedGramSchmidtIR) = (Cint(5), Float64(o.η))

# ---------------------------------------------------------------- L1: VectorInterface verbs
# (SURVEY.md Appendix B; the un-fused fallback that makes EVERY KrylovKit algorithm run)
VectorInterface.scalartype(::Type{HipVec}) = Float64
function VectorInterface.inner(x::HipVec, y::HipVec)
    r = Ref{Float64}()
    chk(ccall((:kk_vec_dot, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ref{Float64}), x.slab.h, x.col, y.slab.h, y.col, r))
    return r[]
end
function LinearAlgebra.norm(x::HipVec)
    r = Ref{Float64}()
    chk(ccall((:kk_vec_nrm2, lib), Cint, (Ptr{Cvoid}, Cint, Ref{Float64}), x.slab.h, x.col, r))
    return r[]
end
function VectorInterface.add!!(y::HipVec, x::HipVec, α::Number = 1, β::Number = 1)
    chk(ccall((:kk_vec_axpby, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64, Float64), y.slab.h, y.col, x.slab.h, x.col, α, β))
    return y
end
function VectorInterface.scale!!(x::HipVec, α::Number)
    chk(ccall((:kk_vec_scal, lib), Cint, (Ptr{Cvoid}, Cint, Float64), x.slab.h, x.col, α))
    return x
end
function VectorInterface.scale!!(y::HipVec, x::HipVec, α::Number)
    chk(ccall((:kk_vec_copy_scal, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64), y.slab.h, y.col, x.slab.h, x.col, α))
    return y
end
# ---- host <-> device, fresh vectors
function upload!(x::HipVec, h::Vector{Float64})
    length(h) == x.slab.n || throw(DimensionMismatch())
    chk(ccall((:kk_basis_upload, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), x.slab.h, x.col, h))
    return x
end
function download(x::HipVec)
    h = Vector{Float64}(undef, x.slab.n)
    chk(ccall((:kk_basis_download, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), x.slab.h, x.col, h))
    return h
end
# Fresh vectors (`scale(x, α)`, `zerovector(x)`, the result of `apply`) are columns of a per-context
# scratch slab handed out from a free list; a finalizer-bearing wrapper returns the column.
mutable struct ScratchPool
    slab::HipSlab
    free::Vector{Cint}
end
const POOLS = IdDict{HipContext,ScratchPool}()
function scratch_like(x::HipVec; ncols::Int = 8)
    pool = get!(POOLS, x.slab.ctx) do
        ScratchPool(HipSlab(x.slab.ctx, x.slab.n, ncols), collect(Cint(0):Cint(ncols - 1)))
    end
    isempty(pool.free) && error("KrylovKitHIP: scratch pool exhausted (raise ncols)")
    return HipVec(pool.slab, pop!(pool.free))
end
release!(x::HipVec) = (p = get(POOLS, x.slab.ctx, nothing); p !== nothing && p.slab === x.slab && push!(p.free, x.col); nothing)
function VectorInterface.zerovector(x::HipVec, ::Type{Float64} = Float64)
    y = scratch_like(x)
    chk(ccall((:kk_vec_zero, lib), Cint, (Ptr{Cvoid}, Cint), y.slab.h, y.col))
    return y
end
VectorInterface.zerovector!!(x::HipVec) = (chk(ccall((:kk_vec_zero, lib), Cint, (Ptr{Cvoid}, Cint), x.slab.h, x.col)); x)
VectorInterface.scale(x::HipVec, α::Number) = scale!!(scratch_like(x), x, α)

# ---------------------------------------------------------------- L1: operator protocol (src/apply.jl:1-19)
function apply(A::HipOperator, x::HipVec, y::HipVec = scratch_like(x); transpose::Bool = false)
    chk(ccall((:kk_spmv, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), A.h, transpose, x.slab.h, x.col, y.slab.h, y.col))
    return y
end
apply_normal(A::HipOperator, x::HipVec) = apply(A, x)
apply_adjoint(A::HipOperator, x::HipVec) = apply(A, x; transpose = true)

# ---------------------------------------------------------------- L2: orthonormal.jl entry points
# project!! (orthonormal.jl:88-118), unproject!! (:132-196), orthogonalize!! (:378-452),
# basistransform! (:291-354), rmul!(b, Givens / Householder) (dense/givens.jl, reflector.jl:143-154)
slab_range(b::OrthonormalBasis{HipVec}) = (first(b).slab, first(b).col, Cint(length(b)))   # columns are contiguous by construction

function orthogonalize!!(w::HipVec, b::OrthonormalBasis{HipVec}, x::AbstractVector, alg::KrylovKit.Orthogonalizer)
    slab, c0, m = slab_range(b)
    code, η = orthcode(alg)
    xs = Vector{Float64}(undef, m)
    chk(ccall((:kk_orthogonalize, lib), Cint,
              (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Cint, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Cint}),
              slab.h, c0, m, w.slab.h, w.col, code, η, xs, C_NULL, C_NULL))
    copyto!(x, xs)
    return (w, x)
end
function project!!(y::AbstractVector, b::OrthonormalBasis{HipVec}, x::HipVec, α::Number = true, β::Number = false,
                   r = Base.OneTo(length(b)))
    slab, c0, _ = slab_range(b)
    length(y) == length(r) || throw(DimensionMismatch())
    ys = Vector{Float64}(y)
    chk(ccall((:kk_project, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Float64, Float64, Ptr{Float64}),
              slab.h, c0 + first(r) - 1, length(r), x.slab.h, x.col, α, β, ys))
    copyto!(y, ys)
    return y
end
function unproject!!(y::HipVec, b::OrthonormalBasis{HipVec}, x::AbstractVector, α::Number = true, β::Number = false,
                     r = Base.OneTo(length(b)))
    slab, c0, _ = slab_range(b)
    length(x) == length(r) || throw(DimensionMismatch())
    chk(ccall((:kk_unproject, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Float64, Float64),
              y.slab.h, y.col, slab.h, c0 + first(r) - 1, length(r), Vector{Float64}(x), α, β))
    return y
end
function rank1update!(b::OrthonormalBasis{HipVec}, y::HipVec, x::AbstractVector, α::Number = true, β::Number = true,
                      r = Base.OneTo(length(b)))
    slab, c0, _ = slab_range(b)
    chk(ccall((:kk_rank1update, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, Float64),
              slab.h, c0 + first(r) - 1, length(r), y.slab.h, y.col, Vector{Float64}(x), α, β))
    return b
end
function LinearAlgebra.rmul!(b::OrthonormalBasis{HipVec}, H::KrylovKit.Householder)   # dense/reflector.jl:143-154
    iszero(H.β) && return b
    slab, c0, _ = slab_range(b)
    chk(ccall((:kk_householder_rmul, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Float64),
              slab.h, c0 + first(H.r) - 1, length(H.r), Vector{Float64}(H.v), H.β))
    return b
end
function basistransform!(b::OrthonormalBasis{HipVec}, U::AbstractMatrix)
    slab, c0, m = slab_range(b)
    Ud = Matrix{Float64}(U)
    size(Ud, 1) == m || throw(DimensionMismatch())
    chk(ccall((:kk_basistransform, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Cint), slab.h, c0, m, size(Ud, 2), Ud, m))
    return b
end
function LinearAlgebra.rmul!(b::OrthonormalBasis{HipVec}, G::LinearAlgebra.Givens)
    slab, c0, _ = slab_range(b)
    chk(ccall((:kk_givens_rmul, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Float64, Float64), slab.h, c0 + G.i1 - 1, c0 + G.i2 - 1, G.c, G.s))
    return b
end

# ---------------------------------------------------------------- L3: initialize / shrink!
# initialize(iter::LanczosIterator) (factorizations/lanczos.jl:180-222): x0 is column 0 of a slab with
# krylovdim + 2 columns; on return column 0 = v1, column 1 = r.
function initialize(iter::LanczosIterator{HipOperator,HipVec}; verbosity::Int = 0)
    x0 = iter.x₀
    code, η = orthcode(iter.orth)
    α, β = Ref{Float64}(), Ref{Float64}()
    chk(ccall((:kk_lanczos_initialize, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Float64, Ref{Float64}, Ref{Float64}),
              iter.operator.h, x0.slab.h, x0.col, code, η, α, β))
    V = OrthonormalBasis([HipVec(x0.slab, x0.col)])
    return LanczosFactorization(1, V, [α[]], [β[]], HipVec(x0.slab, x0.col + 1))
end
# shrink!(state, k) (lanczos.jl:273-291): the vectors stay where they are; only the residual is rescaled
function shrink!(state::LanczosFactorization{HipVec}, k; verbosity::Int = 0)
    length(state) <= k && return state
    V = state.V
    while length(V) > k + 1
        pop!(V)
    end
    r = pop!(V)
    resize!(state.αs, k); resize!(state.βs, k)
    state.k = k
    state.r = scale!!(r, KrylovKit.normres(state))
    return state
end

# ---------------------------------------------------------------- L3: fused expand! (the hot path)
# Lanczos: factorizations/lanczos.jl:250-272.  One ccall = scale + SpMV(+three-term tail, alpha)
# + projection pass + update (+ norm): one host sync.
function expand!(iter::LanczosIterator{HipOperator,HipVec}, state::LanczosFactorization; verbosity::Int = 0)
    βold = KrylovKit.normres(state)
    V = state.V
    slab, c0, k = slab_range(V)
    code, η = orthcode(iter.orth)
    α, β, np = Ref{Float64}(), Ref{Float64}(), Ref{Cint}()
    chk(ccall((:kk_lanczos_expand, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ref{Float64}, Ref{Float64}, Ref{Cint}),
              iter.operator.h, slab.h, c0, k, code, η, βold, α, β, np))
    push!(V, HipVec(slab, c0 + k))             # V = push!(V, scale!!(r, 1/βold))      lanczos.jl:257
    push!(state.αs, α[]); push!(state.βs, β[])  #                                       :261-262
    state.k += 1
    state.r = HipVec(slab, c0 + k + 1)
    return state
end

# Arnoldi: factorizations/arnoldi.jl:199-219 (GMRES: linsolve/gmres.jl:59)
function expand!(iter::ArnoldiIterator{HipOperator,HipVec}, state::ArnoldiFactorization; verbosity::Int = 0)
    state.k += 1
    k = state.k
    V, H = state.V, state.H
    slab, c0, kk = slab_range(V)
    code, η = orthcode(iter.orth)
    β = KrylovKit.normres(state)
    m = length(H)
    resize!(H, m + k + 1)
    βn, np = Ref{Float64}(), Ref{Cint}()
    chk(ccall((:kk_arnoldi_expand, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ptr{Float64}, Ref{Float64}, Ref{Cint}),
              iter.operator.h, slab.h, c0, kk, code, η, β, pointer(H, m + 1), βn, np))
    H[m + k + 1] = βn[]
    push!(V, HipVec(slab, c0 + kk))
    state.r = HipVec(slab, c0 + kk + 1)
    return state
end

# GKL: factorizations/gkl.jl:246-269
function expand!(iter::GKLIterator{HipOperator,HipVec}, state::GKLFactorization; verbosity::Int = 0)
    βold = KrylovKit.normres(state)
    U, V = state.U, state.V
    su, _, k = slab_range(U)
    sv, _, _ = slab_range(V)
    code, η = orthcode(iter.orth)
    α, β, pv, pu = Ref{Float64}(), Ref{Float64}(), Ref{Cint}(), Ref{Cint}()
    chk(ccall((:kk_gkl_expand, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Float64, Float64, Ref{Float64}, Ref{Float64}, Ref{Cint}, Ref{Cint}),
              iter.operator.h, su.h, sv.h, k, code, η, βold, α, β, pv, pu))
    push!(U, HipVec(su, k)); push!(V, HipVec(sv, k))
    push!(state.αs, α[]); push!(state.βs, β[])
    state.k += 1
    state.r = HipVec(su, k + 1)
    return state
end

# ---------------------------------------------------------------- BlockLanczos (factorizations/blocklanczos.jl)
# A Block{HipVec} holds consecutive columns of one slab.
block_range(B::KrylovKit.Block{HipVec}) = (first(B.vec).slab, first(B.vec).col, Cint(length(B)))
function KrylovKit.block_inner(B₁::KrylovKit.Block{HipVec}, B₂::KrylovKit.Block{HipVec})
    s1, c1, p = block_range(B₁); s2, c2, q = block_range(B₂)
    M = Matrix{Float64}(undef, p, q)
    chk(ccall((:kk_block_inner, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Cint),
              s1.h, c1, p, s2.h, c2, q, M, p))
    return M
end
function KrylovKit.block_qr!(block::KrylovKit.Block{HipVec}, tol::Real)
    s, c, p = block_range(block)
    R = zeros(p, p); good = Vector{Cint}(undef, p); ng = Ref{Cint}(); drift = Ref{Cint}()
    chk(ccall((:kk_block_qr, lib), Cint,
              (Ptr{Cvoid}, Cint, Cint, Cint, Float64, Ptr{Float64}, Cint, Ptr{Cint}, Ref{Cint}, Ref{Cint}),
              s.h, c, p, c, tol, R, p, good, ng, drift))
    gi = Int.(good[1:ng[]]) .+ 1
    return R[1:ng[], :], gi, drift[] != 0      # NB: the good vectors are compacted to the first ng columns
end
function KrylovKit.block_reorthogonalize!(R::KrylovKit.Block{HipVec}, V::OrthonormalBasis{HipVec})
    sv, c0, m = slab_range(V); sr, cr, q = block_range(R)
    sv === sr || error("block and basis must share a slab")
    chk(ccall((:kk_block_reorthogonalize, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cint), sv.h, c0, m, cr, q))
    return R
end
function apply(A::HipOperator, X::KrylovKit.Block{HipVec})
    s, c, nb = block_range(X)
    Y = KrylovKit.Block([scratch_like(X[1]) for _ in 1:nb])            # contiguous by construction of the pool
    sy, cy, _ = block_range(Y)
    chk(ccall((:kk_block_apply, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint), A.h, s.h, c, sy.h, cy, nb))
    return Y
end

# ---------------------------------------------------------------- short-recurrence solvers (SURVEY 8(f)-3)
# The loop bodies of linsolve(::CG) and linsolve(::BiCGStab) as fused device calls.  KrylovKit's own methods work
# unchanged through the L1 verbs above; these specialisations replace the 6-10 verb calls of an iteration by one or
# two library calls that keep the recurrence scalars on the device.  The surrounding control flow (tolerance checks,
# explicit residual on convergence, ConvergenceInfo) is the reference's, see linsolve/cg.jl:60-101 and
# linsolve/bicgstab.jl:118-199; krylovkit_hip/linsolve.py holds the executed mirror of exactly this sequence.

# one CG iteration: [p = r + beta p]; q = a0 p + a1 A p; alpha = rho/<p,q>; x += alpha p; r -= alpha q; returns |r|
function cg_iterate!(A::HipOperator, slab::HipSlab, cx, cr, cp, cq, a0, a1, beta, first::Bool, rho)
    pq = Ref{Float64}(); nr = Ref{Float64}()
    chk(ccall((:kk_cg_iterate, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Float64, Float64, Float64, Cint, Float64, Ref{Float64}, Ref{Float64}),
              A.h, slab.h, cx, cr, cp, cq, a0, a1, beta, first, rho, pq, nr))
    return nr[]
end

# BiCGStab half steps; cols = (x, r, r_shadow, p, v, s, t, p_prev, v_prev) are columns of one slab
function bicgstab_half!(A::HipOperator, slab::HipSlab, cols::NTuple{9,Cint}, a0, a1, mode::Integer, rho)
    sn = Ref{Float64}(); al = Ref{Float64}()
    c = collect(cols)
    chk(ccall((:kk_bicgstab_half, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Float64, Float64, Cint, Float64, Ref{Float64}, Ref{Float64}),
              A.h, slab.h, c, a0, a1, mode, rho, sn, al))
    return sn[], al[]
end
function bicgstab_full!(A::HipOperator, slab::HipSlab, cols::NTuple{9,Cint}, a0, a1, redo_t::Bool,
                        ahead::Union{Nothing,NTuple{9,Cint}})
    rn = Ref{Float64}(); rho = Ref{Float64}(); om = Ref{Float64}()
    c = collect(cols)
    a = ahead === nothing ? Cint[] : collect(ahead)
    GC.@preserve a begin
        pa = isempty(a) ? Ptr{Cint}(C_NULL) : pointer(a)
        chk(ccall((:kk_bicgstab_full, lib), Cint,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Float64, Float64, Cint, Ptr{Cint}, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                  A.h, slab.h, c, a0, a1, redo_t, pa, rn, rho, om))
    end
    return rn[], rho[], om[]
end

# LSMR vector updates (lssolve/lsmr.jl:64-68, 115-121)
function lsmr_step_u!(slab::HipSlab, c_av, c_ah, c_u, c, alpha)
    beta = Ref{Float64}()
    chk(ccall((:kk_lsmr_step_u, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Float64, Float64, Ref{Float64}),
              slab.h, c_av, c_ah, c_u, c, alpha, beta))
    return beta[]
end
function lsmr_update!(slab::HipSlab, ch, chbar, cx, vslab::Union{Nothing,HipSlab}, cv, c1, c2, c3)
    chk(ccall((:kk_lsmr_update, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Cint, Float64, Float64, Float64),
              slab.h, ch, chbar, cx, vslab === nothing ? C_NULL : vslab.h, cv, c1, c2, c3))
    return nothing
end

end # module
