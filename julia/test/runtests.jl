# julia/test/runtests.jl -- KrylovKit.jl's own factorization invariants with a `:hip` wrap mode.
#
#   julia --project=julia julia/test/runtests.jl            (needs a gfx950 GPU, KrylovKit.jl, and libkrylov_hip.so on the
#                                                            loader path: export LD_LIBRARY_PATH=$PWD/krylovkit.jl_amd/lib)
#
# The reference tests every factorization under the wrap modes :vector / :inplace / :outplace / :mixed
# (/root/reference/test/testsetup.jl:62-98 -- `wrapvec`, `wrapop`, `unwrapvec`).  This file adds the mode `:hip` next to
# them -- vectors are `HipVec` columns of an HBM slab, the operator a `HipOperator` (kk_op) -- and runs the SAME
# invariants the reference asserts after EVERY `expand!`:
#   Lanczos       test/factorize.jl:140-148   V'V ~ I, norm(r) ~ beta, A V ~ V H + r e'
#   Arnoldi       test/factorize.jl:185-193   same with the Hessenberg H
#   GKL           test/factorize.jl:285-296   U'U ~ I, V'V ~ I, A V ~ U B + r e', A'U ~ V B'
#   BlockLanczos  test/factorize.jl:387-401   V'V ~ I, norm(R) ~ norm_R, A V ~ V H + R e
# plus shrink! / initialize! round trips, the thick-restart eigsolve / GMRES linsolve / svdsolve drivers end to end
# against dense LAPACK, and the issue-#143 matrix (test/issues.jl:39-129) through `eigsolve(A, Block, ...)`.
# All six orthogonalisers, both MGS modes of the library (KK_MGS_MODE=0 strict reference order, 1 low-synchronisation).
#
# NOTE: the build image of this repository has no Julia: this file has never been executed.  It is the harness a
# maintainer runs the day Julia and a GPU are on the same machine; tests/test_julia_shim_lint.py keeps every `ccall` and
# every overloaded method signature of ../KrylovKitHIP.jl in step with include/krylov_hip.h and the reference's methods.
using Test, LinearAlgebra, SparseArrays, Random
using KrylovKit
using KrylovKit: OrthonormalBasis, Block
include(joinpath(@__DIR__, "..", "KrylovKitHIP.jl"))
using .KrylovKitHIP: HipContext, HipOperator, HipVec, download

const ctx = HipContext(0)

# ---- the reference's wrappers, extended by the :hip mode (testsetup.jl:65-98)
wrapvec(v, ::Val{:vector}) = v
wrapvec(v, ::Val{:hip}) = HipVec(ctx, Vector{Float64}(v))
unwrapvec(v::HipVec) = download(v)
unwrapvec(v) = v
wrapop(A, ::Val{:vector}) = A
wrapop(A, ::Val{:hip}) = HipOperator(ctx, SparseMatrixCSC{Float64,Int64}(sparse(A)))

const orths = (ClassicalGramSchmidt(), ModifiedGramSchmidt(), ClassicalGramSchmidt2(), ModifiedGramSchmidt2(),
               ClassicalGramSchmidtIR(), ModifiedGramSchmidtIR())
const modes = (:vector, :hip)        # :vector = the reference's own CPU path on the same inputs, side by side
tolerance(::Type{Float64}) = 1.0e-11
Random.seed!(76543210)
n, N = 10, 100

@testset "Lanczos - factorization ($mode, $orth)" for mode in modes, orth in orths
    A = rand(Float64, (N, N)); A = A + A'
    v = rand(Float64, N)
    iter = LanczosIterator(wrapop(A, Val(mode)), wrapvec(v, Val(mode)), orth)
    fact = initialize(iter)
    while normres(fact) > eps(Float64) && length(fact) < n
        expand!(iter, fact)
        Ṽ, H, r̃, β, e = fact
        V = stack(unwrapvec, Ṽ); r = unwrapvec(r̃)
        @test V' * V ≈ I                                   # factorize.jl:144
        @test norm(r) ≈ β                                  # :145
        @test A * V ≈ V * H + r * e'                       # :146
    end
    fact = shrink!(fact, div(n, 2))                        # :149-155
    V = stack(unwrapvec, basis(fact)); H = rayleighquotient(fact); r = unwrapvec(residual(fact))
    @test V' * V ≈ I
    @test norm(r) ≈ normres(fact)
    @test A * V ≈ V * H + r * rayleighextension(fact)'
end

@testset "Arnoldi - factorization ($mode, $orth)" for mode in modes, orth in orths
    A = rand(Float64, (N, N))
    v = rand(Float64, N)
    iter = ArnoldiIterator(wrapop(A, Val(mode)), wrapvec(v, Val(mode)), orth)
    fact = initialize(iter)
    while normres(fact) > eps(Float64) && length(fact) < 3n
        expand!(iter, fact)
        Ṽ, H, r̃, β, e = fact
        V = stack(unwrapvec, Ṽ); r = unwrapvec(r̃)
        @test V' * V ≈ I                                   # factorize.jl:189
        @test norm(r) ≈ β
        @test A * V ≈ V * H + r * e'
    end
    fact = shrink!(fact, div(n, 2))
    V = stack(unwrapvec, basis(fact)); H = rayleighquotient(fact); r = unwrapvec(residual(fact))
    @test V' * V ≈ I
    @test A * V ≈ V * H + r * rayleighextension(fact)'
end

@testset "GKL - factorization ($mode, $orth)" for mode in modes, orth in orths
    A = rand(Float64, (N, 2N)) .- 0.5
    v = A * rand(Float64, 2N)
    iter = GKLIterator(wrapop(A, Val(mode)), wrapvec(v, Val(mode)), orth)
    fact = initialize(iter)
    while normres(fact) > eps(Float64) && length(fact) < 3n
        expand!(iter, fact)
        Ũ, Ṽ, B, r̃, β, e = fact
        U = stack(unwrapvec, Ũ); V = stack(unwrapvec, Ṽ); r = unwrapvec(r̃)
        @test U' * U ≈ I                                   # factorize.jl:290
        if !(orth isa ClassicalGramSchmidt2)               # CGS2 re-orthogonalises r against U only (gkl.jl:308-323)
            @test V' * V ≈ I
        end
        @test norm(r) ≈ β
        @test A * V ≈ U * B + r * e'
        @test A' * U ≈ V * B'
    end
    fact = shrink!(fact, div(n, 2))
    U = stack(unwrapvec, basis(fact, Val(:U))); V = stack(unwrapvec, basis(fact, Val(:V)))
    B = rayleighquotient(fact); r = unwrapvec(residual(fact))
    @test A * V ≈ U * B + r * rayleighextension(fact)'
    @test A' * U ≈ V * B'
end

@testset "BlockLanczos - factorization ($mode, block size $bs)" for mode in modes, bs in (2, 5, 16)
    A = rand(Float64, (N, N)); A = (A + A') / 2
    x₀ = Block([wrapvec(rand(Float64, N), Val(mode)) for _ in 1:bs])
    iter = BlockLanczosIterator(wrapop(A, Val(mode)), x₀, N, KrylovKit.KrylovDefaults.orth, tolerance(Float64))
    fact = initialize(iter)
    while fact.norm_R > eps(Float64) && fact.k < 3n
        expand!(iter, fact)
        k, rs = fact.k, fact.R_size
        V = hcat([unwrapvec(v) for v in fact.V[1:k]]...)
        r = hcat([unwrapvec(fact.R[i]) for i in 1:rs]...)
        H = fact.H[1:k, 1:k]
        e = hcat(zeros(rs, k - rs), I)
        @test V' * V ≈ I                                   # factorize.jl:398
        @test norm(r) ≈ fact.norm_R
        @test A * V ≈ V * H + r * e
    end
end

@testset "device and host paths agree step by step ($orth)" for orth in orths
    # the parity the C-ABI tests hold the library to (1e-10 relative), here through the Julia shim
    A = sprand(Float64, 400, 400, 0.02); A = A + A' + 10I
    v = rand(Float64, 400)
    it_h = LanczosIterator(A, v, orth); it_d = LanczosIterator(wrapop(A, Val(:hip)), wrapvec(v, Val(:hip)), orth)
    f_h, f_d = initialize(it_h), initialize(it_d)
    for _ in 1:25
        expand!(it_h, f_h); expand!(it_d, f_d)
    end
    rtol = orth isa KrylovKit.Reorthogonalizer ? 1.0e-10 : 1.0e-6
    @test f_d.αs ≈ f_h.αs rtol = rtol
    @test f_d.βs ≈ f_h.βs rtol = rtol
end

@testset "drivers end to end (:hip)" begin
    A = sprand(Float64, 500, 500, 0.02); A = A + A' + Diagonal(range(1.0, 40.0; length = 500))
    v = rand(Float64, 500)
    Ad, vd = wrapop(A, Val(:hip)), wrapvec(v, Val(:hip))
    D, V, info = eigsolve(Ad, vd, 4, :LR, Lanczos(; krylovdim = 30, tol = 1.0e-10, maxiter = 200))
    Dh, Vh, infoh = eigsolve(A, v, 4, :LR, Lanczos(; krylovdim = 30, tol = 1.0e-10, maxiter = 200))
    @test info.converged >= 4
    @test (info.numiter, info.numops) == (infoh.numiter, infoh.numops)     # thick restarts included: same control flow
    @test D[1:4] ≈ Dh[1:4] rtol = 1.0e-10
    @test D[1:4] ≈ eigvals(Symmetric(Matrix(A)))[end:-1:(end - 3)] rtol = 1.0e-9
    B = sprand(Float64, 500, 500, 0.02) + 8I
    b = rand(Float64, 500)
    x, ginfo = linsolve(wrapop(B, Val(:hip)), wrapvec(b, Val(:hip)), GMRES(; krylovdim = 30, maxiter = 30, tol = 1.0e-10 * norm(b)))
    xh, ginfoh = linsolve(B, b, GMRES(; krylovdim = 30, maxiter = 30, tol = 1.0e-10 * norm(b)))
    @test ginfo.converged == 1
    @test (ginfo.numiter, ginfo.numops) == (ginfoh.numiter, ginfoh.numops)   # north_star: equal iteration counts
    @test norm(B * unwrapvec(x) - b) <= 2.0e-10 * norm(b)
    C = sprand(Float64, 600, 250, 0.03)
    u = rand(Float64, 600)
    S, L, R, sinfo = svdsolve(wrapop(C, Val(:hip)), wrapvec(u, Val(:hip)), 4, :LR, GKL(; krylovdim = 20, tol = 1.0e-10, maxiter = 100))
    @test S[1:4] ≈ svdvals(Matrix(C))[1:4] rtol = 1.0e-9
end

@testset "issue 143: rank-deficient start block (:hip)" begin
    # test/issues.jl:39-129: 71 x 71 toric-code matrix, 20 random start vectors -> blocks 20 + 20 + 20 + 11, all 71 eigenvalues
    # the fixture the Python tests use (tests/golden/issue143_A.npy, extracted from test/issues.jl:40-112 by
    # tests/golden/make_golden.py): NPY v1, little-endian Float64, 71 x 71, symmetric (so the storage order is immaterial)
    A143 = let f = joinpath(@__DIR__, "..", "..", "tests", "golden", "issue143_A.npy")
        if isfile(f)
            raw = read(f)
            hlen = Int(reinterpret(UInt16, raw[9:10])[1])
            data = reinterpret(Float64, raw[(11 + hlen):end])
            length(data) == 71 * 71 ? Matrix(reshape(collect(data), 71, 71)) : nothing
        else
            nothing
        end
    end
    if A143 === nothing
        @test_skip "tests/golden/issue143_A.npy not found"
    else
        x₀ = Block([wrapvec(randn(Float64, 71), Val(:hip)) for _ in 1:20])
        D, V, info = eigsolve(wrapop(A143, Val(:hip)), x₀, 4, :SR, BlockLanczos(; tol = 1.0e-8))
        @test length(D) == 71
        @test sort(D) ≈ eigvals(Symmetric(Matrix(A143))) atol = 1.0e-10 * opnorm(Matrix(A143))
        @test info.numiter == 1 && info.numops == 72
    end
end
