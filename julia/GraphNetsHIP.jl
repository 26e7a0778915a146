# GraphNetsHIP.jl — thin `ccall` shim over libgnx.so (include/gnx.h) that keeps GraphNets.jl's API for the forward
# hot path: GNBlock(in => out), block(x), GNCore, GNCoreList, batch, unbatch, efview/nfview/gfview, flatunpadded*.
#
# Status: shipped as source.  Julia is not installed in the build image nor on the GPU box, so this file is NOT
# exercised by the test-suite; the tested host mirror of the same ABI is graphnets.jl_amd/api.py.  Host `Array`s are
# staged through `hipMalloc`/`hipMemcpy` (libamdhip64); with AMDGPU.jl, `ROCArray` pointers can be passed straight
# to the `gnx_*` calls instead (the ABI takes raw device pointers).
#
# Layout note: a Julia `Array{Float32,3}` of size (D, T, B) is byte-identical to the ABI's packed [B][T][D] rows, and
# `Dense.weight` (out × in, column-major) is byte-identical to `gnx_dense.weight`; nothing is transposed or copied on
# the host.  The batched tuple is PACKED: `x.ef` is (DE, ΣE, 1) for a vector of graphs instead of the reference's
# padded (DE, PN², B); `unbatch`/views/`flatunpadded*` return what the reference returns.
module GraphNetsHIP

# every name GraphNets.jl exports (src/GraphNets.jl:12-50) ...
export GNGraphBatch, batch, unbatch, getedgefninput, getnodefninput, getgraphfninput, GNBlock, zerodim2nothing, GNCore, GNCoreList,
       efview, nfview, gfview, flatunpaddednf, flatunpaddedef, collapsef, unpaddedcollapsedef, flatunpaddedcollapsedef
# ... plus what the drop-in adds: layers as plain structs, pullbacks, the library-side hipGraph model, the multi-GPU split
export Dense, LayerNorm, ChainBlock, chain_pullback, block_pullback, core_pullback, Model, partition_graphs, DistBlock

const libgnx = get(ENV, "GNX_LIB", joinpath(@__DIR__, "..", "graphnets.jl_amd", "libgnx.so"))
const libhip = get(ENV, "GNX_HIP_LIB", "libamdhip64.so")

# ---- error handling: every gnx_* returns Int32 (0 ok, <0 argument error mirroring an @assert, >0 hipError_t) ----
function check(rc::Integer)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:gnx_last_error, libgnx), Cstring, ()))
    rc < 0 ? throw(AssertionError("gnx $(rc): $(msg)")) : error("gnx HIP error $(rc): $(msg)")
end
hipcheck(rc) = rc == 0 ? nothing : error("HIP error $(rc)")

# ---- tiny device-buffer helper (replace by AMDGPU.ROCArray if available) ----
mutable struct DevBuf
    ptr::Ptr{Cvoid}
    bytes::Int
    function DevBuf(bytes::Integer)
        p = Ref{Ptr{Cvoid}}(C_NULL)
        hipcheck(ccall((:hipMalloc, libhip), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, max(bytes, 16)))
        b = new(p[], bytes)
        finalizer(x -> ccall((:hipFree, libhip), Cint, (Ptr{Cvoid},), x.ptr), b)
        b
    end
end
function upload(a::Array{Float32})
    b = DevBuf(sizeof(a))
    hipcheck(ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), b.ptr, a, sizeof(a), 1))
    b
end
upload(::Nothing) = nothing
function download!(a::Array{Float32}, b::DevBuf)
    hipcheck(ccall((:hipMemcpy, libhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), a, b.ptr, sizeof(a), 2))
    a
end
devptr(b::DevBuf) = Ptr{Cfloat}(b.ptr)
devptr(::Nothing) = Ptr{Cfloat}(C_NULL)

# ---- C structs of include/gnx.h ----
struct GnxDense
    weight::Ptr{Cfloat}; bias::Ptr{Cfloat}; act::Int32; reserved::Int32
end
struct GnxBlockParams
    de::Int32; dn::Int32; dg::Int32; oe::Int32; on::Int32; og::Int32
    edgefn::GnxDense; nodefn::GnxDense; graphfn::GnxDense
end
struct GnxGraphsInfo
    n_graphs::Int64; n_nodes::Int64; n_edges::Int64; node_block_size::Int64; edge_block_size::Int64
    n_tiles::Int64; max_in_degree::Int64; device::Int32; reserved::Int32
end
struct GnxDenseGrad; weight::Ptr{Cfloat}; bias::Ptr{Cfloat}; end
struct GnxBlockGrads; edgefn::GnxDenseGrad; nodefn::GnxDenseGrad; graphfn::GnxDenseGrad; end
const ACT = Dict(identity => 0, :relu => 1, :tanh => 2, :sigmoid => 3, :gelu => 4)

# ---- GNGraphBatch(adj_mats)  (replaces src/gngraphbatch.jl:33-54) ----
mutable struct GNGraphBatch
    handle::Ptr{Cvoid}
    adj_mats::Vector
    node_block_size::Int
    edge_block_size::Int
    node_off::Vector{Int64}   # 0-based offsets of each graph's nodes / edges in the packed arrays
    edge_off::Vector{Int64}
    function GNGraphBatch(adj_mats::AbstractVector)
        @assert length(adj_mats) > 0
        mats = [Matrix{Float32}(a) for a in adj_mats]            # column-major, as Julia stores them
        ptrs = [pointer(m) for m in mats]
        nn = Int64[size(m, 1) for m in mats]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve mats check(ccall((:gnx_graphs_create_dense, libgnx), Int32,
            (Ptr{Ptr{Cvoid}}, Ptr{Int64}, Int64, Int32, Int32, Ptr{Ptr{Cvoid}}),
            ptrs, nn, length(mats), 3 #=GNX_ELEM_F32=#, 0 #=column-major=#, h))
        info = Ref{GnxGraphsInfo}()
        check(ccall((:gnx_graphs_get_info, libgnx), Int32, (Ptr{Cvoid}, Ptr{GnxGraphsInfo}), h[], info))
        no = zeros(Int64, length(mats) + 1); eo = zeros(Int64, length(mats) + 1)
        check(ccall((:gnx_graphs_get_offsets, libgnx), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), h[], no, eo))
        g = new(h[], collect(adj_mats), info[].node_block_size, info[].edge_block_size, no, eo)
        finalizer(x -> ccall((:gnx_graphs_destroy, libgnx), Int32, (Ptr{Cvoid},), x.handle), g)
        g
    end
    # CSC form (API extension: dense N x N matrices cannot hold 100k-node graphs).  colptrs[g] / rowvals[g] are the 1-based `colptr` /
    # `rowval` of a SparseMatrixCSC whose column j lists the sources i of the edges i -> j — its nz order IS the reference's edge order
    # (src/pad.jl:30).  adj_mats keeps whatever the caller passed (the sparse matrices), for unbatch.
    function GNGraphBatch(colptrs::AbstractVector{<:AbstractVector{<:Integer}}, rowvals::AbstractVector{<:AbstractVector{<:Integer}},
                          n_nodes::AbstractVector{<:Integer}; adj_mats::AbstractVector=collect(zip(colptrs, rowvals, n_nodes)))
        @assert length(colptrs) > 0 && length(colptrs) == length(rowvals) == length(n_nodes)
        cps = [Vector{Int64}(c) for c in colptrs]; rvs = [Vector{Int64}(r) for r in rowvals]
        cp_ptrs = [pointer(c) for c in cps]; rv_ptrs = [pointer(r) for r in rvs]
        nn = Int64.(n_nodes)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve cps rvs check(ccall((:gnx_graphs_create_csc, libgnx), Int32,
            (Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Int64}, Int64, Int32, Ptr{Ptr{Cvoid}}),
            cp_ptrs, rv_ptrs, nn, length(cps), 1 #=index_base: Julia=#, h))
        info = Ref{GnxGraphsInfo}()
        check(ccall((:gnx_graphs_get_info, libgnx), Int32, (Ptr{Cvoid}, Ptr{GnxGraphsInfo}), h[], info))
        no = zeros(Int64, length(cps) + 1); eo = zeros(Int64, length(cps) + 1)
        check(ccall((:gnx_graphs_get_offsets, libgnx), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), h[], no, eo))
        g = new(h[], collect(adj_mats), info[].node_block_size, info[].edge_block_size, no, eo)
        finalizer(x -> ccall((:gnx_graphs_destroy, libgnx), Int32, (Ptr{Cvoid},), x.handle), g)
        g
    end
end
# one graph as (colptr, rowval, n); and anything with SparseMatrixCSC's fields (no dependency on SparseArrays: duck-typed)
GNGraphBatch(colptr::AbstractVector{<:Integer}, rowval::AbstractVector{<:Integer}, n::Integer) = GNGraphBatch([colptr], [rowval], [n])
issparsecsc(a) = hasproperty(a, :colptr) && hasproperty(a, :rowval) && hasproperty(a, :n)
sparse_batch(mats::AbstractVector) = GNGraphBatch([m.colptr for m in mats], [m.rowval for m in mats], [m.n for m in mats]; adj_mats=mats)
nnodes(g::GNGraphBatch) = Int(g.node_off[end]); nedges(g::GNGraphBatch) = Int(g.edge_off[end])
ngraphs(g::GNGraphBatch) = length(g.adj_mats)

# ---- batch / unbatch / views  (src/batch.jl:53-64, src/unbatch.jl, src/unpad.jl, src/views.jl) ----
function batch(t::NamedTuple)
    @assert Set(keys(t)) == Set((:graphs, :ef, :nf, :gf))
    (; graphs, ef, nf, gf) = t
    @assert !isnothing(ef) || !isnothing(nf) || !isnothing(gf)
    if graphs isa AbstractMatrix                                  # shared adjacency: ef (DE,E,B), nf (DN,N,B), gf (DG,B)
        g = issparsecsc(graphs) ? sparse_batch([graphs]) : GNGraphBatch([graphs])
        isnothing(ef) || @assert ndims(ef) == 3 && size(ef, 2) == nedges(g) "$(size(ef, 2)) != num_edges"
        isnothing(nf) || @assert ndims(nf) == 3 && size(nf, 2) == nnodes(g)
        isnothing(gf) || @assert ndims(gf) == 2
        return (graphs=g, ef=ef, nf=nf, gf=isnothing(gf) ? nothing : reshape(gf, size(gf, 1), 1, size(gf, 2)))
    end
    g = all(issparsecsc, graphs) ? sparse_batch(graphs) : GNGraphBatch(graphs)   # vector of graphs: pack graph-major
    cat2(v) = isnothing(v) ? nothing : reshape(reduce(hcat, v), size(v[1], 1), :, 1)
    isnothing(ef) || @assert length(ef) == ngraphs(g)
    isnothing(nf) || @assert length(nf) == ngraphs(g)
    bef, bnf = cat2(ef), cat2(nf)
    isnothing(bef) || @assert size(bef, 2) == nedges(g)
    isnothing(bnf) || @assert size(bnf, 2) == nnodes(g)
    (graphs=g, ef=bef, nf=bnf, gf=isnothing(gf) ? nothing : reshape(reduce(hcat, gf), length(gf[1]), :, 1))
end

sharedlike(g::GNGraphBatch) = ngraphs(g) == 1                      # unbatch.jl:15-17
function unbatch(t::NamedTuple)
    (; graphs, ef, nf, gf) = t
    g = graphs
    if sharedlike(g)
        return (graphs=g.adj_mats[1], ef=ef, nf=nf, gf=isnothing(gf) ? nothing : reshape(gf, size(gf, 1), :))
    end
    rng(off, i) = (off[i]+1):off[i+1]
    (graphs=g.adj_mats,
     ef=isnothing(ef) ? nothing : [view(ef, :, rng(g.edge_off, i), 1) for i in 1:ngraphs(g)],
     nf=isnothing(nf) ? nothing : [view(nf, :, rng(g.node_off, i), 1) for i in 1:ngraphs(g)],
     gf=isnothing(gf) ? nothing : [view(gf, :, i, 1) for i in 1:ngraphs(g)])
end
efview(t::NamedTuple, d1, d2, d3) = isnothing(t.ef) ? nothing :
    sharedlike(t.graphs) ? view(t.ef, d1, d2, d3) : view(view(t.ef, :, (t.graphs.edge_off[d3]+1):t.graphs.edge_off[d3+1], 1), d1, d2)
nfview(t::NamedTuple, d1, d2, d3) = isnothing(t.nf) ? nothing :
    sharedlike(t.graphs) ? view(t.nf, d1, d2, d3) : view(view(t.nf, :, (t.graphs.node_off[d3]+1):t.graphs.node_off[d3+1], 1), d1, d2)
gfview(t::NamedTuple, d1, d2) = isnothing(t.gf) ? nothing :
    sharedlike(t.graphs) ? view(t.gf, d1, 1, d2) : view(t.gf, d1, d2, 1)
flatunpaddednf(t::NamedTuple) = reshape(t.nf, size(t.nf, 1), :)       # the packed layout already is it (views.jl:80-88)
flatunpaddedef(t::NamedTuple) = reshape(t.ef, size(t.ef, 1), :)

# ---- the exported building blocks (src/edgefninput.jl:1-47, nodefninput.jl:1-24, graphfninput.jl:1-13) → gnx_fn_input.
#      Same argument order as the reference; `graphs` is the GNGraphBatch of a batched tuple; features are the packed (D, T, R) arrays
#      (`nothing` drops the segment, as the reference's methods do).  Result: (K, T, R) over the real edges / nodes / graphs. ----
function fninput(kind::Integer, g::GNGraphBatch, ef, nf, gf)
    @assert !isnothing(ef) || !isnothing(nf) || !isnothing(gf)
    R = size(something(ef, nf, gf), 3)
    w(a) = isnothing(a) ? 0 : size(a, 1)
    T = (nedges(g), nnodes(g), ngraphs(g))[kind + 1]
    K = w(ef) + (kind == 0 ? 2 : 1) * w(nf) + w(gf)
    out = zeros(Float32, K, T, R)
    d_ef, d_nf, d_gf, b_out = upload(ef), upload(nf), upload(gf), DevBuf(sizeof(out))
    GC.@preserve d_ef d_nf d_gf b_out check(ccall((:gnx_fn_input, libgnx), Int32,
        (Ptr{Cvoid}, Int32, Ptr{Cfloat}, Int32, Ptr{Cfloat}, Int32, Ptr{Cfloat}, Int32, Int64, Ptr{Cfloat}, Ptr{Cvoid}),
        g.handle, kind, devptr(d_ef), w(ef), devptr(d_nf), w(nf), devptr(d_gf), w(gf), R, devptr(b_out), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    download!(out, b_out)
end
getedgefninput(graphs, edge_features, node_features, graph_features) = fninput(0, graphs, edge_features, node_features, graph_features)
getnodefninput(graphs, edge_features, node_features, graph_features) = fninput(1, graphs, edge_features, node_features, graph_features)
getgraphfninput(graphs, edge_features, node_features, graph_features) = fninput(2, graphs, edge_features, node_features, graph_features)

# ---- edge collapsing (src/gngraphbatch.jl:56-111) → gnx_collapse_padded / gnx_collapse_offsets / gnx_collapse_edges ----
function collapsef(t::NamedTuple)                                      # (DE, PN(PN+1)/2, B), padded array form
    g = t.graphs; ef = t.ef
    D, R = size(ef, 1), size(ef, 3)
    B = sharedlike(g) ? R : ngraphs(g)
    PN = g.node_block_size
    out = zeros(Float32, D, PN * (PN + 1) ÷ 2, B)
    d_ef, b_out = upload(ef), DevBuf(sizeof(out))
    GC.@preserve d_ef b_out check(ccall((:gnx_collapse_padded, libgnx), Int32, (Ptr{Cvoid}, Ptr{Cfloat}, Int32, Int64, Ptr{Cfloat}, Ptr{Cvoid}),
        g.handle, devptr(d_ef), D, R, devptr(b_out), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    download!(out, b_out)
end
function collapsed_packed(t::NamedTuple)                               # (DE, total, R) over the real lower-triangle edges + offsets
    g = t.graphs; ef = t.ef
    D, R = size(ef, 1), size(ef, 3)
    off = zeros(Int64, ngraphs(g) + 1)
    check(ccall((:gnx_collapse_offsets, libgnx), Int32, (Ptr{Cvoid}, Ptr{Int64}), g.handle, off))
    out = zeros(Float32, D, Int(off[end]), R)
    d_ef, b_out = upload(ef), DevBuf(max(sizeof(out), 4))
    GC.@preserve d_ef b_out check(ccall((:gnx_collapse_edges, libgnx), Int32, (Ptr{Cvoid}, Ptr{Cfloat}, Int32, Int64, Ptr{Cfloat}, Ptr{Cvoid}),
        g.handle, devptr(d_ef), D, R, devptr(b_out), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    download!(out, b_out), off
end
function unpaddedcollapsedef(t::NamedTuple)                            # gngraphbatch.jl:87-107
    out, off = collapsed_packed(t)
    sharedlike(t.graphs) ? [view(out, :, :, b) for b in 1:size(out, 3)] :
        [view(out, :, (off[i]+1):off[i+1], 1) for i in 1:ngraphs(t.graphs)]
end
flatunpaddedcollapsedef(t::NamedTuple) = reduce(hcat, unpaddedcollapsedef(t))   # gngraphbatch.jl:109-111
zerodim2nothing(t::NamedTuple) = (graphs=t.graphs, ef=t.ef, nf=t.nf, gf=t.gf)  # zero-width outputs are already `nothing`

# ---- layers ----
struct Dense
    weight::Matrix{Float32}; bias::Vector{Float32}; σ
end
glorot(out, in) = (rand(Float32, out, in) .* 2f0 .- 1f0) .* sqrt(6f0 / max(in + out, 1))
Dense(in::Integer, out::Integer, σ=identity) = Dense(glorot(out, in), zeros(Float32, out), σ)
actcode(σ) = get(ACT, σ, get(ACT, Symbol(σ), nothing))

struct GNBlock
    edgefn::Dense; nodefn::Dense; graphfn::Dense; dropout
    in::NTuple{3,Int}; out::NTuple{3,Int}
end
function GNBlock((in, out)::Pair; dropout=0)                          # src/gnblock.jl:47-61
    @assert any(in .> (0, 0, 0)); @assert any(out .> (0, 0, 0))
    (de, dn, dg), (oe, on, og) = in, out
    GNBlock(Dense(de + 2dn + dg, oe), Dense(dn + oe + dg, on), Dense(on + oe + dg, og), dropout, Tuple(in), Tuple(out))
end

function (m::GNBlock)(x)                                              # src/gnblock.jl:63-69 → gnx_block_forward
    (; graphs, ef, nf, gf) = x
    g::GNGraphBatch = graphs
    R = size(something(ef, nf, gf), 3)
    (oe, on, og) = m.out
    W = [upload(m.edgefn.weight), upload(m.nodefn.weight), upload(m.graphfn.weight)]
    B = [upload(m.edgefn.bias), upload(m.nodefn.bias), upload(m.graphfn.bias)]
    mk(d::Dense, w, b) = GnxDense(devptr(w), devptr(b), Int32(actcode(d.σ)), 0)
    p = Ref(GnxBlockParams(m.in..., m.out..., mk(m.edgefn, W[1], B[1]), mk(m.nodefn, W[2], B[2]), mk(m.graphfn, W[3], B[3])))
    d_ef, d_nf, d_gf = upload(ef), upload(nf), upload(gf)
    o_ef, o_nf, o_gf = zeros(Float32, oe, nedges(g), R), zeros(Float32, on, nnodes(g), R), zeros(Float32, og, ngraphs(g), R)
    b_ef, b_nf, b_gf = DevBuf(sizeof(o_ef)), DevBuf(sizeof(o_nf)), DevBuf(sizeof(o_gf))
    wsb = ccall((:gnx_block_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxBlockParams}, Int64), g.handle, p, R)
    ws = DevBuf(wsb)
    GC.@preserve W B d_ef d_nf d_gf b_ef b_nf b_gf ws check(ccall((:gnx_block_forward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{GnxBlockParams}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat},
         Ptr{Cvoid}, Csize_t, UInt32, Ptr{Cvoid}),
        g.handle, p, devptr(d_ef), devptr(d_nf), devptr(d_gf), R, devptr(b_ef), devptr(b_nf), devptr(b_gf),
        ws.ptr, wsb, UInt32(0), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    (graphs=g, ef=oe == 0 ? nothing : download!(o_ef, b_ef), nf=on == 0 ? nothing : download!(o_nf, b_nf),
     gf=og == 0 ? nothing : download!(o_gf, b_gf))                    # zero-width outputs → nothing (gnblock.jl:71-78)
end

# ---- GNCore (src/gncore.jl:46-68): core(x) = x + block(gn1(x)) + ffwd(gn2(x)) → gnx_core_forward ----
struct LayerNorm                                                       # Flux.LayerNorm(d): diag scale γ, bias β
    γ::Vector{Float32}; β::Vector{Float32}
end
LayerNorm(d::Integer) = LayerNorm(ones(Float32, d), zeros(Float32, d))

struct GnxLayerNorm; gamma::Ptr{Cfloat}; beta::Ptr{Cfloat}; end       # gnx_layernorm
struct GnxFfn; fc1::GnxDense; fc2::GnxDense; end                       # gnx_ffn
struct GnxCoreParams                                                   # gnx_core_params (344 bytes)
    block::GnxBlockParams
    ln1::NTuple{3,GnxLayerNorm}; ln2::NTuple{3,GnxLayerNorm}; ff::NTuple{3,GnxFfn}
    eps::Cfloat; eps_mode::Int32
end

struct GNCore
    block::GNBlock
    ffwd::NTuple{3,Tuple{Dense,Dense}}                                 # gnfeedforward.jl:17-31: Dense(d => 4d, relu), Dense(4d => d)
    gn1::NTuple{3,LayerNorm}; gn2::NTuple{3,LayerNorm}                 # gngraphnorm.jl:9-17
    dims::NTuple{3,Int}
end
function GNCore(dims; dropout=0)                                       # src/gncore.jl:46-54
    @assert all(dims .> 0)                                             # gnfeedforward.jl:18, gngraphnorm.jl:10
    d = Tuple(dims)
    GNCore(GNBlock(d => d; dropout), map(k -> (Dense(k, 4k, :relu), Dense(4k, k)), d), map(LayerNorm, d), map(LayerNorm, d), d)
end

function (m::GNCore)(x)                                                # src/gncore.jl:56-68
    (; graphs, ef, nf, gf) = x
    @assert ef !== nothing && nf !== nothing && gf !== nothing         # graphnetadd needs all three (gncore.jl:61-68)
    g::GNGraphBatch = graphs
    R = size(ef, 3)
    keep = DevBuf[]                                                    # device copies of every parameter, alive across the ccall
    up(a) = (b = upload(a); push!(keep, b); devptr(b))
    dn(d::Dense) = GnxDense(up(d.weight), up(d.bias), Int32(actcode(d.σ)), 0)
    ln(l::LayerNorm) = GnxLayerNorm(up(l.γ), up(l.β))
    b = m.block
    bp = GnxBlockParams(b.in..., b.out..., dn(b.edgefn), dn(b.nodefn), dn(b.graphfn))
    p = Ref(GnxCoreParams(bp, map(ln, m.gn1), map(ln, m.gn2), map(t -> GnxFfn(dn(t[1]), dn(t[2])), m.ffwd), 1f-5, Int32(0)))
    d_ef, d_nf, d_gf = upload(ef), upload(nf), upload(gf)
    o_ef, o_nf, o_gf = similar(ef), similar(nf), similar(gf)
    b_ef, b_nf, b_gf = DevBuf(sizeof(o_ef)), DevBuf(sizeof(o_nf)), DevBuf(sizeof(o_gf))
    wsb = ccall((:gnx_core_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxCoreParams}, Int64), g.handle, p, R)
    ws = DevBuf(wsb)
    GC.@preserve keep d_ef d_nf d_gf b_ef b_nf b_gf ws check(ccall((:gnx_core_forward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{GnxCoreParams}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat},
         Ptr{Cvoid}, Csize_t, UInt32, Ptr{Cvoid}),
        g.handle, p, devptr(d_ef), devptr(d_nf), devptr(d_gf), R, devptr(b_ef), devptr(b_nf), devptr(b_gf),
        ws.ptr, wsb, UInt32(0), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    (graphs=g, ef=download!(o_ef, b_ef), nf=download!(o_nf, b_nf), gf=download!(o_gf, b_gf))
end

# pullback of (m::GNCore)(x) → gnx_core_backward: takes the forward's INPUT x and the cotangent ȳ of its output; every intermediate is
# recomputed inside the library.  Returns ∂ef, ∂nf, ∂gf and the parameter gradients in the order of the struct fields.
struct GnxLayerNormGrad; gamma::Ptr{Cfloat}; beta::Ptr{Cfloat}; end
struct GnxFfnGrad; fc1::GnxDenseGrad; fc2::GnxDenseGrad; end
struct GnxCoreGrads
    block::GnxBlockGrads
    ln1::NTuple{3,GnxLayerNormGrad}; ln2::NTuple{3,GnxLayerNormGrad}; ff::NTuple{3,GnxFfnGrad}
end
function core_pullback(m::GNCore, x, ȳ)
    g::GNGraphBatch = x.graphs
    R = size(x.ef, 3)
    keep = DevBuf[]
    up(a) = (b = upload(a); push!(keep, b); devptr(b))
    dn(d::Dense) = GnxDense(up(d.weight), up(d.bias), Int32(actcode(d.σ)), 0)
    ln(l::LayerNorm) = GnxLayerNorm(up(l.γ), up(l.β))
    b = m.block
    bp = GnxBlockParams(b.in..., b.out..., dn(b.edgefn), dn(b.nodefn), dn(b.graphfn))
    p = Ref(GnxCoreParams(bp, map(ln, m.gn1), map(ln, m.gn2), map(t -> GnxFfn(dn(t[1]), dn(t[2])), m.ffwd), 1f-5, Int32(0)))
    gbuf = Any[]                                                       # (host template, device buffer) of every gradient, in struct order
    gnew(a) = (bf = DevBuf(sizeof(a)); push!(gbuf, (a, bf)); Ptr{Cfloat}(bf.ptr))
    gd(d::Dense) = GnxDenseGrad(gnew(d.weight), gnew(d.bias))
    gl(l::LayerNorm) = GnxLayerNormGrad(gnew(l.γ), gnew(l.β))
    grads = Ref(GnxCoreGrads(GnxBlockGrads(gd(b.edgefn), gd(b.nodefn), gd(b.graphfn)), map(gl, m.gn1), map(gl, m.gn2),
                             map(t -> GnxFfnGrad(gd(t[1]), gd(t[2])), m.ffwd)))
    ins = (upload(x.ef), upload(x.nf), upload(x.gf)); cots = (upload(ȳ.ef), upload(ȳ.nf), upload(ȳ.gf))
    dins = (DevBuf(sizeof(x.ef)), DevBuf(sizeof(x.nf)), DevBuf(sizeof(x.gf)))
    wsb = ccall((:gnx_core_backward_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxCoreParams}, Int64), g.handle, p, R)
    ws = DevBuf(wsb)
    GC.@preserve keep gbuf ins cots dins ws check(ccall((:gnx_core_backward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{GnxCoreParams}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64,
         Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{GnxCoreGrads}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
        g.handle, p, devptr(ins[1]), devptr(ins[2]), devptr(ins[3]), devptr(cots[1]), devptr(cots[2]), devptr(cots[3]), R,
        devptr(dins[1]), devptr(dins[2]), devptr(dins[3]), grads, ws.ptr, wsb, C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    (ef=download!(similar(x.ef), dins[1]), nf=download!(similar(x.nf), dins[2]), gf=download!(similar(x.gf), dins[3]),
     params=[download!(similar(a), bf) for (a, bf) in gbuf])           # block (W, b) x 3, gn1 (γ, β) x 3, gn2 (γ, β) x 3, ffwd (W1, b1, W2, b2) x 3
end

# GNCoreList is `foldl((x, f) -> f(x), list; init=x)` exactly as src/gncorelist.jl:43-45.
struct GNCoreList{T}; list::T; end
(m::GNCoreList)(x) = foldl((i, fn) -> fn(i), m.list; init=x)

# ---- training: the pullback of (m::GNBlock)(x) = gnx_block_backward (what Flux.withgradient obtains from Zygote in
#      examples/sort/sort.jl:122-132).  `block_pullback(m, x, y, ȳ)` returns (∂ef, ∂nf, ∂gf, (∂W, ∂b) for the three Dense layers);
#      with ChainRulesCore loaded it is the body of the rrule below.  (gnx_core_backward is bound the same way; the tested
#      binding of both is graphnets.jl_amd/api.py: _BlockFn / _CoreFn.) ----
function block_pullback(m::GNBlock, x, y, ȳ)
    g::GNGraphBatch = x.graphs
    R = size(something(x.ef, x.nf, x.gf), 3)
    W = [upload(m.edgefn.weight), upload(m.nodefn.weight), upload(m.graphfn.weight)]
    B = [upload(m.edgefn.bias), upload(m.nodefn.bias), upload(m.graphfn.bias)]
    mk(d::Dense, w, b) = GnxDense(devptr(w), devptr(b), Int32(actcode(d.σ)), 0)
    p = Ref(GnxBlockParams(m.in..., m.out..., mk(m.edgefn, W[1], B[1]), mk(m.nodefn, W[2], B[2]), mk(m.graphfn, W[3], B[3])))
    ins = (upload(x.ef), upload(x.nf), upload(x.gf)); outs = (upload(y.ef), upload(y.nf), upload(y.gf))
    cots = (upload(ȳ.ef), upload(ȳ.nf), upload(ȳ.gf))
    zlike(a) = isnothing(a) ? nothing : DevBuf(sizeof(a))
    dins = (zlike(x.ef), zlike(x.nf), zlike(x.gf))
    layers = (m.edgefn, m.nodefn, m.graphfn)
    gW = [DevBuf(sizeof(l.weight)) for l in layers]; gB = [DevBuf(sizeof(l.bias)) for l in layers]
    gptr(b) = Ptr{Cfloat}(b.ptr)
    grads = Ref(GnxBlockGrads(GnxDenseGrad(gptr(gW[1]), gptr(gB[1])), GnxDenseGrad(gptr(gW[2]), gptr(gB[2])), GnxDenseGrad(gptr(gW[3]), gptr(gB[3]))))
    wsb = ccall((:gnx_block_backward_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxBlockParams}, Int64), g.handle, p, R)
    ws = DevBuf(wsb)
    GC.@preserve W B ins outs cots dins gW gB ws check(ccall((:gnx_block_backward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{GnxBlockParams}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat},
         Ptr{Cfloat}, Int64, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{GnxBlockGrads}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
        g.handle, p, devptr(ins[1]), devptr(ins[2]), devptr(ins[3]), devptr(outs[1]), devptr(outs[2]), devptr(outs[3]),
        devptr(cots[1]), devptr(cots[2]), devptr(cots[3]), R, devptr(dins[1]), devptr(dins[2]), devptr(dins[3]), grads, ws.ptr, wsb, C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    dl(a, b) = isnothing(a) ? nothing : download!(similar(a), b)
    (ef=dl(x.ef, dins[1]), nf=dl(x.nf, dins[2]), gf=dl(x.gf, dins[3]),
     params=[(weight=download!(similar(l.weight), gW[i]), bias=download!(similar(l.bias), gB[i])) for (i, l) in enumerate(layers)])
end

# The rrule a maintainer adds once ChainRulesCore is a dependency (kept as a comment: this module has no dependencies):
#   function ChainRulesCore.rrule(m::GNBlock, x)
#       y = m(x)
#       pb(ȳ) = (g = block_pullback(m, x, y, ȳ);
#                (Tangent{GNBlock}(edgefn=Tangent{Dense}(; g.params[1]...), nodefn=Tangent{Dense}(; g.params[2]...), graphfn=Tangent{Dense}(; g.params[3]...)),
#                 Tangent{typeof(x)}(ef=g.ef, nf=g.nf, gf=g.gf)))
#       y, pb
#   end

# ---- GNBlock whose update functions are Chains of Dense layers (src/gnblock.jl:1-6) → gnx_chain_block_forward / _backward ----
struct GnxChain; layers::Ptr{GnxDense}; widths::Ptr{Int32}; n_layers::Int32; reserved::Int32; end
struct GnxChainBlockParams; de::Int32; dn::Int32; dg::Int32; reserved::Int32; edgefn::GnxChain; nodefn::GnxChain; graphfn::GnxChain; end
struct GnxChainBlockGrads; edgefn::Ptr{GnxDenseGrad}; nodefn::Ptr{GnxDenseGrad}; graphfn::Ptr{GnxDenseGrad}; end
struct ChainBlock                                                      # GNBlock(Chain(Dense...), Chain(Dense...), Chain(Dense...))
    edgefn::Vector{Dense}; nodefn::Vector{Dense}; graphfn::Vector{Dense}; in::NTuple{3,Int}
end
outwidth(c::Vector{Dense}) = isempty(c) ? 0 : size(c[end].weight, 1)
# device copies of every layer + the host arrays the parameter struct points at; `keep` holds what must outlive the call
function chain_params(m::ChainBlock, keep::Vector{Any})
    function one(c::Vector{Dense})
        W = [upload(l.weight) for l in c]; B = [upload(l.bias) for l in c]
        descr = [GnxDense(devptr(W[i]), devptr(B[i]), Int32(actcode(c[i].σ)), 0) for i in eachindex(c)]
        widths = Int32[size(l.weight, 1) for l in c]
        push!(keep, W, B, descr, widths)
        GnxChain(isempty(c) ? C_NULL : pointer(descr), isempty(c) ? C_NULL : pointer(widths), Int32(length(c)), 0)
    end
    GnxChainBlockParams(m.in..., 0, one(m.edgefn), one(m.nodefn), one(m.graphfn))
end
function (m::ChainBlock)(x)
    g::GNGraphBatch = x.graphs
    R = size(something(x.ef, x.nf, x.gf), 3)
    keep = Any[]
    p = Ref(chain_params(m, keep))
    oe, on, og = outwidth(m.edgefn), outwidth(m.nodefn), outwidth(m.graphfn)
    ins = (upload(x.ef), upload(x.nf), upload(x.gf))
    o = (zeros(Float32, oe, nedges(g), R), zeros(Float32, on, nnodes(g), R), zeros(Float32, og, ngraphs(g), R))
    b = (DevBuf(sizeof(o[1])), DevBuf(sizeof(o[2])), DevBuf(sizeof(o[3])))
    wsb = ccall((:gnx_chain_block_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxChainBlockParams}, Int64), g.handle, p, R)
    wsb == 0 && error("gnx: ", unsafe_string(ccall((:gnx_last_error, libgnx), Cstring, ())))
    ws = DevBuf(wsb)
    GC.@preserve keep ins b ws check(ccall((:gnx_chain_block_forward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{GnxChainBlockParams}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cvoid}, Csize_t, UInt32, Ptr{Cvoid}),
        g.handle, p, devptr(ins[1]), devptr(ins[2]), devptr(ins[3]), R, devptr(b[1]), devptr(b[2]), devptr(b[3]), ws.ptr, wsb, UInt32(0), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    (graphs=g, ef=oe == 0 ? nothing : download!(o[1], b[1]), nf=on == 0 ? nothing : download!(o[2], b[2]), gf=og == 0 ? nothing : download!(o[3], b[3]))
end
# pullback: gradients w.r.t. the inputs and every layer's (weight, bias); the forward is recomputed inside the library
function chain_pullback(m::ChainBlock, x, ȳ)
    g::GNGraphBatch = x.graphs
    R = size(something(x.ef, x.nf, x.gf), 3)
    keep = Any[]
    p = Ref(chain_params(m, keep))
    ins = (upload(x.ef), upload(x.nf), upload(x.gf)); cots = (upload(ȳ.ef), upload(ȳ.nf), upload(ȳ.gf))
    zlike(a) = isnothing(a) ? nothing : DevBuf(sizeof(a))
    dins = (zlike(x.ef), zlike(x.nf), zlike(x.gf))
    chains = (m.edgefn, m.nodefn, m.graphfn)
    gW = [[DevBuf(sizeof(l.weight)) for l in c] for c in chains]; gB = [[DevBuf(sizeof(l.bias)) for l in c] for c in chains]
    arrs = [[GnxDenseGrad(Ptr{Cfloat}(gW[t][i].ptr), Ptr{Cfloat}(gB[t][i].ptr)) for i in eachindex(chains[t])] for t in 1:3]
    gp(t) = isempty(arrs[t]) ? Ptr{GnxDenseGrad}(C_NULL) : pointer(arrs[t])
    grads = Ref(GnxChainBlockGrads(gp(1), gp(2), gp(3)))
    wsb = ccall((:gnx_chain_block_backward_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxChainBlockParams}, Int64), g.handle, p, R)
    ws = DevBuf(wsb)
    GC.@preserve keep ins cots dins gW gB arrs ws check(ccall((:gnx_chain_block_backward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{GnxChainBlockParams}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Int64,
         Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{GnxChainBlockGrads}, Ptr{Cvoid}, Csize_t, Ptr{Cvoid}),
        g.handle, p, devptr(ins[1]), devptr(ins[2]), devptr(ins[3]), devptr(cots[1]), devptr(cots[2]), devptr(cots[3]), R,
        devptr(dins[1]), devptr(dins[2]), devptr(dins[3]), grads, ws.ptr, wsb, C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    dl(a, b) = isnothing(a) ? nothing : download!(similar(a), b)
    (ef=dl(x.ef, dins[1]), nf=dl(x.nf, dins[2]), gf=dl(x.gf, dins[3]),
     params=[[(weight=download!(similar(l.weight), gW[t][i]), bias=download!(similar(l.bias), gB[t][i])) for (i, l) in enumerate(chains[t])] for t in 1:3])
end

# ---- a chain of layers as ONE hipGraph inside libgnx (gnx_model_*): decoder(core(encoder(x))) of examples/sort/sort.jl:68-75 ----
struct GnxLayer; kind::Int32; reserved::Int32; params::Ptr{Cvoid}; end
mutable struct Model
    handle::Ptr{Cvoid}; graphs::GNGraphBatch; keep::Vector{Any}; outdims::NTuple{3,Int}
end
function Model(layers::AbstractVector, x)
    g::GNGraphBatch = x.graphs
    R = size(something(x.ef, x.nf, x.gf), 3)
    keep = Any[]
    up(a) = (b = upload(a); push!(keep, b); devptr(b))
    dn(d::Dense) = GnxDense(up(d.weight), up(d.bias), Int32(actcode(d.σ)), 0)
    ln(l::LayerNorm) = GnxLayerNorm(up(l.γ), up(l.β))
    bparams(b::GNBlock) = GnxBlockParams(b.in..., b.out..., dn(b.edgefn), dn(b.nodefn), dn(b.graphfn))
    descs = GnxLayer[]
    for l in layers
        if l isa GNBlock
            r = Ref(bparams(l)); push!(keep, r)
            push!(descs, GnxLayer(0, 0, Base.unsafe_convert(Ptr{Cvoid}, r)))
        else
            r = Ref(GnxCoreParams(bparams(l.block), map(ln, l.gn1), map(ln, l.gn2), map(t -> GnxFfn(dn(t[1]), dn(t[2])), l.ffwd), 1f-5, Int32(0)))
            push!(keep, r)
            push!(descs, GnxLayer(1, 0, Base.unsafe_convert(Ptr{Cvoid}, r)))
        end
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:gnx_model_create, libgnx), Int32, (Ptr{Cvoid}, Ptr{GnxLayer}, Int32, Int64, Ptr{Ptr{Cvoid}}), g.handle, descs, length(descs), R, h))
    dims = zeros(Int32, 3)
    check(ccall((:gnx_model_out_dims, libgnx), Int32, (Ptr{Cvoid}, Ptr{Int32}), h[], dims))
    m = Model(h[], g, keep, Tuple(Int.(dims)))
    finalizer(x -> ccall((:gnx_model_destroy, libgnx), Int32, (Ptr{Cvoid},), x.handle), m)
    m
end
function (m::Model)(x)                                                 # one hipGraphLaunch after the first call with these buffers
    g = m.graphs; R = size(something(x.ef, x.nf, x.gf), 3)
    d_ef, d_nf, d_gf = upload(x.ef), upload(x.nf), upload(x.gf)
    (oe, on, og) = m.outdims
    o_ef, o_nf, o_gf = zeros(Float32, oe, nedges(g), R), zeros(Float32, on, nnodes(g), R), zeros(Float32, og, ngraphs(g), R)
    b_ef, b_nf, b_gf = DevBuf(sizeof(o_ef)), DevBuf(sizeof(o_nf)), DevBuf(sizeof(o_gf))
    GC.@preserve d_ef d_nf d_gf b_ef b_nf b_gf check(ccall((:gnx_model_forward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, UInt32, Ptr{Cvoid}),
        m.handle, devptr(d_ef), devptr(d_nf), devptr(d_gf), devptr(b_ef), devptr(b_nf), devptr(b_gf), UInt32(0), C_NULL))
    hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
    (graphs=g, ef=oe == 0 ? nothing : download!(o_ef, b_ef), nf=on == 0 ? nothing : download!(o_nf, b_nf), gf=og == 0 ? nothing : download!(o_gf, b_gf))
end

# ---- multi-GPU: whole graphs sharded over the devices of this process, gf' all-gathered (gnx_dist_*, SURVEY §8e).
#      partition_graphs: equal graph counts per rank, snake order by edge count; DistBlock: one GNGraphBatch per device built from
#      ITS graphs, per-device gnx_block_forward, one RCCL all-gather of gf' restored to the ORIGINAL graph order. ----
function partition_graphs(edge_counts::AbstractVector{<:Integer}, n_ranks::Integer)
    counts = Int64.(edge_counts); G = length(counts)
    off = zeros(Int64, n_ranks + 1); ids = zeros(Int64, G)
    check(ccall((:gnx_dist_partition, libgnx), Int32, (Ptr{Int64}, Int64, Int32, Ptr{Int64}, Ptr{Int64}), counts, G, n_ranks, off, ids))
    [ids[off[r]+1:off[r+1]] .+ 1 for r in 1:n_ranks]               # 1-based original graph ids per rank
end

mutable struct DistBlock
    handle::Ptr{Cvoid}; devices::Vector{Int32}; shards::Vector{Vector{Int64}}; batches::Vector{GNGraphBatch}; block::GNBlock
end
function DistBlock(block::GNBlock, adj_mats::AbstractVector, devices::AbstractVector{<:Integer})
    n = length(devices)
    shards = partition_graphs([count(isone, a) for a in adj_mats], n)
    off = Int64[0; cumsum(length.(shards))]; ids = Int64.(reduce(vcat, shards) .- 1)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    devs = Int32.(devices)
    check(ccall((:gnx_dist_create, libgnx), Int32, (Ptr{Int32}, Int32, Ptr{Int64}, Ptr{Int64}, Int64, Int32, Ptr{Ptr{Cvoid}}),
                devs, n, off, ids, length(adj_mats), block.out[3], h))
    batches = map(1:n) do r
        hipcheck(ccall((:hipSetDevice, libhip), Cint, (Cint,), devs[r]))
        GNGraphBatch([adj_mats[i] for i in shards[r]])                 # the handle lives on the device that is current at creation
    end
    d = DistBlock(h[], devs, shards, batches, block)
    finalizer(x -> ccall((:gnx_dist_destroy, libgnx), Int32, (Ptr{Cvoid},), x.handle), d)
    d
end
# (d::DistBlock)(xs): xs[r] = the batched tuple of rank r's graphs (`batch` of ITS graphs, in the order of d.shards[r]).  Every rank's
# inputs, parameters (replicated), outputs and workspace live on ITS device; the call is ONE gnx_dist_block_forward = per-device
# gnx_block_forward + one RCCL all-gather of gf' + the permutation back to the ORIGINAL graph order.  Returns (ys, gf_all):
# ys[r] = rank r's (ef', nf', gf') and gf_all = (DG', n_graphs) of the whole batch (identical on every device; rank 1's copy).
function (d::DistBlock)(xs::AbstractVector)
    n = length(d.devices); m = d.block
    @assert length(xs) == n
    (oe, on, og) = m.out
    G = sum(length, d.shards)
    setdev(r) = hipcheck(ccall((:hipSetDevice, libhip), Cint, (Cint,), d.devices[r]))
    prev = Ref{Cint}(0); hipcheck(ccall((:hipGetDevice, libhip), Cint, (Ptr{Cint},), prev))
    keep = Any[]
    params = Vector{Base.RefValue{GnxBlockParams}}(undef, n)
    ins = Vector{Any}(undef, n); outs = Vector{Any}(undef, n); hosts = Vector{Any}(undef, n)
    wss = Vector{DevBuf}(undef, n); wsb = zeros(Csize_t, n); gall = Vector{DevBuf}(undef, n)
    mk(dl::Dense, w, b) = GnxDense(devptr(w), devptr(b), Int32(actcode(dl.σ)), 0)
    for r in 1:n
        setdev(r)                                                      # hipMalloc / hipMemcpy below land on device r
        g = d.batches[r]; x = xs[r]
        R = size(something(x.ef, x.nf, x.gf), 3); @assert R == 1      # by-graph sharding: vector batches
        W = [upload(m.edgefn.weight), upload(m.nodefn.weight), upload(m.graphfn.weight)]
        B = [upload(m.edgefn.bias), upload(m.nodefn.bias), upload(m.graphfn.bias)]
        params[r] = Ref(GnxBlockParams(m.in..., m.out..., mk(m.edgefn, W[1], B[1]), mk(m.nodefn, W[2], B[2]), mk(m.graphfn, W[3], B[3])))
        ins[r] = (upload(x.ef), upload(x.nf), upload(x.gf))
        hosts[r] = (zeros(Float32, oe, nedges(g), 1), zeros(Float32, on, nnodes(g), 1), zeros(Float32, og, ngraphs(g), 1))
        outs[r] = (DevBuf(sizeof(hosts[r][1])), DevBuf(sizeof(hosts[r][2])), DevBuf(sizeof(hosts[r][3])))
        wsb[r] = ccall((:gnx_block_workspace_bytes, libgnx), Csize_t, (Ptr{Cvoid}, Ptr{GnxBlockParams}, Int64), g.handle, params[r], 1)
        wss[r] = DevBuf(wsb[r]); gall[r] = DevBuf(4 * G * og)
        push!(keep, W, B)
    end
    col(f) = [f(r) for r in 1:n]
    hs = col(r -> d.batches[r].handle)
    ps = col(r -> Base.unsafe_convert(Ptr{GnxBlockParams}, params[r]))
    efs = col(r -> devptr(ins[r][1])); nfs = col(r -> devptr(ins[r][2])); gfs = col(r -> devptr(ins[r][3]))
    eos = col(r -> devptr(outs[r][1])); nos = col(r -> devptr(outs[r][2])); gos = col(r -> devptr(outs[r][3]))
    gas = col(r -> devptr(gall[r])); wsp = col(r -> wss[r].ptr)
    GC.@preserve keep params ins outs wss gall check(ccall((:gnx_dist_block_forward, libgnx), Int32,
        (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{GnxBlockParams}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}},
         Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cfloat}}, Ptr{Ptr{Cvoid}}, Ptr{Csize_t}, UInt32, Ptr{Ptr{Cvoid}}),
        d.handle, hs, ps, efs, nfs, gfs, eos, nos, gos, gas, wsp, wsb, UInt32(0), C_NULL))
    ys = map(1:n) do r
        setdev(r); hipcheck(ccall((:hipDeviceSynchronize, libhip), Cint, ()))
        (graphs=d.batches[r], ef=oe == 0 ? nothing : download!(hosts[r][1], outs[r][1]), nf=on == 0 ? nothing : download!(hosts[r][2], outs[r][2]),
         gf=og == 0 ? nothing : download!(hosts[r][3], outs[r][3]))
    end
    setdev(1)
    gf_all = download!(zeros(Float32, og, G), gall[1])
    hipcheck(ccall((:hipSetDevice, libhip), Cint, (Cint,), prev[]))
    ys, gf_all
end

end # module
