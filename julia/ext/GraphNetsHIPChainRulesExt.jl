# Package extension of GraphNetsHIP: loaded by Julia (>= 1.9) only where ChainRulesCore is in the environment — Project.toml:
#
#     [weakdeps]
#     ChainRulesCore = "d360d2e6-b24c-11e9-a2a3-2a2ae2dbcce4"
#     [extensions]
#     GraphNetsHIPChainRulesExt = "ChainRulesCore"
#
# so the shim itself keeps no hard dependency.  With it, Zygote differentiates through the HIP forward: `Flux.withgradient(model)` of the
# reference's training loop (/root/reference/examples/sort/sort.jl:122-132) calls these rrules, whose pullbacks are ONE gnx_block_backward /
# gnx_core_backward each (deterministic, no atomics) — what the Python mirror's torch.autograd.Functions do (graphnets.jl_amd/api.py: _BlockFn,
# _CoreFn; tests/test_gpu_backward.py checks them against float64 autograd of an independent restatement).
# An optimiser step rewrites the weights in place, and gpu() prepared the layers' weight planes once: every rule below therefore starts with
# `refresh!(m)` (gnx_prepared_refresh, stream-ordered in front of the forward; a no-op for a layer that was never prepared) — the forward and
# the pullback's recompute of ONE gradient call see the same, current planes.  Inference calls between optimiser steps on the same layer
# objects still need `GraphNetsHIP.refresh!(layer)` (or `unprepare!`: the forward then prepares per call).
module GraphNetsHIPChainRulesExt

using GraphNetsHIP
using GraphNetsHIP: GNBlock, GNCore, GNCoreList, Dense, LayerNorm, block_pullback, core_pullback, core_train, newdropout, refresh!
import ChainRulesCore
using ChainRulesCore: Tangent, NoTangent, ZeroTangent, unthunk

feat(t) = t === nothing ? ZeroTangent() : t
dense_tangent(l::Dense, w, b) = Tangent{typeof(l)}(weight=w, bias=b)            # (σ carries no tangent)
ln_tangent(l::LayerNorm, g, b) = Tangent{typeof(l)}(γ=g, β=b)
upstream(ȳ, y) = (graphs=y.graphs, ef=zero_or(ȳ, :ef), nf=zero_or(ȳ, :nf), gf=zero_or(ȳ, :gf))
zero_or(ȳ, k) = (v = getproperty(ȳ, k); v isa ChainRulesCore.AbstractZero ? nothing : v)

# (m::GNBlock)(x)  (src/gnblock.jl:63-69): tangents of the three Dense layers and of the batched tuple's ef / nf / gf
function ChainRulesCore.rrule(m::GNBlock, x::NamedTuple)
    refresh!(m)
    y = m(x)
    function block_pb(ȳ_)
        ȳ = upstream(unthunk(ȳ_), y)
        g = block_pullback(m, x, y, ȳ)
        q = g.params
        m̄ = Tangent{typeof(m)}(edgefn=dense_tangent(m.edgefn, q[1].weight, q[1].bias), nodefn=dense_tangent(m.nodefn, q[2].weight, q[2].bias),
                               graphfn=dense_tangent(m.graphfn, q[3].weight, q[3].bias))
        x̄ = Tangent{typeof(x)}(graphs=NoTangent(), ef=feat(g.ef), nf=feat(g.nf), gf=feat(g.gf))
        (m̄, x̄)
    end
    y, block_pb
end

# (m::GNCore)(x)  (src/gncore.jl:56-68).  core_pullback returns the parameter gradients in struct order:
# block (W, b) x 3, gn1 (γ, β) x 3, gn2 (γ, β) x 3, ffwd (W1, b1, W2, b2) x 3
# A gradient call is Flux's training mode: with GNCore(dims; dropout = p > 0) the FeedForwards' Dropout (gnfeedforward.jl:27-31) is applied —
# gnx_core_forward_train with a fresh seed — and the pullback regenerates the same masks from it (gnx_core_backward_train).
function ChainRulesCore.rrule(m::GNCore, x::NamedTuple)
    refresh!(m)
    drop = newdropout(m)
    y = drop === nothing ? m(x) : core_train(m, x, drop)
    function core_pb(ȳ_)
        ȳ = upstream(unthunk(ȳ_), y)
        g = core_pullback(m, x, ȳ, drop)
        p = g.params
        b = m.block
        blk = Tangent{typeof(b)}(edgefn=dense_tangent(b.edgefn, p[1], p[2]), nodefn=dense_tangent(b.nodefn, p[3], p[4]), graphfn=dense_tangent(b.graphfn, p[5], p[6]))
        gn1 = ntuple(t -> ln_tangent(m.gn1[t], p[5 + 2t], p[6 + 2t]), 3)        # p[7..12]
        gn2 = ntuple(t -> ln_tangent(m.gn2[t], p[11 + 2t], p[12 + 2t]), 3)      # p[13..18]
        ffwd = ntuple(t -> (dense_tangent(m.ffwd[t][1], p[15 + 4t], p[16 + 4t]), dense_tangent(m.ffwd[t][2], p[17 + 4t], p[18 + 4t])), 3)   # p[19..30]
        m̄ = Tangent{typeof(m)}(block=blk, ffwd=ffwd, gn1=gn1, gn2=gn2)
        x̄ = Tangent{typeof(x)}(graphs=NoTangent(), ef=feat(g.ef), nf=feat(g.nf), gf=feat(g.gf))
        (m̄, x̄)
    end
    y, core_pb
end

end # module
