# Package entry point (Project.toml sits beside this directory): the module itself is ../GraphNetsHIP.jl — one file, usable on its own with
# `include("julia/GraphNetsHIP.jl"); using .GraphNetsHIP` — and ../ext/GraphNetsHIPChainRulesExt.jl is its ChainRulesCore extension.
include(joinpath(@__DIR__, "..", "GraphNetsHIP.jl"))
