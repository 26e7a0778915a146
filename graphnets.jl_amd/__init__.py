"""graphnets.jl_amd — MI355X-native GNBlock / GNCore forward behind GraphNets.jl's API surface.

The directory name follows the project naming contract and is not a valid Python identifier; import it as
`graphnets_jl_amd` (the one-line loader module at the repo root).
"""
from . import _lib
from ._lib import GnxError, LIB_PATH, profile_calibrate, profile_enable, profile_read, profile_reset
from .api import (NT, BlockPlan, Chain, Dense, Dropout, Graphed, Model, GNBlock, GNCore, GNCoreList, GNFeedForward, GNGraphBatch, GNGraphNorm, LayerNorm, batch,
                  efview, flatunpaddedef, flatunpaddednf, getedgefninput, getgraphfninput, getnodefninput, gfview, nfview,
                  collapsef, dropout_mask, flatunpaddedcollapsedef, logitcrossentropy, padded, testmode, trainmode, unbatch, unpadded, unpaddedcollapsedef, zerodim2nothing)

FLAG_FORCE_GENERIC = _lib.FLAG_FORCE_GENERIC
FLAG_NO_MFMA = _lib.FLAG_NO_MFMA
