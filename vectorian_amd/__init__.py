"""vectorian_amd -- MI355X (gfx950) native brute-force alignment search behind
Vectorian's Session / Index.find() operator surface.

Only the hot path of poke1024/vectorian is implemented here:
query x sentence similarity -> alignment / transport score -> bounded result set.
All scoring runs in hand-written HIP kernels behind the C-ABI of
include/vectorian_hip.h; there is no CPU fallback.
"""

__version__ = "0.1.0"
