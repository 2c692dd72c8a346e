"""Drop-in `lxrt_vis` package (the reference's visualisation variant of `lxrt`): `from lxrt_vis.entry import
LXRTEncoder` resolves to rgqa_amd.lxrt_vis (same engine; forward(..., output_attention=True) also returns the cross-attention
probabilities). Other submodules still resolve to the reference's own files if its src/ is on sys.path."""
import os
import sys

for _d in sys.path:
    _c = os.path.join(_d or ".", "lxrt_vis")
    if os.path.isdir(_c) and os.path.abspath(_c) != os.path.dirname(os.path.abspath(__file__)) and _c not in __path__:
        __path__.append(_c)
