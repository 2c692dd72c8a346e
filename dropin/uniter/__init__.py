"""Drop-in `uniter` package (the UNITER backbone): `from uniter.uniter import GQAUNITER`, `from uniter.entry import
UniterEncoder` resolve to rgqa_amd.uniter (the same HIP engine, arch 2). Other submodules still resolve to the reference's own
files if its src/ is on sys.path."""
import os
import sys

for _d in sys.path:
    _c = os.path.join(_d or ".", "uniter")
    if os.path.isdir(_c) and os.path.abspath(_c) != os.path.dirname(os.path.abspath(__file__)) and _c not in __path__:
        __path__.append(_c)
