"""Drop-in `lxrt` package: put /root/repo/dropin (and /root/repo) in front of the reference's src/ on PYTHONPATH and
`from lxrt.entry import LXRTEncoder`, `from lxrt.modeling import BertLayerNorm, GeLU`, `from lxrt.optimization import
BertAdam`, `from lxrt.tokenization import BertTokenizer` resolve to the MI355X implementation (rgqa_amd.lxrt.*).
Other submodules (e.g. lxrt.file_utils) still resolve to the reference's own files if its src/ is on sys.path."""
import os
import sys

for _d in sys.path:
    _c = os.path.join(_d or ".", "lxrt")
    if os.path.isdir(_c) and os.path.abspath(_c) != os.path.dirname(os.path.abspath(__file__)) and _c not in __path__:
        __path__.append(_c)
