"""uniter/tokenization.py is a copy of lxrt/tokenization.py in the reference; here it is the same module."""
from ..lxrt.tokenization import *            # noqa: F401,F403
from ..lxrt.tokenization import BertTokenizer  # noqa: F401
