"""Drop-in `tasks` package: `from tasks.gqa_model import GQAModel` resolves to rgqa_amd.tasks.gqa_model; every other
`tasks.*` module (gqa_conf, gqa_data, gqa_mixup_vis, ...) resolves to the reference's own files on sys.path."""
import os
import sys

for _d in sys.path:
    _c = os.path.join(_d or ".", "tasks")
    if os.path.isdir(_c) and os.path.abspath(_c) != os.path.dirname(os.path.abspath(__file__)) and _c not in __path__:
        __path__.append(_c)
