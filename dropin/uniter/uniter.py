from rgqa_amd.uniter.uniter import *  # noqa: F401,F403
from rgqa_amd.uniter import uniter as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
