from rgqa_amd.uniter.entry import *  # noqa: F401,F403
from rgqa_amd.uniter import entry as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
