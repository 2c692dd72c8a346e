from rgqa_amd.butd.butd import *  # noqa: F401,F403
from rgqa_amd.butd.butd import GQABUTD, MAX_GQA_LENGTH  # noqa: F401
