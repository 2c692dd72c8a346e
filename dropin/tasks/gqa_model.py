from rgqa_amd.tasks.gqa_model import *  # noqa: F401,F403
from rgqa_amd.tasks.gqa_model import GQAModel, GQAModel_maha, MAX_GQA_LENGTH, args  # noqa: F401
