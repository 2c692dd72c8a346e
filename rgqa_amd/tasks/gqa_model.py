"""Mirror of the reference's tasks/gqa_model.py (14-75): `GQAModel(num_answers)`, `GQAModel_maha`, `MAX_GQA_LENGTH`.

`logit_fc` is a real nn.Sequential(Linear, GeLU, LayerNorm, Linear) so checkpoint surgery that addresses
`logit_fc.3.weight` (pretrain/qa_answer_table.py:118-155) keeps working; its arithmetic is fused into the engine."""
import types

import torch.nn as nn

try:                                   # the reference's process-global flags (src/param.py:150) when run drop-in
    from param import args
except Exception:                      # stand-alone use: the launchers' defaults (run/gqa_conf_finetune.bash:14)
    args = types.SimpleNamespace(llayers=9, xlayers=5, rlayers=5, from_scratch=False)

from ..lxrt.entry import LXRTEncoder
from ..lxrt.modeling import BertLayerNorm, GeLU

# Max length including <bos> and <eos>
MAX_GQA_LENGTH = 30


class GQAModel(nn.Module):
    def __init__(self, num_answers, max_seq_length=MAX_GQA_LENGTH, model_args=None):
        super().__init__()
        self.lxrt_encoder = LXRTEncoder(model_args if model_args is not None else args, max_seq_length=max_seq_length)
        hid_dim = self.lxrt_encoder.dim
        self.logit_fc = nn.Sequential(
            nn.Linear(hid_dim, hid_dim * 2),
            GeLU(),
            BertLayerNorm(hid_dim * 2, eps=1e-12),
            nn.Linear(hid_dim * 2, num_answers)
        )
        self.logit_fc.apply(self.lxrt_encoder.model.init_bert_weights)
        self.lxrt_encoder.model.attach_head(self.logit_fc)

    def forward(self, feat, pos, sent):
        """feat (b, o, f), pos (b, o, 4), sent list[str] of length b -> logits (b, num_answers)."""
        logit, _ = self.lxrt_encoder.forward_with_head(sent, (feat, pos))
        return logit


class GQAModel_maha(GQAModel):
    def forward(self, feat, pos, sent):
        logit, x = self.lxrt_encoder.forward_with_head(sent, (feat, pos))
        return logit, x
