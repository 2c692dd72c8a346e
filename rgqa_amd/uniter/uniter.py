"""Mirror of the reference's uniter/uniter.py (:15-75): `GQAUNITER(num_answers)`, `GQAUNITER_maha`, `MAX_VQA_LENGTH`."""
import types

import torch.nn as nn

try:
    from param import args
except Exception:
    args = types.SimpleNamespace(from_scratch=False)

from .entry import UniterEncoder
from .modeling import BertLayerNorm, GeLU

# Max length including <bos> and <eos>
MAX_VQA_LENGTH = 20


class GQAUNITER(nn.Module):
    def __init__(self, num_answers, model_args=None):
        super().__init__()
        self.encoder = UniterEncoder(model_args if model_args is not None else args)
        hid_dim = self.encoder.dim
        self.logit_fc = nn.Sequential(
            nn.Linear(hid_dim, hid_dim * 2),
            GeLU(),
            BertLayerNorm(hid_dim * 2, eps=1e-12),
            nn.Linear(hid_dim * 2, num_answers)
        )
        self.logit_fc.apply(self.encoder.model.init_bert_weights)
        self.encoder.model.attach_head(self.logit_fc)

    def forward(self, feat, pos, sent):
        """feat (b, o, f), pos (b, o, 7), sent list[str] of length b -> logits (b, num_answers)."""
        logit, _ = self.encoder.forward_with_head(sent, feat, pos)
        return logit


class GQAUNITER_maha(GQAUNITER):
    def forward(self, feat, pos, sent):
        logit, x = self.encoder.forward_with_head(sent, feat, pos)
        return logit, x
