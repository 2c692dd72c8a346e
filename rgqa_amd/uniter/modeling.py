"""Mirror of the reference's uniter/modeling.py surface that uniter/entry.py and uniter/uniter.py use:
`UniterFeatureExtraction(config)` with `forward(input_ids, token_type_ids, attention_mask, visual_feats, visual_token_type_ids,
visual_attention_mask, img_pos_feat)` -> pooled_output (reference :638-655), `VISUAL_CONFIG`, `BertConfig`, `GeLU`, `BertLayerNorm`.
The module tree (`uniter.embeddings`, `uniter.img_embeddings`, `uniter.encoder.layer.N`, `uniter.pooler`) is built from the engine's
parameter table, so state_dict keys and shapes are the reference's (:560-635)."""
import torch

from ..engine import Engine
from ..lxrt.modeling import (BertConfig, BertLayerNorm, BertPreTrainedModel, GeLU, LXRTFeatureExtraction as _Base,  # noqa: F401
                             VisualConfig, gelu)

VISUAL_CONFIG = VisualConfig()          # uniter/modeling.py:372-400; the entry sets 9/5/5 on it and never uses them (entry.py:57-60)
UNITER_POS_DIM = 7                      # nn.Linear(7, hidden) (:600)


class UniterFeatureExtraction(_Base):
    TREE_PREFIX = "encoder.model."      # GQAUNITER.encoder = UniterEncoder, .model = this module (uniter/uniter.py:18, entry.py:70)

    def __init__(self, config, precision=None):
        super().__init__(config, mode='x', precision=precision)

    def _make_engine(self, num_answers):
        c = self.config
        self._binding.engine = Engine(vocab_size=c.vocab_size, hidden=c.hidden_size, heads=c.num_attention_heads, inter=c.intermediate_size,
                                      max_pos=c.max_position_embeddings, type_vocab=c.type_vocab_size, l_layers=c.num_hidden_layers, x_layers=0,
                                      r_layers=0, feat_dim=VISUAL_CONFIG.visual_feat_dim, pos_dim=UNITER_POS_DIM, num_answers=num_answers,
                                      precision=self.precision, hidden_dropout=c.hidden_dropout_prob, attn_dropout=c.attention_probs_dropout_prob,
                                      arch=2)

    def load_pending_bert(self):
        """BERT weights found locally by from_pretrained: `bert.*` / bare keys become `uniter.*` (what uniter/entry.py:103-107 does for
        UNITER checkpoints); keys the single-stream model lacks are ignored (strict=False)."""
        sd = getattr(self, "_pending_bert_state", None)
        if sd is None:
            return
        mapped = {}
        for k, v in sd.items():
            k = k.replace("gamma", "weight").replace("beta", "bias")
            k = k[len("bert."):] if k.startswith("bert.") else k
            mapped["uniter." + k] = v
        self.load_state_dict(mapped, strict=False)
        self._pending_bert_state = None

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, visual_feats=None, visual_token_type_ids=None,
                visual_attention_mask=None, img_pos_feat=None, token_lengths=None):
        self._check_visual(input_ids, visual_feats, visual_token_type_ids, visual_attention_mask)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        _, pooled = self._run(visual_feats, img_pos_feat, input_ids, attention_mask, token_type_ids, want_logits=False, lengths=token_lengths)
        return pooled

    def forward_with_head(self, input_ids, token_type_ids, attention_mask, visual_feats, img_pos_feat, token_lengths=None):
        if self._head is None:
            raise RuntimeError("no head attached")
        return self._run(visual_feats, img_pos_feat, input_ids, attention_mask, token_type_ids, want_logits=True, lengths=token_lengths)

    @staticmethod
    def _check_visual(input_ids, visual_feats, visual_token_type_ids, visual_attention_mask):
        # the entry always passes all-ones region masks and token type 1 for every region (uniter/entry.py:95-96): that is what the
        # engine computes; anything else is refused rather than silently ignored
        if visual_attention_mask is not None and not bool((visual_attention_mask == 1).all()):
            raise NotImplementedError("UNITER on rgqa_amd: every region is attended (visual_attention_mask of ones, uniter/entry.py:96)")
        if visual_token_type_ids is not None and not bool((visual_token_type_ids == 1).all()):
            raise NotImplementedError("UNITER on rgqa_amd: regions carry token type 1 (uniter/entry.py:95)")
