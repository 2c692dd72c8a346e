"""Mirror of the reference's uniter/entry.py: `UniterEncoder(args)` with `forward(sents, feats, boxes, visual_attention_mask=None)`
(:62-101), `convert_sents_to_features`, `InputFeatures`, `set_visual_config`."""
import os

import torch
import torch.nn as nn

from ..lxrt.entry import InputFeatures, convert_sents_to_features, LXRTEncoder as _LXRTEncoder  # noqa: F401
from .modeling import VISUAL_CONFIG, BertConfig, UniterFeatureExtraction as UFE
from .tokenization import BertTokenizer


def set_visual_config(args):
    VISUAL_CONFIG.l_layers = 9          # uniter/entry.py:57-60: fixed values, unused by the single-stream model
    VISUAL_CONFIG.x_layers = 5
    VISUAL_CONFIG.r_layers = 5


class UniterEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.max_seq_length = 20
        set_visual_config(args)
        self.tokenizer = BertTokenizer.from_pretrained("bert-base-cased", do_lower_case=True)
        self.model = UFE.from_pretrained("bert-base-cased")
        self.model.load_pending_bert()
        if getattr(args, "from_scratch", False):
            print("initializing all the weights")
            self.model.apply(self.model.init_bert_weights)
        self._id_cache = {}
        self._native = None

    @property
    def dim(self):
        return self.model.config.hidden_size

    _tokenize = _LXRTEncoder._tokenize

    def _prepare(self, sents, feats, boxes):
        assert feats.shape[1] == 36 or os.environ.get("RGQA_UNITER_ANY_ROIS"), "Not Using 36 ROIs, please change the following 2 lines"   # entry.py:94
        return self._tokenize(sents, feats.device)

    def forward(self, sents, feats, boxes, visual_attention_mask=None):
        input_ids, segment_ids, input_mask, lengths = self._prepare(sents, feats, boxes)
        return self.model(input_ids=input_ids, token_type_ids=segment_ids, attention_mask=input_mask, visual_feats=feats,
                          visual_attention_mask=visual_attention_mask, img_pos_feat=boxes, token_lengths=lengths)

    def forward_with_head(self, sents, feats, boxes):
        input_ids, segment_ids, input_mask, lengths = self._prepare(sents, feats, boxes)
        return self.model.forward_with_head(input_ids, segment_ids, input_mask, feats, boxes, token_lengths=lengths)

    def load(self, path):
        """uniter/entry.py:103-118: a UNITER checkpoint names the trunk `bert.*`; it is loaded as `uniter.*`, non-strictly."""
        state_dict = torch.load(path, map_location="cpu")
        for key in list(state_dict.keys()):
            if 'bert.' in key:
                state_dict[key.replace('bert.', 'uniter.')] = state_dict.pop(key)
        print("Load UNITER PreTrained Model from %s" % path)
        load_keys = set(state_dict.keys())
        model_keys = set(self.model.state_dict().keys())
        print()
        print("Weights in loaded but not in model:")
        for key in sorted(load_keys.difference(model_keys)):
            print(key)
        print()
        print("Weights in model but not in loaded:")
        for key in sorted(model_keys.difference(load_keys)):
            print(key)
        print()
        self.model.load_state_dict(state_dict, strict=False)
