"""Mirror of the reference's lxrt_vis/entry.py: `LXRTEncoder.forward(sents, feats, visual_attention_mask=None,
output_attention=False)` returns `(output, att_probs, input_ids)` (reference :109-121)."""
from ..lxrt import entry as _entry
from ..lxrt.entry import InputFeatures, convert_sents_to_features, set_visual_config  # noqa: F401
from .modeling import LXRTFeatureExtraction as VisualBertForLXRFeature, VISUAL_CONFIG  # noqa: F401


class LXRTEncoder(_entry.LXRTEncoder):
    MODEL_CLASS = VisualBertForLXRFeature

    def forward(self, sents, feats, visual_attention_mask=None, output_attention=False):
        input_ids, segment_ids, input_mask, lengths = self._tokenize(sents, feats[0].device)
        output, att_probs = self.model(input_ids, segment_ids, input_mask, visual_feats=feats,
                                       visual_attention_mask=visual_attention_mask, output_attention=output_attention,
                                       token_lengths=lengths)
        return output, att_probs, input_ids
