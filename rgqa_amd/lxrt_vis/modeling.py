"""Mirror of the reference's lxrt_vis/modeling.py surface used by lxrt_vis/entry.py: `LXRTFeatureExtraction.forward(...,
output_attention=False)` returns `(pooled_output, (l2v_atts, v2l_atts))` for mode 'x' (reference :1029-1036): one entry per
cross-modality layer, `[B, heads, T, O]` / `[B, heads, O, T]` f32 probabilities when output_attention is set, else None
(reference :347-350)."""
from ..lxrt.modeling import *                 # noqa: F401,F403
from ..lxrt.modeling import LXRTFeatureExtraction as _LXRTFeatureExtraction, VISUAL_CONFIG  # noqa: F401


class LXRTFeatureExtraction(_LXRTFeatureExtraction):
    def forward(self, input_ids, token_type_ids=None, attention_mask=None, visual_feats=None, visual_attention_mask=None,
                output_attention=False, token_lengths=None):
        pooled = super().forward(input_ids, token_type_ids, attention_mask, visual_feats=visual_feats,
                                 visual_attention_mask=visual_attention_mask, token_lengths=token_lengths)
        n = VISUAL_CONFIG.x_layers
        if not output_attention:
            return pooled, ([None] * n, [None] * n)
        e = self._binding.engine
        l2v = [e.cross_attention(i, "l2v") for i in range(n)]
        v2l = [e.cross_attention(i, "v2l") for i in range(n)]
        return pooled, (l2v, v2l)
