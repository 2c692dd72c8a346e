"""Mirror of the reference's lxrt/entry.py (27-152): `LXRTEncoder(args, max_seq_length, mode='x')`,
`convert_sents_to_features`, `set_visual_config`, `InputFeatures` — same names, arguments and error behaviour,
with the nn.Module underneath replaced by the HIP engine."""
import os

import torch
import torch.nn as nn

from .tokenization import BertTokenizer
from .modeling import LXRTFeatureExtraction as VisualBertForLXRFeature, VISUAL_CONFIG
from . import optimization as _optimization

# The unchanged trainers call `nn.utils.clip_grad_norm_(self.model.parameters(), 5.)` (tasks/gqa_conf.py:201, gqa_mixup_vis.py:258).  While an
# engine-backed LXRTEncoder is ALIVE that name is routed through rgqa_amd.lxrt.optimization.clip_grad_norm_ (identical results, a fused fast path
# when the parameters are the views of one engine's arena, torch's own implementation for everything else): LXRTEncoder.__init__ installs the
# routing, the encoder's finaliser removes it when the last one is collected (optimization.install_clip_routing).  Importing this module changes
# nothing (until round 5 the import itself rebound the name for the whole process); RGQA_PATCH_CLIP=0 leaves torch untouched altogether.


class InputFeatures(object):
    """A single set of features of data."""

    def __init__(self, input_ids, input_mask, segment_ids):
        self.input_ids = input_ids
        self.input_mask = input_mask
        self.segment_ids = segment_ids


def convert_sents_to_features(sents, max_seq_length, tokenizer):
    """Tokenise, truncate to max_seq_length-2, add [CLS]/[SEP], zero-pad; mask 1/0; segment all 0 (reference :36-71)."""
    features = []
    for sent in sents:
        tokens_a = tokenizer.tokenize(sent.strip())
        if len(tokens_a) > max_seq_length - 2:
            tokens_a = tokens_a[:(max_seq_length - 2)]
        tokens = ["[CLS]"] + tokens_a + ["[SEP]"]
        input_ids = tokenizer.convert_tokens_to_ids(tokens)
        n = len(input_ids)
        pad = max_seq_length - n
        input_mask = [1] * n + [0] * pad
        segment_ids = [0] * max_seq_length
        input_ids = input_ids + [0] * pad
        assert len(input_ids) == max_seq_length
        assert len(input_mask) == max_seq_length
        assert len(segment_ids) == max_seq_length
        features.append(InputFeatures(input_ids=input_ids, input_mask=input_mask, segment_ids=segment_ids))
    return features


def set_visual_config(args):
    VISUAL_CONFIG.l_layers = args.llayers
    VISUAL_CONFIG.x_layers = args.xlayers
    VISUAL_CONFIG.r_layers = args.rlayers


class LXRTEncoder(nn.Module):
    MODEL_CLASS = VisualBertForLXRFeature      # lxrt_vis substitutes its variant

    def __init__(self, args, max_seq_length, mode='x'):
        super().__init__()
        self.max_seq_length = max_seq_length
        set_visual_config(args)
        self.tokenizer = BertTokenizer.from_pretrained("bert-base-uncased", do_lower_case=True)
        self.model = self.MODEL_CLASS.from_pretrained("bert-base-uncased", mode=mode)
        self.model.load_pending_bert()
        if getattr(args, "from_scratch", False):
            print("initializing all the weights")
            self.model.apply(self.model.init_bert_weights)
        self._id_cache = {}
        self._native = None

    def multi_gpu(self):
        """The reference wraps the inner model in single-process nn.DataParallel (:102-103).  Here data parallelism is one process
        per GPU (torchrun / torch.distributed.run): once the default process group exists, every backward of this module averages
        its gradient arena over the ranks (RCCL all-reduce, rgqa_amd.parallel) and every rank draws its own dropout stream; give
        the DataLoader a DistributedSampler.  In a single process nothing can be parallelised: say so."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            print("rgqa: data parallel over %d processes (rank %d): gradients are averaged in backward()" % (dist.get_world_size(), dist.get_rank()))
        elif torch.cuda.is_available() and torch.cuda.device_count() > 1:
            import warnings
            warnings.warn("rgqa: multi_gpu() in a single process uses ONE of the %d visible GPUs; launch one process per GPU with "
                          "`python -m torch.distributed.run --nproc-per-node N ...` and call torch.distributed.init_process_group('nccl') "
                          "for data parallelism" % torch.cuda.device_count())
        return None

    @property
    def dim(self):
        return self.model.config.hidden_size   # 768 for bert-base, the only size the reference ever builds (:105-107)

    def _tokenize(self, sents, device):
        """ids / segment / mask `[B, T]` on `device` + the host token counts. The batch goes through the native tokenizer
        (rgqa_tokenizer_encode) in one call; sentences it hands back (non-ASCII) take the Python path, memoised per string."""
        if self.tokenizer is None:
            raise RuntimeError("LXRTEncoder has no tokenizer: the BERT vocabulary was not found (set RGQA_BERT_VOCAB)")
        import numpy as np
        T = self.max_seq_length
        n = len(sents)
        ids = torch.zeros(n, T, dtype=torch.long)
        mask = torch.zeros(n, T, dtype=torch.long)
        if device.type == "cuda":
            ids, mask = ids.pin_memory(), mask.pin_memory()
        ids_np, mask_np = ids.numpy(), mask.numpy()
        if self._native is None and os.environ.get("RGQA_NATIVE_TOKENIZER", "1") != "0":
            from .tokenization import NativeBatchEncoder
            self._native = NativeBatchEncoder(self.tokenizer)
        if self._native is not None:
            lengths, needs = self._native.encode(sents, T, ids_np, mask_np)
            todo = np.nonzero(needs)[0]
        else:
            lengths, todo = np.zeros(n, dtype=np.int32), range(n)
        for i in todo:
            s = sents[i]
            hit = self._id_cache.get(s)
            if hit is None:
                f = convert_sents_to_features([s], T, self.tokenizer)[0]
                hit = (f.input_ids, f.input_mask)
                if len(self._id_cache) < 2000000:
                    self._id_cache[s] = hit
            ids_np[i, :] = hit[0]
            mask_np[i, :] = hit[1]
            lengths[i] = sum(hit[1])
        seg = torch.zeros(n, T, dtype=torch.long, device=device)
        # real token counts ([CLS] .. [SEP]; the mask is a prefix of ones, :56-66): lets the engine skip the padding rows.
        # RGQA_VARLEN=0 computes every padded position, as the reference does.
        lens = [int(v) for v in lengths] if os.environ.get("RGQA_VARLEN", "1") != "0" else None
        return ids.to(device, non_blocking=True), seg, mask.to(device, non_blocking=True), lens

    def forward(self, sents, feats, visual_attention_mask=None):
        input_ids, segment_ids, input_mask, lengths = self._tokenize(sents, feats[0].device)
        return self.model(input_ids, segment_ids, input_mask, visual_feats=feats, visual_attention_mask=visual_attention_mask, token_lengths=lengths)

    def forward_with_head(self, sents, feats):
        input_ids, segment_ids, input_mask, lengths = self._tokenize(sents, feats[0].device)
        return self.model.forward_with_head(input_ids, segment_ids, input_mask, feats, token_lengths=lengths)

    def save(self, path):
        torch.save(self.model.state_dict(), os.path.join("%s_LXRT.pth" % path))

    def load(self, path):
        print("Load LXMERT pre-trained model from %s" % path)
        state_dict = torch.load("%s_LXRT.pth" % path, map_location="cpu")
        state_dict = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
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
