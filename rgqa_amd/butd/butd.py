"""Mirror of the reference's butd/butd.py `GQABUTD(num_answers, dictionary, dropout=True)` (108-221) on the HIP engine
(rgqa_config.arch = 1): same constructor, `tokenize`, `forward(feat, pos, sent, attention=False)` and state_dict keys
(`w_emb.emb.weight`, `q_enc.rnn.weight_ih_l0`, `att.image_proj.mlp.0.weight_g/_v`, `ans_classifier.3.bias`, ...).
The module tree only carries parameters (views of the engine's flat arena); GRU, weight-norm projections, the
attention over the 36 regions and the classifier run in librgqa_hip.so."""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ..engine import Engine
from ..lxrt.modeling import ArenaBinding, _EngineFunction

MAX_GQA_LENGTH = 40


class GQABUTD(nn.Module):
    def __init__(self, num_answers, dictionary, dropout=True, precision=None):
        super().__init__()
        self.num_answers = num_answers
        self.dictionary = dictionary
        self.emb_dim, self.hidden, self.v_dim = 300, 1024, 2048
        self.attention_dropout, self.answer_dropout = (0.2, 0.5) if dropout else (0.0, 0.0)
        self.precision = precision or os.environ.get("RGQA_PRECISION", "bf16")
        self._binding = ArenaBinding()
        self._fwd_counter = 0
        self._seed_base = None
        self.__dict__["_anchor"] = None
        e = Engine(arch=1, vocab_size=dictionary.ntoken + 1, hidden=self.hidden, emb_dim=self.emb_dim, feat_dim=self.v_dim, pos_dim=4,
                   num_answers=num_answers, precision=self.precision, hidden_dropout=self.answer_dropout, attn_dropout=self.attention_dropout,
                   heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0)
        # parameter-only module tree with the reference's names; defaults as torch gives the reference's modules:
        # Linear / GRU U(-1/sqrt(fan), 1/sqrt(fan)), weight_g = ||V||_F (weight_norm init), Embedding N(0,1) with a zero padding row
        named = {}
        for sp in e.specs:
            parts = sp.name.split(".")
            cur = self
            for part in parts[:-1]:
                if part not in cur._modules:
                    cur.add_module(part, nn.Module())
                cur = cur._modules[part]
            p = nn.Parameter(torch.empty(sp.shape))
            cur.register_parameter(parts[-1], p)
            named[sp.name] = p
        with torch.no_grad():
            for name, p in named.items():
                if name == "w_emb.emb.weight":
                    p.normal_(0.0, 1.0)
                    p[dictionary.ntoken].zero_()
                elif "rnn." in name:
                    k = 1.0 / math.sqrt(self.hidden)
                    p.uniform_(-k, k)
                elif name.endswith("weight_v"):
                    k = 1.0 / math.sqrt(p.shape[1])
                    p.uniform_(-k, k)
                    named[name[:-1] + "g"].copy_(p.norm())
                elif name.endswith(".bias"):
                    k = 1.0 / math.sqrt(named[name[:-4] + "weight_v"].shape[1])
                    p.uniform_(-k, k)
        self._binding.bind(e, named.items())

    def load_embeddings(self, weights):
        """GloVe rows for the dictionary words (reference WordEmbedding.load_embeddings, butd.py:39-42)."""
        w = dict(self.named_parameters())["w_emb.emb.weight"]
        assert weights.shape == (self.dictionary.ntoken, self.emb_dim)
        with torch.no_grad():
            w[: self.dictionary.ntoken].copy_(torch.from_numpy(weights))

    def tokenize(self, sentences):
        """Tokenize and FRONT-pad to 40 (reference butd.py:180-193)."""
        rows = []
        pad = self.dictionary.padding_idx
        for sentence in sentences:
            tokens = self.dictionary.tokenize(sentence, False)[:MAX_GQA_LENGTH]
            tokens = [pad] * (MAX_GQA_LENGTH - len(tokens)) + tokens
            assert len(tokens) == MAX_GQA_LENGTH, "Tokenized & Padded Question != Max Length!"
            rows.append(tokens)
        return torch.from_numpy(np.asarray(rows, dtype=np.int64))

    # engine plumbing shared with the LXMERT modules (lxrt/modeling.py)
    def _engine_forward(self, feats, boxes, toks, mask, seg, train):
        b = self._binding
        if b.engine._params is None or b.engine.device != feats.device or not b.packed():
            b.materialize(feats.device)
        e = b.engine
        e.ensure_shape(feats.shape[0], toks.shape[1], feats.shape[1])
        if self._seed_base is None:
            self._seed_base = int(torch.initial_seed()) & 0x7FFFFFFFFFFF
        self._fwd_counter += 1
        return e.forward(feats, boxes, toks, toks, None, train=train, seed=self._seed_base + 7919 * self._fwd_counter)

    def forward(self, feat, pos, sent, attention=False):
        if feat.device.type != "cuda":
            raise RuntimeError("rgqa_amd: inputs must be on the MI355X (device 'cuda'); there is no CPU path")
        toks = self.tokenize(sent).to(feat.device)
        feat, pos = feat.contiguous().float(), pos.contiguous().float()
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != feat.device:
                self.__dict__["_anchor"] = torch.zeros(1, device=feat.device, requires_grad=True)
            logits, _ = _EngineFunction.apply(self._anchor, self, feat, pos, toks, toks, None, True)
        else:
            lg, _ = self._engine_forward(feat, pos, toks, toks, None, train=self.training)
            logits = lg.clone()
        if not attention:
            return logits
        B, O = feat.shape[0], feat.shape[1]
        return logits, self._binding.engine.activation("att", B, O).view(B, O, 1)
