"""UNITER backbone: oracle vs the vectors the reference's uniter/modeling.py produced (g11), the HIP engine
(arch 2) vs the same vectors in both row layouts, and the drop-in GQAUNITER surface."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import lxmert_ref as R
from oracle import uniter_ref as U
from rgqa_amd.synth import U_SMALL, U_FULL, uniter_batch, sample_idx
from rgqa_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {("small", 5): (3, 6, 91), ("small", 8): (3, 6, 92), ("full", 20): (4, 36, 93)}


def load_params(cfg, requires_grad=False):
    P = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(U.param_shapes(cfg)).items()}
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P


def case(tag, T):
    cfgd = U_SMALL if tag == "small" else U_FULL
    B, O, seed = CASES[(tag, T)]
    return cfgd, uniter_batch(cfgd, T, B, O, seed)


@pytest.mark.parametrize("tag,T", [("small", 5), ("small", 8), ("full", 20)])
def test_oracle_matches_reference_vectors(golden_dir, tag, T):
    g = np.load(os.path.join(golden_dir, "g11_uniter_%s_T%d.npz" % (tag, T)))
    cfgd, b = case(tag, T)
    cfg = R.RefConfig(**cfgd)
    P = load_params(cfg, True)
    t = {k: torch.from_numpy(v) for k, v in b.items() if k != "lengths"}
    assert np.array_equal(g["input_ids"], b["input_ids"])
    trace = {}
    logits, pooled = U.gqa_forward(P, cfg, t["feats"], t["pos7"], t["input_ids"], t["input_mask"], t["segment_ids"], trace)
    loss = R.bce_loss(logits, t["target"])
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=5e-6 if tag == "small" else 5e-5)
    np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    if tag == "small":
        for k in [k for k in g.files if k.startswith("act.")]:
            np.testing.assert_allclose(trace[k[4:]].detach().numpy(), g[k], rtol=0, atol=5e-6, err_msg=k)
        for k, p in P.items():
            ref = g["grad." + k]
            np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-6 + 1e-5 * np.abs(ref).max(), err_msg=k)
    else:
        off = 0
        for k, n in zip(g["grad_names"].tolist(), g["grad_counts"].tolist()):
            gr = P[k].grad.numpy().reshape(-1)
            ref = g["grad_samples"][off:off + n]; off += n
            np.testing.assert_allclose(gr[sample_idx(k, gr.size)], ref, rtol=1e-3, atol=1e-6 + 1e-4 * np.abs(gr).max(), err_msg=k)


def make_engine(cfgd, precision, dropout=0.0):
    from rgqa_amd.engine import Engine
    e = Engine(precision=precision, hidden_dropout=dropout, attn_dropout=dropout, arch=2, **cfgd).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    return e


def dev(b):
    return {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}


@pytest.mark.gpu
def test_engine_state_dict_contract():
    e = make_engine(U_SMALL, "f32")
    shapes = U.param_shapes(R.RefConfig(**U_SMALL))
    assert {sp.name: tuple(sp.shape) for sp in e.specs} == {k: tuple(v) for k, v in shapes.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("tag,T", [("small", 5), ("small", 8)])
def test_engine_f32_small_vs_golden(golden_dir, tag, T, packed):
    g = np.load(os.path.join(golden_dir, "g11_uniter_%s_T%d.npz" % (tag, T)))
    cfgd, raw = case(tag, T)
    b = dev(raw)
    B, O = raw["feats"].shape[:2]
    lens = [int(v) for v in raw["input_mask"].sum(1)] if packed else None
    e = make_engine(cfgd, "f32")
    e.ensure_shape(B, T, O)
    e.sync_weights()
    lg, pl = e.forward(b["feats"], b["pos7"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)
    np.testing.assert_allclose(lg.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(pl.cpu().numpy(), g["pooled"], rtol=0, atol=1e-4)
    loss = e.loss_backward(b["target"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for sp in e.specs:
        ref = g["grad." + sp.name]
        got = e.view(e.grads, sp).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-6 + 2e-4 * np.abs(ref).max(), err_msg=sp.name)
    # the other layout on the same engine (row maps are rebuilt), then back
    lens2 = None if packed else [int(v) for v in raw["input_mask"].sum(1)]
    lg2, _ = e.forward(b["feats"], b["pos7"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens2)
    np.testing.assert_allclose(lg2.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)
    lg3, _ = e.forward(b["feats"], b["pos7"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)
    np.testing.assert_allclose(lg3.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,packed", [("f32", False), ("f32", True), ("bf16", True), ("bf16x3", True), ("bf16x3", False), ("bf16x3_fwd", True)])
def test_engine_full_config_vs_golden(golden_dir, prec, packed):
    """bert-base UNITER: 12 layers over 56-token sequences. f32 and split-f32 (bf16x3) operands: logits within the north-star 1e-3 of the
    reference's CPU path; bf16: within the tolerance the LXMERT bf16 path is held to."""
    g = np.load(os.path.join(golden_dir, "g11_uniter_full_T20.npz"))
    cfgd, raw = case("full", 20)
    b = dev(raw)
    lens = [int(v) for v in raw["input_mask"].sum(1)] if packed else None
    e = make_engine(cfgd, prec)
    e.ensure_shape(4, 20, 36)
    e.sync_weights()
    lg, pl = e.forward(b["feats"], b["pos7"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)
    err = np.abs(lg.cpu().numpy() - g["logits"]).max()
    print("uniter full config %s packed=%s: max |logit - reference| %.3e" % (prec, packed, err))
    assert err <= (6e-2 if prec == "bf16" else 1e-3), err
    perr = np.abs(pl.cpu().numpy() - g["pooled"]).max()
    np.testing.assert_allclose(pl.cpu().numpy(), g["pooled"], rtol=0, atol=2e-4 if prec == "f32" else 3e-2)
    loss = e.loss_backward(b["target"])
    print("uniter full %s packed=%s: logits max %.3e pooled max %.3e loss rel %.3e grad norm rel %.3e" % (prec, packed, err, perr, abs(loss.item() - g["loss"]) / abs(g["loss"]),
          abs(e.grad_norm().item() - g["grad_norm"]) / g["grad_norm"]))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4 if prec == "f32" else 2e-3)
    np.testing.assert_allclose(e.grad_norm().item(), g["grad_norm"], rtol=1e-3 if prec == "f32" else 3e-2)
    if prec == "f32":
        byname = {sp.name: sp for sp in e.specs}
        off = 0
        for k, n in zip(g["grad_names"].tolist(), g["grad_counts"].tolist()):
            gr = e.view(e.grads, byname[k]).cpu().numpy().reshape(-1)
            ref = g["grad_samples"][off:off + n]; off += n
            np.testing.assert_allclose(gr[sample_idx(k, gr.size)], ref, rtol=5e-3, atol=1e-6 + 5e-4 * np.abs(gr).max(), err_msg=k)


@pytest.mark.gpu
def test_engine_train_mode_dropout_consistent_between_layouts():
    """Dropout streams are indexed by (site, local row): the packed and the padded layout drop different elements only where the
    row numbering differs, so the test is statistical - finite loss, non-zero gradients everywhere, run-to-run reproducible."""
    cfgd, raw = case("small", 8)
    b = dev(raw)
    lens = [int(v) for v in raw["input_mask"].sum(1)]
    e = make_engine(cfgd, "bf16", dropout=0.1)
    e.ensure_shape(3, 8, 6)
    e.sync_weights()
    out = []
    for _ in range(2):
        e.forward(b["feats"], b["pos7"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=5, lengths=lens)
        loss = e.loss_backward(b["target"])
        torch.cuda.synchronize()
        out.append((loss.item(), e.grads.clone()))
    assert np.isfinite(out[0][0]) and abs(out[0][0] - out[1][0]) <= 1e-6 * abs(out[0][0])      # the scalar loss is an atomic sum over blocks
    live = [sp for sp in e.specs if "_embeddings.weight" not in sp.name]
    for sp in live:
        a, c = out[0][1][sp.offset:sp.offset + sp.numel], out[1][1][sp.offset:sp.offset + sp.numel]
        assert torch.equal(a, c), sp.name
        assert float(a.abs().max()) > 0, sp.name
    with pytest.raises(RuntimeError):
        e.cross_attention(0, "l2v")


@pytest.mark.gpu
def test_dropin_gqauniter_matches_oracle(golden_dir, monkeypatch):
    """`from uniter.uniter import GQAUNITER` driven like the reference's trainer: forward(feat, pos, sent), BCE, backward; state_dict
    keys are the reference's; logits and a weight gradient match the oracle."""
    monkeypatch.setenv("RGQA_BERT_VOCAB", os.path.join(golden_dir, "g4_vocab.txt"))
    monkeypatch.setenv("RGQA_PRECISION", "f32")
    monkeypatch.setenv("RGQA_UNITER_ANY_ROIS", "1")
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        import rgqa_amd.uniter.modeling as M
        cfgd = dict(U_SMALL, vocab_size=80, max_pos=64)
        monkeypatch.setattr(M.VISUAL_CONFIG, "visual_feat_dim", cfgd["feat_dim"])
        monkeypatch.setattr(M.UniterFeatureExtraction, "from_pretrained", classmethod(
            lambda cls, name, **kw: cls(M.BertConfig(cfgd["vocab_size"], hidden_size=cfgd["hidden"], num_hidden_layers=cfgd["l_layers"],
                                                     num_attention_heads=cfgd["heads"], intermediate_size=cfgd["inter"],
                                                     max_position_embeddings=cfgd["max_pos"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), **kw)))
        from uniter.uniter import GQAUNITER
        m = GQAUNITER(cfgd["num_answers"], model_args=types.SimpleNamespace(from_scratch=False))
        cfg = R.RefConfig(**cfgd)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v) for k, v in U.param_shapes(cfg).items()}
        filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
        m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})
        m = m.cuda().eval()
        sents = ["What color is the dog?", "is it", "who is holding the red bottle on the left side of the table near the window today"]
        raw = uniter_batch(cfgd, 20, 3, 6, 55)
        feats, pos7, target = torch.from_numpy(raw["feats"]), torch.from_numpy(raw["pos7"]), torch.from_numpy(raw["target"])
        logit = m(feats.cuda(), pos7.cuda(), sents)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)
        loss.backward()
        ids, mask, _ = R.sents_to_features(sents, 20, m.encoder.tokenizer.vocab)
        P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in filled.items()}
        lg, _ = U.gqa_forward(P, cfg, feats, pos7, torch.tensor(ids), torch.tensor(mask))
        R.bce_loss(lg, target).backward()
        np.testing.assert_allclose(logit.detach().cpu().numpy(), lg.detach().numpy(), rtol=0, atol=1e-4)
        for w in ("encoder.model.uniter.encoder.layer.1.attention.self.value.weight", "encoder.model.uniter.img_embeddings.pos_linear.weight",
                  "encoder.model.uniter.embeddings.token_type_embeddings.weight", "logit_fc.3.weight"):
            got = dict(m.named_parameters())[w].grad.cpu().numpy()
            ref = P[w].grad.numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-6 + 2e-4 * np.abs(ref).max(), err_msg=w)
    finally:
        sys.path.remove(os.path.join(ROOT, "dropin"))
