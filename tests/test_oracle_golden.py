"""Oracle (oracle/lxmert_ref.py) pinned against fixtures produced by the reference itself
(oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import lxmert_ref as R
from rgqa_amd.synth import SMALL, FULL, small_batch, full_batch, sample_idx
from rgqa_amd import synth


def load_params(cfg, requires_grad=False):
    shapes = R.param_shapes(cfg)
    P = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes).items()}
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P


def to_t(b):
    return {k: torch.from_numpy(v) for k, v in b.items()}


@pytest.mark.parametrize("T", [5, 8])
def test_g1_small_full_trace_and_grads(golden_dir, T):
    g = np.load(os.path.join(golden_dir, "g1_small_T%d.npz" % T))
    cfg = R.RefConfig(**SMALL)
    P = load_params(cfg, True)
    b = to_t(small_batch(T))
    assert np.array_equal(g["input_ids"], b["input_ids"].numpy())
    feats = b["feats"].clone().requires_grad_(True)
    boxes = b["boxes"].clone().requires_grad_(True)
    trace = {}
    logits, pooled = R.gqa_forward(P, cfg, feats, boxes, b["input_ids"], b["input_mask"], b["segment_ids"], trace)
    loss = R.bce_loss(logits, b["target"])
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    for k in [k for k in g.files if k.startswith("act.")]:
        np.testing.assert_allclose(trace[k[4:]].detach().numpy(), g[k], rtol=0, atol=5e-6, err_msg=k)
    dead = set(g["dead"].tolist())
    for k, p in P.items():
        if k in dead:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            ref = g["grad." + k]
            np.testing.assert_allclose(p.grad.numpy(), ref, rtol=1e-4, atol=1e-6 + 1e-5 * np.abs(ref).max(), err_msg=k)
    np.testing.assert_allclose(feats.grad.numpy(), g["dfeats"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(boxes.grad.numpy(), g["dboxes"], rtol=1e-4, atol=1e-7)
    # the 16 dead tensors are exactly x_layers.<last>.visn_* 
    assert len(dead) == 16 and all(".x_layers.1.visn_" in k for k in dead)


@pytest.mark.parametrize("T", [20, 30])
def test_g2_full_config(golden_dir, T):
    path = os.path.join(golden_dir, "g2_full_T%d.npz" % T)
    if not os.path.exists(path):
        pytest.skip("full fixture not generated")
    g = np.load(path)
    cfg = R.RefConfig(**FULL)
    P = load_params(cfg, True)
    b = to_t(full_batch(T))
    assert np.array_equal(g["input_ids"], b["input_ids"].numpy())
    trace = {}
    logits, pooled = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], None, trace)
    loss = R.bce_loss(logits, b["target"])
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(pooled.detach().numpy(), g["pooled"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    names = g["act_names"].tolist()
    for i, k in enumerate(names):
        a = trace[k].detach().numpy()
        np.testing.assert_allclose(np.sqrt((a.astype(np.float64) ** 2).sum()), g["act_l2"][i], rtol=1e-5, err_msg=k)
        np.testing.assert_allclose(a.reshape(-1)[:16], g["act_first"][i], rtol=0, atol=5e-5, err_msg=k)
    dead = set(g["dead"].tolist())
    off, sq = 0, 0.0
    for k, n in zip(g["grad_names"].tolist(), g["grad_counts"].tolist()):
        gr = P[k].grad.numpy()
        sq += float((gr.astype(np.float64) ** 2).sum())
        ref = g["grad_samples"][off:off + n]
        off += n
        np.testing.assert_allclose(gr.reshape(-1)[sample_idx(k, gr.size)], ref, rtol=2e-3,
                                   atol=1e-7 + 2e-5 * np.abs(gr).max(), err_msg=k)
    np.testing.assert_allclose(np.sqrt(sq), g["grad_norm"], rtol=1e-4)
    assert len(dead) == 16
    for k in dead:
        assert P[k].grad is None or float(P[k].grad.abs().max()) == 0.0


def test_g3_bertadam(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_bertadam.npz"))
    shapes = {"a": (7, 5), "b": (13,), "c": (3, 4, 2), "d": (1,), "e": (6,)}
    ps = {k: torch.from_numpy(synth.uniform("adam.p." + k, s, -1, 1)) for k, s in shapes.items()}
    opt = R.BertAdamRef(list(ps.values()), lr=1e-2, warmup=0.1, t_total=10)
    for step in range(3):
        grads = []
        for k in shapes:
            if k == "e" or (k == "d" and step == 0):
                grads.append(None)
            else:
                grads.append(torch.from_numpy(synth.uniform("adam.g%d.%s" % (step, k), shapes[k], -2, 2)))
        opt.step(grads)
        for k, p in ps.items():
            np.testing.assert_allclose(p.numpy(), g["p%d.%s" % (step, k)], rtol=1e-6, atol=1e-7, err_msg="%d %s" % (step, k))
    # first update has lr 0 under warmup (step counter read before increment)
    np.testing.assert_array_equal(g["p0.a"], synth.uniform("adam.p.a", (7, 5), -1, 1))


def test_g4_tokenizer(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g4_tokenizer.json"), encoding="utf-8"))
    vocab = {w.rstrip("\n"): i for i, w in enumerate(open(os.path.join(golden_dir, "g4_vocab.txt"), encoding="utf-8"))}
    for T in (20, 30):
        ids, mask, seg = R.sents_to_features(g["sentences"], T, vocab)
        assert ids == g["T%d" % T]["input_ids"]
        assert mask == g["T%d" % T]["input_mask"]
        assert seg == g["T%d" % T]["segment_ids"]


@pytest.mark.parametrize("mode", ["mixup_v1", "mixup_v2", "mixup_v3"])
def test_g5_mixup(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "g5_mixup.npz"))
    B, O, Fd, NA = 6, 36, 8, 5
    feats = torch.from_numpy(synth.uniform("mix.f", (B, O, Fd), 0, 1))
    boxes = torch.from_numpy(synth.uniform("mix.b", (B, O, 4), 0, 1))
    target = torch.from_numpy(synth.uniform("mix.t", (B, NA), 0, 1))
    prop = g[mode + ".prop"]
    idx = [g[mode + ".perm"][j][: int(prop[j] * O)] for j in range(B)]
    f, b, t = R.roi_mixup(feats, boxes, target, g[mode + ".partner"], prop, idx, mode)
    np.testing.assert_array_equal(f.numpy(), g[mode + ".feats"])
    np.testing.assert_array_equal(b.numpy(), g[mode + ".boxes"])
    np.testing.assert_allclose(t.numpy(), g[mode + ".target"], rtol=1e-6)


def test_g5_perturb_and_weighted_sum(golden_dir):
    """the other batch constructions of gqa_mixup_vis.py (:124-133, :217-244) against the reference's own statements (G5)"""
    g = np.load(os.path.join(golden_dir, "g5_mixup.npz"))
    B, O, Fd, NA = 6, 36, 8, 5
    feats = torch.from_numpy(synth.uniform("mix.f", (B, O, Fd), 0, 1))
    boxes = torch.from_numpy(synth.uniform("mix.b", (B, O, 4), 0, 1))
    target = torch.from_numpy(synth.uniform("mix.t", (B, NA), 0, 1))
    f, b, t = R.perturb_batch(feats, boxes, target, g["perturb.perm"])
    for got, key in ((f, "feats"), (b, "boxes"), (t, "target")):
        np.testing.assert_array_equal(got.numpy(), g["perturb." + key])
    for mode in ("weighted_sum_v1", "weighted_sum_v2"):
        f, b, t = R.weighted_sum_batch(feats, boxes, target, g[mode + ".partner"], g[mode + ".prop"], mode)
        np.testing.assert_array_equal(f.numpy(), g[mode + ".feats"])
        np.testing.assert_array_equal(b.numpy(), g[mode + ".boxes"])
        np.testing.assert_array_equal(t.numpy(), g[mode + ".target"])


def test_bce_and_uq_column():
    """G6: BCE x NA with an all-zero (pseudo-UQ) row and the dropped UQ column (gqa_conf.py:153,197-198)."""
    z = torch.from_numpy(synth.uniform("bce.z", (4, 7), -3, 3))
    t = torch.zeros(4, 8)
    t[0, 2], t[1, 7], t[3, 5] = 1.0, 1.0, 0.6   # row 1 answers 'UQ' -> all-zero after the drop; row 2 empty
    tt = R.drop_uq_column(t)
    assert tt.shape == (4, 7) and float(tt[1].sum()) == 0.0
    loss = R.bce_loss(z, tt)
    man = (torch.clamp(z, min=0) - z * tt + torch.log1p(torch.exp(-z.abs()))).sum() / 4
    np.testing.assert_allclose(loss.item(), man.item(), rtol=1e-6)


def butd_fill(cfgb):
    from oracle import butd_ref as BR
    out = {}
    for k, shp in BR.param_shapes(cfgb).items():
        if k.endswith("weight_g"):
            out[k] = np.asarray(1.5 + 0.5 * synth.uniform(k, (1,), -1, 1)[0], dtype=np.float32)
        elif k == "w_emb.emb.weight":
            w = synth.uniform(k, shp, -0.5, 0.5)
            w[-1] = 0.0
            out[k] = w
        else:
            out[k] = synth.uniform(k, shp, -0.05, 0.05)
    return out


def test_g7_butd(golden_dir):
    """BUTD oracle (oracle/butd_ref.py) vs the reference's GQABUTD run in the build container."""
    from oracle import butd_ref as BR
    from rgqa_amd.synth import BUTD_WORDS, BUTD_SENTS
    g = np.load(os.path.join(golden_dir, "g7_butd.npz"))
    c = BR.ButdConfig(ntoken=len(BUTD_WORDS), num_answers=23)
    P = {k: torch.from_numpy(np.asarray(v)).requires_grad_(True) for k, v in butd_fill(c).items()}
    word2idx = {w: i for i, w in enumerate(BUTD_WORDS)}
    toks = torch.tensor(BR.tokenize(BUTD_SENTS, word2idx))
    assert np.array_equal(toks.numpy(), g["toks"])
    b = synth.synth_batch(len(BUTD_SENTS), 8, O=36, F=2048, NA=23, vocab=64, seed=606, uq_frac=0.2)
    logits, att = BR.butd_forward(P, c, torch.from_numpy(b["feats"]), torch.from_numpy(b["boxes"]), toks, want_att=True)
    loss = R.bce_loss(logits, torch.from_numpy(b["target"]))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(att.detach().numpy(), g["att"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for k, p in P.items():
        gr = p.grad.numpy()
        np.testing.assert_allclose(np.sqrt((gr.astype(np.float64) ** 2).sum()), g["gnorm." + k], rtol=2e-4, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(gr.reshape(-1)[sample_idx(k, gr.size)], g["gsamp." + k], rtol=2e-3, atol=1e-7 + 1e-4 * np.abs(gr).max(), err_msg=k)
    assert float(P["w_emb.emb.weight"].grad[-1].abs().max()) == 0.0     # padding row gets no gradient


@pytest.mark.parametrize("tag,T", [("small", 5), ("small", 8), ("full", 20)])
def test_g8_cross_attention_probabilities(golden_dir, tag, T):
    """Oracle cross-attention probabilities vs the reference's lxrt_vis variant (`output_attention=True`); the same
    fixture's pooled output must equal the plain lxrt goldens, i.e. lxrt_vis computes what lxrt computes."""
    g = np.load(os.path.join(golden_dir, "g8_xatt.npz"))
    cfgd = SMALL if tag == "small" else FULL
    cfg = R.RefConfig(**cfgd)
    P = load_params(cfg)
    b = to_t(small_batch(T) if tag == "small" else full_batch(T))
    trace = {}
    with torch.no_grad():
        _, pooled = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], trace)
    pre = "%s_T%d." % (tag, T)
    np.testing.assert_allclose(pooled.numpy(), g[pre + "pooled"], rtol=0, atol=5e-6)
    ref_file = "g1_small_T%d.npz" % T if tag == "small" else "g2_full_T%d.npz" % T
    np.testing.assert_allclose(g[pre + "pooled"], np.load(os.path.join(golden_dir, ref_file))["pooled"], rtol=0, atol=5e-6)
    keys = [k for k in g.files if k.startswith(pre + "x")]
    assert len(keys) == (2 * cfgd["x_layers"] if tag == "small" else 4)
    for k in keys:
        got = trace[k[len(pre):]].numpy()
        assert got.shape == g[k].shape
        np.testing.assert_allclose(got, g[k], rtol=0, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)
