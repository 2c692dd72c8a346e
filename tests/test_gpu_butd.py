"""BUTD path (BASELINE config 5) on the HIP engine vs the golden fixture produced by the reference's
own GQABUTD (tests/golden/g7_butd.npz) and vs the oracle, through the drop-in class."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from rgqa_amd.synth import BUTD_WORDS, BUTD_SENTS, sample_idx     # noqa: E402
from rgqa_amd import synth                                           # noqa: E402


def build(precision, dropout=False):
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        from butd.butd import GQABUTD
        from butd.preprocess import Dictionary
    finally:
        sys.path.pop(0)
    from tests.test_oracle_golden import butd_fill
    from oracle import butd_ref as BR
    d = Dictionary()
    for w in BUTD_WORDS:
        d.add_word(w)
    m = GQABUTD(23, d, dropout=dropout, precision=precision)
    c = BR.ButdConfig(ntoken=len(BUTD_WORDS), num_answers=23)
    filled = butd_fill(c)
    assert set(m.state_dict()) == set(filled) and all(tuple(m.state_dict()[k].shape) == tuple(np.shape(filled[k])) for k in filled)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in filled.items()})
    return m.cuda(), c, filled


@pytest.mark.parametrize("precision,tol,gtol", [("f32", 2e-4, 3e-3), ("bf16", 8e-2, 6e-2), ("bf16x3", 1e-3, 3e-3)])      # bf16x3 (round 4): inside the north star's 1e-3
def test_butd_vs_reference_golden(golden_dir, precision, tol, gtol):
    g = np.load(os.path.join(golden_dir, "g7_butd.npz"))
    m, c, filled = build(precision)
    b = synth.synth_batch(len(BUTD_SENTS), 8, O=36, F=2048, NA=23, vocab=64, seed=606, uq_frac=0.2)
    feat, pos, target = (torch.from_numpy(b[k]).cuda() for k in ("feats", "boxes", "target"))
    assert np.array_equal(m.tokenize(BUTD_SENTS).numpy(), g["toks"])
    m.eval()
    logits, att = m(feat, pos, BUTD_SENTS, attention=True)
    assert att.shape == (len(BUTD_SENTS), 36, 1)
    print("butd %s: logits max err %.3e, att max err %.3e" % (precision, np.abs(logits.detach().cpu().numpy() - g["logits"]).max(), np.abs(att.cpu().numpy() - g["att"]).max()))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=0, atol=tol)
    np.testing.assert_allclose(att.cpu().numpy(), g["att"], rtol=0, atol=tol / 4)
    loss = torch.nn.BCEWithLogitsLoss()(logits, target) * logits.size(1)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=50 * tol / 15)
    loss.backward()
    wn = ws = 0.0
    for k, p in m.named_parameters():
        gr = p.grad.cpu().numpy()
        ref_norm = float(g["gnorm." + k])
        wn = max(wn, abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - ref_norm) / max(ref_norm, 1e-12))
        ws = max(ws, float(np.abs(gr.reshape(-1)[sample_idx(k, gr.size)] - g["gsamp." + k]).max() / max(np.abs(g["gsamp." + k]).max(), 1e-6)))
        assert abs(np.sqrt((gr.astype(np.float64) ** 2).sum()) - ref_norm) <= gtol * ref_norm + 1e-7, k
        ref = g["gsamp." + k]
        got = gr.reshape(-1)[sample_idx(k, gr.size)]
        assert np.abs(got - ref).max() <= 3 * gtol * max(np.abs(ref).max(), 1e-6) + 1e-7, k
    print("butd %s: worst gradient-norm rel err %.3e, worst sampled-entry err / max %.3e, loss rel %.3e" % (precision, wn, ws, abs(loss.item() - g["loss"]) / abs(g["loss"])))
    assert float(dict(m.named_parameters())["w_emb.emb.weight"].grad[-1].abs().max()) == 0.0


@pytest.mark.parametrize("precision,tol", [("f32", 5e-4), ("bf16x3", 1e-3)])
def test_butd_weight_norm_follows_the_weights_over_several_steps(precision, tol):
    """Several forward / backward passes at hidden 1024 against the oracle: the weight-norm scale ||V||_F must be re-taken from
    the CURRENT weights on every forward pass.  Round 3's one-launch sum-of-squares kept its ticket word in the scratch that backward's
    column sums also use, so from the second training step on the norm stayed at its first value.  Between the passes every weight_v grows
    by 25 % and the biases move (the same edit on both sides, after a real backward pass has used the scratch): W = g V / ||V|| is unchanged
    by the scaling only if the norm is taken again."""
    from oracle import butd_ref as BR
    m, c, filled = build(precision)
    m.eval()          # dropout off; gradients still flow
    b = synth.synth_batch(len(BUTD_SENTS), 8, O=36, F=2048, NA=23, vocab=64, seed=606, uq_frac=0.2)
    feat, pos, target = (torch.from_numpy(b[k]) for k in ("feats", "boxes", "target"))
    toks = m.tokenize(BUTD_SENTS)
    P = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in filled.items()}
    fg, pg, tg = feat.cuda(), pos.cuda(), target.cuda()
    first = None
    for step in range(4):
        lo = BR.butd_forward(P, c, feat, pos, toks)
        lg = m(fg, pg, BUTD_SENTS)
        err = float((lg.detach().cpu() - lo).abs().max())
        print("butd %s pass %d: logits max err %.3e (|logits| max %.3e)" % (precision, step, err, float(lo.abs().max())))
        assert err < tol, (step, err)
        first = lo if first is None else first
        loss = torch.nn.BCEWithLogitsLoss()(lg, tg) * lg.size(1)
        m.zero_grad()
        loss.backward()          # the engine's backward pass runs its column sums through the shared scratch
        with torch.no_grad():
            for k, p_ in m.named_parameters():
                if k.endswith("weight_v"):
                    p_ *= 1.25; P[k] *= 1.25
                elif k.endswith(".bias"):
                    p_ += 0.01; P[k] += 0.01
    assert float((lo - first).abs().max()) > 10 * tol       # the passes are not all the same computation (the biases moved)


@pytest.mark.parametrize("precision,ltol,lrel,gtol", [("bf16x3", 1e-3, 1e-2, 3e-3), ("bf16", 1e-2, 1.5e-1, 8e-2), ("f32", 5e-4, 2e-3, 3e-3)])
def test_butd_step_at_config5_size_vs_oracle(precision, ltol, lrel, gtol):
    """BASELINE config 5 at its per-GPU size: B = 256 QA pairs, 40 front-padded tokens over a dictionary of 3000 words, 36
    RoIs, 1842 answers, through the engine as bench.py --butd drives it (rgqa_config.arch = 1).  Forward: the logits of ALL 256 rows and the
    loss against the oracle (butd/butd.py:195-221 restated; the BUTD model is ~0.5 GFLOP per sample, the host does all rows in seconds).
    Backward: every gradient tensor against the oracle's on the full batch."""
    from oracle import butd_ref as BR
    from rgqa_amd.engine import Engine
    from tests.test_oracle_golden import butd_fill
    B, L, O, NA, NT = 256, 40, 36, 1842, 3000
    c = BR.ButdConfig(ntoken=NT, num_answers=NA)
    filled = butd_fill(c)
    e = Engine(arch=1, vocab_size=NT + 1, hidden=1024, emb_dim=300, feat_dim=2048, pos_dim=4, num_answers=NA, precision=precision,
               hidden_dropout=0.0, attn_dropout=0.0, heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0).allocate("cuda")
    assert {sp.name for sp in e.specs} == set(filled)
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(np.asarray(filled[sp.name]).reshape(sp.shape)))
    b = synth.synth_batch(B, L, seed=515, uq_frac=0.25, vocab=NT)
    rng = np.random.RandomState(3)
    toks = np.full((B, L), NT, dtype=np.int64)                     # front padding with the padding index (butd.py:180-193)
    for r in range(B):
        n = int(rng.randint(1, L + 1)) if r else L                # a full-length question and 1-token questions occur
        toks[r, L - n:] = rng.randint(0, NT, size=n)
    feat, pos, target, tk = torch.from_numpy(b["feats"]), torch.from_numpy(b["boxes"]), torch.from_numpy(b["target"]), torch.from_numpy(toks)
    e.ensure_shape(B, L, O)
    e.sync_weights()
    fg, pg, tg, kg = feat.cuda(), pos.cuda(), target.cuda(), tk.cuda()
    lg, _ = e.forward(fg, pg, kg, kg, None, train=False)
    lg = lg.clone()
    loss = e.loss_backward(tg).item()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    P = {k: torch.from_numpy(np.asarray(v)).clone().requires_grad_(True) for k, v in filled.items()}
    lo = BR.butd_forward(P, c, feat, pos, tk)
    lr = torch.nn.functional.binary_cross_entropy_with_logits(lo, target) * NA
    lr.backward()
    lr = lr.detach()
    err = (lg.cpu() - lo.detach()).abs()
    worst, wname, num, den = 0.0, "", 0.0, 0.0
    for sp in e.specs:
        got, ref = e.view(e.grads, sp).float().cpu(), P[sp.name].grad.reshape(sp.shape)
        dd, rr = float((got - ref).norm()), float(ref.norm())
        num += dd * dd
        den += rr * rr
        if rr > 1e-9 and dd / rr > worst:
            worst, wname = dd / rr, sp.name
    print("butd %s at B=256 / 40 tokens / 3000 words: logits max err %.3e mean %.3e, loss rel %.3e, gradients rel %.3e, worst tensor %s %.3e" % (
        precision, float(err.max()), float(err.mean()), abs(loss - lr.item()) / abs(lr.item()), (num / den) ** 0.5, wname, worst))
    assert float(err.max()) < min(ltol, lrel * float(lo.detach().abs().max())), (float(err.max()), float(lo.detach().abs().max()))     # (the filler's logits are small: |z| <= 0.05)
    assert abs(loss - lr.item()) < (2e-3 if precision == "bf16" else 5e-5) * abs(lr.item())
    assert (num / den) ** 0.5 < gtol / 2 and worst < gtol, (wname, worst)
    assert float(e.view(e.grads, [sp for sp in e.specs if sp.name == "w_emb.emb.weight"][0])[NT].abs().max()) == 0.0       # the padding row gets no gradient


@pytest.mark.parametrize("B", [70, 256, 5])
def test_butd_gru_forms_agree(B):
    """The question encoder's GRU (butd/butd.py:48-73) in its four forms - rgqa_debug_set key 18: 0 = one GEMM + one gate kernel per token from the
    host, 1 / 2 = persistent with 64-sample row groups x 16-unit slices (4 / 8 waves), 3 = persistent with 32-sample row groups x 32-unit
    slices and the weights in registers (default) - on batch sizes that end inside a row group: same logits and gradients to bf16 rounding (the
    persistent forms keep h W_hh^T in f32 between the MFMAs and the gates; they differ from each other only in summation order), two passes each
    so that a counter left behind by a launch would show."""
    from rgqa_amd import _lib
    from rgqa_amd.engine import Engine
    from oracle import butd_ref as BR
    from tests.test_oracle_golden import butd_fill
    L_ = _lib.load()
    L, O, NA, NT = 40, 36, 1842, 3000
    c = BR.ButdConfig(ntoken=NT, num_answers=NA)
    filled = butd_fill(c)
    b = synth.synth_batch(B, L, seed=99, uq_frac=0.25, vocab=NT)
    rng = np.random.RandomState(4)
    toks = np.full((B, L), NT, dtype=np.int64)
    for r in range(B):
        n = int(rng.randint(1, L + 1)) if r else L
        toks[r, L - n:] = rng.randint(0, NT, size=n)
    fg, pg, tg, kg = (torch.from_numpy(x).cuda() for x in (b["feats"], b["boxes"], b["target"], toks))
    res = {}
    try:
        for form in (0, 1, 2, 3):
            assert L_.rgqa_debug_set(18, form) == 0
            e = Engine(arch=1, vocab_size=NT + 1, hidden=1024, emb_dim=300, feat_dim=2048, pos_dim=4, num_answers=NA, precision="bf16",
                       hidden_dropout=0.0, attn_dropout=0.0, heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0).allocate("cuda")
            for sp in e.specs:
                e.view(e.params, sp).copy_(torch.from_numpy(np.asarray(filled[sp.name]).reshape(sp.shape)))
            e.ensure_shape(B, L, O)
            e.sync_weights()
            out = []
            for _ in range(2):
                lg = e.forward(fg, pg, kg, kg, None, train=False)[0].clone()
                e.loss_backward(tg)
                out.append((lg, e.grads.clone()))
            torch.cuda.synchronize()
            # every form is deterministic, the word-embedding gradient included (summed per table row in a fixed order since round 6)
            assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), form
            res[form] = out[0]
    finally:
        L_.rgqa_debug_set(18, 3)
    l0, g0 = res[0]
    assert torch.isfinite(l0).all() and torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    for form in (1, 2, 3):
        lf, gf = res[form]
        assert torch.isfinite(lf).all() and torch.isfinite(gf).all()
        assert float((lf - l0).abs().max()) < 2e-3 * max(1.0, float(l0.abs().max())), form
        assert float((gf - g0).norm()) < 3e-2 * float(g0.norm()), form
    # the two persistent families differ only in the order of the f32 sums
    assert float((res[3][0] - res[1][0]).abs().max()) < 5e-4 * max(1.0, float(l0.abs().max()))
    assert float((res[3][1] - res[1][1]).norm()) < 1e-2 * float(g0.norm())


def test_butd_train_step_runs_with_dropout():
    from rgqa_amd.lxrt.optimization import BertAdam
    m, c, _ = build("bf16", dropout=True)
    b = synth.synth_batch(len(BUTD_SENTS), 8, O=36, F=2048, NA=23, vocab=64, seed=607)
    feat, pos, target = (torch.from_numpy(b[k]).cuda() for k in ("feats", "boxes", "target"))
    opt = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=10)
    m.train()
    losses = []
    for _ in range(4):
        opt.zero_grad()
        logit = m(feat, pos, BUTD_SENTS)
        loss = torch.nn.BCEWithLogitsLoss()(logit, target) * logit.size(1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.)
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
