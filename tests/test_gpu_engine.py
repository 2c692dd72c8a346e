"""End-to-end parity of the HIP engine against the golden fixtures (produced by the reference itself) and the
oracle, through the C ABI.

Three precisions, one bound that matters: BASELINE.json's north star asks for logits within 1e-3 of the reference's f32 CPU path.
  f32      exact f32 FMA arithmetic on the vector ALU: inside the bound (observed ~1e-5); slow, kept as the on-device reference
  bf16x3   split-f32 operands, three bf16 MFMA products per f32 product: the FAST mode inside the bound (observed ~1e-4, printed)
  bf16     BASELINE config 3's mode ("fwd+bwd B=256 bf16"): bf16 operands through 19 blocks leave the logits ~5e-2 from the
           reference's - OUTSIDE the 1e-3 bound by construction.  Its tests therefore gate on what bf16 can promise: the mean
           logit error and the loss / gradient-norm scalars (sums whose rounding errors cancel) first, the element-wise maxima
           at <= 2x the observed values second."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from rgqa_amd.synth import SMALL, FULL, small_batch, full_batch, sample_idx   # noqa: E402
from rgqa_amd import synth                                                      # noqa: E402


def make_engine(cfgd, precision, dropout=0.0):
    from rgqa_amd.engine import Engine
    e = Engine(precision=precision, hidden_dropout=dropout, attn_dropout=dropout, **cfgd).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    return e


def dev(b):
    return {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}


def run(e, b, train=False, seed=0):
    lg, pl = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=train, seed=seed)
    return lg, pl


@pytest.mark.parametrize("T", [5, 8])
def test_f32_small_vs_golden(golden_dir, T):
    g = np.load(os.path.join(golden_dir, "g1_small_T%d.npz" % T))
    e = make_engine(SMALL, "f32")
    b = dev(small_batch(T))
    e.ensure_shape(3, T, 6)
    e.sync_weights()
    lg, pl = run(e, b)
    np.testing.assert_allclose(lg.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(pl.cpu().numpy(), g["pooled"], rtol=0, atol=1e-4)
    for k in [k for k in g.files if k.startswith("act.")]:
        name = k[4:]
        if name == "x1_visn":
            continue   # dead branch in mode 'x' (SURVEY.md §8 A11): not computed
        ref = g[k]
        if name == "x1_lang":      # final language output: only its [CLS] rows are consumed (pooler) and computed
            got = e.activation(name, ref.shape[0]).cpu().numpy()
            np.testing.assert_allclose(got, ref[:, 0, :], rtol=0, atol=1e-4, err_msg=name)
            continue
        got = e.activation(name, ref.shape[0] * ref.shape[1]).cpu().numpy().reshape(ref.shape)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4, err_msg=name)
    loss = e.loss_backward(b["target"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    dead = set(g["dead"].tolist())
    for sp in e.specs:
        got = e.view(e.grads, sp).cpu().numpy()
        if sp.name in dead:
            assert sp.dead and float(np.abs(got).max()) == 0.0
            continue
        ref = g["grad." + sp.name]
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=1e-6 + 2e-4 * np.abs(ref).max(), err_msg=sp.name)


@pytest.mark.parametrize("T", [20, 30])
def test_f32_full_config_vs_golden(golden_dir, T):
    g = np.load(os.path.join(golden_dir, "g2_full_T%d.npz" % T))
    e = make_engine(FULL, "f32")
    b = dev(full_batch(T))
    e.ensure_shape(4, T, 36)
    e.sync_weights()
    lg, pl = run(e, b)
    err = np.abs(lg.cpu().numpy() - g["logits"]).max()
    assert err <= 1e-3, err            # the north-star bound
    assert err <= 2e-4, err            # what exact-f32 operands actually deliver
    np.testing.assert_allclose(pl.cpu().numpy(), g["pooled"], rtol=0, atol=2e-4)
    loss = e.loss_backward(b["target"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    np.testing.assert_allclose(e.grad_norm().item(), g["grad_norm"], rtol=1e-3)
    off = 0
    byname = {sp.name: sp for sp in e.specs}
    for k, n in zip(g["grad_names"].tolist(), g["grad_counts"].tolist()):
        gr = e.view(e.grads, byname[k]).cpu().numpy()
        ref = g["grad_samples"][off:off + n]
        off += n
        np.testing.assert_allclose(gr.reshape(-1)[sample_idx(k, gr.size)], ref, rtol=5e-3, atol=1e-6 + 5e-4 * np.abs(gr).max(), err_msg=k)


@pytest.mark.parametrize("T", [20, 30])
def test_x3_full_config_vs_golden(golden_dir, T):
    """bf16x3 precision on the real 9/5/5 architecture against the reference's own outputs (G2): the north-star bound on the logits,
    and gradients at f32-class accuracy (the bf16 mode's figures on the same fixture: logits 4e-2, sampled gradients 9e-3)."""
    g = np.load(os.path.join(golden_dir, "g2_full_T%d.npz" % T))
    e = make_engine(FULL, "bf16x3")
    b = dev(full_batch(T))
    e.ensure_shape(4, T, 36)
    e.sync_weights()
    lg, pl = run(e, b)
    err = np.abs(lg.cpu().numpy() - g["logits"])
    perr = np.abs(pl.cpu().numpy() - g["pooled"])
    loss = e.loss_backward(b["target"]).item()
    gn = e.grad_norm().item()
    worst, wname, overall = _grad_sample_errors(e, g["grad_names"].tolist(), g["grad_counts"].tolist(), g["grad_samples"])
    _report("bf16x3 full B=4 T=%d vs G2" % T, logits_max=err.max(), logits_mean=err.mean(), pooled_max=perr.max(), loss_rel=abs(loss - g["loss"]) / abs(g["loss"]),
            grad_norm_rel=abs(gn - g["grad_norm"]) / g["grad_norm"], grad_samples_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    assert err.max() <= 1e-3, err.max()            # the north-star bound
    assert perr.max() <= 1e-3, perr.max()
    assert abs(loss - g["loss"]) < 1e-4 * abs(g["loss"])
    assert abs(gn - g["grad_norm"]) < 1e-3 * g["grad_norm"]
    assert overall < 2e-3, overall
    assert worst < 1e-2, (wname, worst)


MED = dict(vocab_size=512, hidden=128, heads=2, inter=256, max_pos=64, type_vocab=2, l_layers=3, x_layers=2, r_layers=2,
           feat_dim=64, pos_dim=4, num_answers=70)


def oracle_run(cfgd, b, want_grads=True):
    from oracle import lxmert_ref as R
    cfg = R.RefConfig(**cfgd)
    P = {k: torch.from_numpy(v).requires_grad_(want_grads) for k, v in synth.fill_state_dict(R.param_shapes(cfg)).items()}
    t = {k: torch.from_numpy(v) for k, v in b.items() if k != "lengths"}
    lg, pl = R.gqa_forward(P, cfg, t["feats"], t["boxes"], t["input_ids"], t["input_mask"], t["segment_ids"])
    loss = R.bce_loss(lg, t["target"])
    if want_grads:
        loss.backward()
    return lg.detach(), pl.detach(), loss.item(), P


@pytest.mark.parametrize("precision,tol,gtol", [("f32", 1e-4, 2e-3), ("bf16x3", 2e-4, 2e-3), ("bf16", 1.5e-2, 3.5e-2), ("bf16x3_fwd", 2e-4, 3.5e-2)])    # bf16 observed: logits 7.6e-3, worst tensor 1.7e-2
def test_medium_config_vs_oracle(precision, tol, gtol):
    """head size 64 / dims multiple of 64, so the bf16 MFMA kernels are the ones exercised.
    bf16 tolerance (2x observed): bf16 has 8 significant bits; through 7 blocks logits (|z|~1) land within 1.5e-2 abs, gradients within
    3.5% relative Frobenius error per tensor."""
    B, T, O = 6, 12, 10
    b = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=5, min_len=2)
    lg_r, pl_r, loss_r, Pr = oracle_run(MED, b)
    e = make_engine(MED, precision)
    d = dev(b)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    lg, pl = run(e, d)
    lerr = float((lg.cpu() - lg_r).abs().max())
    assert lerr < tol
    loss = e.loss_backward(d["target"])
    assert abs(loss.item() - loss_r) < tol * 50
    gworst = 0.0
    for sp in e.specs:
        got = e.view(e.grads, sp).cpu()
        ref = Pr[sp.name].grad
        if ref is None or sp.dead:
            assert float(got.abs().max()) == 0.0, sp.name
            continue
        den = float(ref.norm())
        if den < 1e-8:
            continue
        gworst = max(gworst, float((got - ref).norm()) / den)
        assert float((got - ref).norm()) / den < gtol, (sp.name, float((got - ref).norm()) / den)
    _report("medium config %s vs oracle" % precision, logits_max=lerr, worst_tensor_rel=gworst)


@pytest.mark.parametrize("packed", [False, True])
def test_embedding_gradients_with_many_repeated_words(packed):
    """The deterministic embedding backward (csrc/embed.hip, round 6) on a batch built to take every one of its paths: 90 distinct words over 64 x 16 tokens, so that
    more than 64 words are named by more than 8 rows (the hot list overflows: the four-wave path of embed_word_grad_kernel), others by 2..8 rows (the one-wave
    path), a few once (the copy path), [CLS] / [SEP] by every sample (the keyed partial sums), and token-type 1 rows.  f32 engine against the oracle, every
    embedding table; two passes must agree bit for bit."""
    B, T, O = 160, 16, 10
    b = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=77, min_len=6)
    rng = np.random.RandomState(3)
    ids = b["input_ids"]
    for r in range(B):
        L = int(b["lengths"][r])
        ids[r, 1:L - 1] = 110 + rng.randint(0, 88, size=L - 2)        # 88 frequent words ...
    ids[0, 1], ids[1, 1], ids[2, 1] = 300, 301, 302                       # ... three that occur once
    ids[3, 2] = ids[4, 2] = ids[5, 2] = 303                               # ... one that occurs three times
    b["segment_ids"][:, 3] = 1
    b["segment_ids"] = b["segment_ids"] * (ids != 0)
    lg_r, pl_r, loss_r, Pr = oracle_run(MED, b)
    e = make_engine(MED, "f32")
    d = dev(b)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    lens = np.ascontiguousarray(b["lengths"], dtype=np.int32) if packed else None
    outs = []
    for _ in range(2):
        e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], lengths=lens)
        e.loss_backward(d["target"])
        outs.append(e.grads.clone())
    assert torch.equal(outs[0], outs[1])
    counts = np.bincount(ids[ids != 0].reshape(-1))
    assert (counts > 8).sum() > 64 and (counts == 1).sum() >= 3 and counts[101] == B
    for sp in e.specs:
        if "embeddings" in sp.name and "LayerNorm" not in sp.name:
            got, ref = e.view(e.grads, sp).cpu(), Pr[sp.name].grad
            rel = float((got - ref).norm()) / max(1e-12, float(ref.norm()))
            assert rel < 2e-3, (sp.name, rel)
            assert float(got[0].abs().max()) == 0.0, sp.name             # padding_idx = 0 on all three tables


def _report(tag, **kv):
    """observed errors go to stdout (pytest -s) and to gpurun_out/parity_observed.txt: every bf16 tolerance below is <= 2x what is printed"""
    line = tag + ": " + ", ".join("%s %.3e" % (k, v) for k, v in kv.items())
    print(line)
    try:
        os.makedirs(os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out", "parity_observed.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _grad_sample_errors(e, names, counts, samples):
    """per-tensor relative error of the gradient entries the golden fixture sampled: (worst tensor, its name, error over all samples)"""
    byname = {sp.name: sp for sp in e.specs}
    off, worst, wname, num, den = 0, 0.0, "", 0.0, 0.0
    for k, n in zip(names, counts):
        gr = e.view(e.grads, byname[k]).float().cpu().numpy().reshape(-1)
        ref = samples[off:off + n]
        off += n
        got = gr[sample_idx(k, gr.size)]
        d, r = float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref))
        num += d * d
        den += r * r
        if r > 1e-6 * max(1.0, float(np.abs(gr).max())) and d / r > worst:
            worst, wname = d / r, k
    return worst, wname, (num / max(den, 1e-30)) ** 0.5


@pytest.mark.parametrize("T", [20, 30])
def test_bf16_full_config_vs_golden(golden_dir, T):
    """The benchmarked precision on the real 9/5/5 architecture (B=4) against the reference's own outputs (G2): logits, pooled,
    loss, gradient norm and the sampled gradients of all 439 live tensors.  bf16 operands (8 significant bits) through 19 blocks."""
    g = np.load(os.path.join(golden_dir, "g2_full_T%d.npz" % T))
    e = make_engine(FULL, "bf16")
    b = dev(full_batch(T))
    e.ensure_shape(4, T, 36)
    e.sync_weights()
    lg, pl = run(e, b)
    err = np.abs(lg.cpu().numpy() - g["logits"])
    perr = np.abs(pl.cpu().numpy() - g["pooled"])
    loss = e.loss_backward(b["target"]).item()
    gn = e.grad_norm().item()
    worst, wname, overall = _grad_sample_errors(e, g["grad_names"].tolist(), g["grad_counts"].tolist(), g["grad_samples"])
    _report("bf16 full B=4 T=%d vs G2" % T, logits_max=err.max(), logits_mean=err.mean(), pooled_max=perr.max(), loss_rel=abs(loss - g["loss"]) / abs(g["loss"]),
            grad_norm_rel=abs(gn - g["grad_norm"]) / g["grad_norm"], grad_samples_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    # tolerances: 2x the largest error observed on MI355X over the round's kernel variants (T=20 / T=30) for the element-wise figures -
    # logits max 3.8e-2 / 5.1e-2, mean 8.2e-3, pooled max 2.4e-2 / 2.7e-2, all sampled gradient entries 9.1e-3 / 9.0e-3 rel, worst
    # tensor 5.8e-2 / 4.1e-2.  The two scalars are sums whose rounding errors largely cancel, so they wander between kernel variants
    # (loss 1.3e-5 .. 1.4e-4 rel, gradient norm 8.7e-5 .. 5.0e-4 rel): bounded at 5e-4 / 1.5e-3.
    # primary gates (what bf16 operands CAN promise; this mode is outside the north star's 1e-3 logits bound - see the module docstring)
    assert err.mean() < 1.6e-2, err.mean()
    assert abs(loss - g["loss"]) < 5e-4 * abs(g["loss"])
    assert abs(gn - g["grad_norm"]) < 1.5e-3 * g["grad_norm"]
    assert overall < 1.8e-2, overall
    # secondary: element-wise maxima at <= 2x the observed values
    assert err.max() < 1.0e-1, err.max()
    assert perr.max() < 5.4e-2, perr.max()
    assert worst < 0.115, (wname, worst)


_B256 = {}


def _b256_oracle():
    """config 3 at size: B=256, T=20, the oracle's forward + backward on the host (once per session, ~30 s)"""
    if not _B256:
        import time
        b = synth.synth_batch(256, 20, seed=777)
        t0 = time.time()
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
        lg, pl, loss, P = oracle_run(FULL, b)
        gsq = sum(float((p.grad.double() ** 2).sum()) for p in P.values() if p.grad is not None)
        _B256.update(b=b, logits=lg.numpy(), pooled=pl.numpy(), loss=loss, grad_norm=gsq ** 0.5,
                     grads={k: p.grad.numpy().reshape(-1)[sample_idx(k, p.grad.numel())].copy() for k, p in P.items() if p.grad is not None})
        print("oracle B=256 fwd+bwd on the host: %.1f s" % (time.time() - t0))
    return _B256


@pytest.mark.parametrize("layout", ["packed", "padded"])
def test_bf16_cfg3_b256_fwd_bwd_vs_oracle(layout):
    """BASELINE config 3 at its real size (B=256, T=20, bf16, dropout off for parity): one forward + backward against the CPU
    oracle - the split-K visual-projection wgrad, the 256-row wgrad tiles with contraction tails and the persistent / phase-
    interleaved NT tiles only exist at this size.  Both language layouts (packed rows = what bench.py runs; padded = the reference's)."""
    o = _b256_oracle()
    b = o["b"]
    e = make_engine(FULL, "bf16")
    d = dev(b)
    e.ensure_shape(256, 20, 36)
    e.sync_weights()
    lengths = np.ascontiguousarray(b["lengths"], dtype=np.int32) if layout == "packed" else None
    lg, pl = e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], train=False, seed=0, lengths=lengths)
    err = np.abs(lg.cpu().numpy() - o["logits"])
    loss = e.loss_backward(d["target"]).item()
    gn = e.grad_norm().item()
    byname = {sp.name: sp for sp in e.specs}
    num = den = 0.0
    worst, wname = 0.0, ""
    for k, ref in o["grads"].items():
        gr = e.view(e.grads, byname[k]).float().cpu().numpy().reshape(-1)
        got = gr[sample_idx(k, gr.size)]
        dd, rr = float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref))
        num += dd * dd
        den += rr * rr
        if rr > 1e-6 * max(1.0, float(np.abs(gr).max())) and dd / rr > worst:
            worst, wname = dd / rr, k
    overall = (num / den) ** 0.5
    _report("bf16 full B=256 T=20 %s vs oracle" % layout, logits_max=err.max(), logits_mean=err.mean(), loss_rel=abs(loss - o["loss"]) / abs(o["loss"]),
            grad_norm_rel=abs(gn - o["grad_norm"]) / o["grad_norm"], grad_samples_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    assert len(o["grads"]) == 439
    # tolerances: 2x the largest observed (both layouts agree to the digits shown): logits max 4.5e-2 .. 5.1e-2, mean 7.75e-3, sampled
    # gradient entries 6.3e-3 rel, worst tensor (word embeddings) 2.4e-2; the scalars (cancelling sums) observed at loss 1.8e-6 .. 7.7e-6 rel,
    # gradient norm 6.2e-5 .. 7.7e-5 rel: bounded at 1e-4 / 4e-4
    # primary gates: mean logit error, loss, gradient norm, the sampled gradient entries; secondary: the element-wise maxima.  (bf16 is
    # BASELINE config 3's mode and is OUTSIDE the north star's 1e-3 logits bound; the bound is met by bf16x3: the next test)
    assert err.mean() < 1.55e-2, err.mean()
    assert abs(loss - o["loss"]) < 1e-4 * abs(o["loss"])
    assert abs(gn - o["grad_norm"]) < 4e-4 * o["grad_norm"]
    assert overall < 1.25e-2, overall
    assert err.max() < 1.0e-1, err.max()
    assert worst < 4.8e-2, (wname, worst)


def test_x3_cfg3_b256_fwd_bwd_vs_oracle():
    """The tolerance-compliant train step at BASELINE's size (B=256, T=20, packed rows, dropout off for parity): bf16x3 forward +
    backward against the CPU oracle: logits inside the north star's 1e-3, every sampled gradient entry at f32-class accuracy."""
    o = _b256_oracle()
    b = o["b"]
    e = make_engine(FULL, "bf16x3")
    d = dev(b)
    e.ensure_shape(256, 20, 36)
    e.sync_weights()
    lengths = np.ascontiguousarray(b["lengths"], dtype=np.int32)
    lg, pl = e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], train=False, seed=0, lengths=lengths)
    err = np.abs(lg.cpu().numpy() - o["logits"])
    loss = e.loss_backward(d["target"]).item()
    gn = e.grad_norm().item()
    byname = {sp.name: sp for sp in e.specs}
    num = den = 0.0
    worst, wname = 0.0, ""
    for k, ref in o["grads"].items():
        gr = e.view(e.grads, byname[k]).float().cpu().numpy().reshape(-1)
        got = gr[sample_idx(k, gr.size)]
        dd, rr = float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref))
        num += dd * dd
        den += rr * rr
        if rr > 1e-6 * max(1.0, float(np.abs(gr).max())) and dd / rr > worst:
            worst, wname = dd / rr, k
    overall = (num / den) ** 0.5
    _report("bf16x3 full B=256 T=20 packed vs oracle", logits_max=err.max(), logits_mean=err.mean(), loss_rel=abs(loss - o["loss"]) / abs(o["loss"]),
            grad_norm_rel=abs(gn - o["grad_norm"]) / o["grad_norm"], grad_samples_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    assert err.max() <= 1e-3, err.max()            # the north-star bound, at the benchmarked size
    assert abs(loss - o["loss"]) < 2e-5 * abs(o["loss"])
    assert abs(gn - o["grad_norm"]) < 2e-4 * o["grad_norm"]
    assert overall < 2e-3, overall
    assert worst < 1e-2, (wname, worst)


# ---------------------------------------------------------------------------------------------------------------------
# bf16x3_fwd precision (round 4): the bf16x3 forward pass (same kernels: logits inside the north star's 1e-3) with the bf16 backward
# pass (BASELINE config 3 prescribes a bf16 backward).  Logits are held to the bf16x3 tests' bounds, gradients to the bf16 tests' gates.
def _sampled_grad_errors(e, ref_grads):
    byname = {sp.name: sp for sp in e.specs}
    num = den = 0.0
    worst, wname = 0.0, ""
    for k, ref in ref_grads.items():
        gr = e.view(e.grads, byname[k]).float().cpu().numpy().reshape(-1)
        got = gr[sample_idx(k, gr.size)]
        dd, rr = float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref))
        num += dd * dd
        den += rr * rr
        if rr > 1e-6 * max(1.0, float(np.abs(gr).max())) and dd / rr > worst:
            worst, wname = dd / rr, k
    return worst, wname, (num / den) ** 0.5


@pytest.mark.parametrize("train", [False, True])
def test_mixed_forward_is_the_bf16x3_forward(train):
    """Same kernels, same operands: logits and pooled output of the bf16x3_fwd engine are bit-identical to the bf16x3 engine's (eval
    and train mode, packed rows) - the extra bf16 images change nothing the forward computes."""
    B, T = 4, 20
    raw = full_batch(T)
    b = dev(raw)
    lens = [int(v) for v in raw["input_mask"].sum(1)]
    outs = {}
    for prec in ("bf16x3", "bf16x3_fwd"):
        e = make_engine(FULL, prec, dropout=0.1 if train else 0.0)
        e.ensure_shape(B, T, 36)
        e.sync_weights()
        lg, pl = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=train, seed=11, lengths=lens)
        outs[prec] = (lg.clone(), pl.clone())
        del e
    assert torch.equal(outs["bf16x3"][0], outs["bf16x3_fwd"][0])
    assert torch.equal(outs["bf16x3"][1], outs["bf16x3_fwd"][1])


@pytest.mark.parametrize("T", [20, 30])
def test_mixed_full_config_vs_golden(golden_dir, T):
    """bf16x3_fwd on the real 9/5/5 architecture against the reference's own outputs (G2, B=4): logits / pooled / loss at the bf16x3
    bounds (the north star's 1e-3), gradient norm and sampled gradients of all 439 live tensors at the bf16 mode's gates."""
    g = np.load(os.path.join(golden_dir, "g2_full_T%d.npz" % T))
    e = make_engine(FULL, "bf16x3_fwd")
    b = dev(full_batch(T))
    e.ensure_shape(4, T, 36)
    e.sync_weights()
    lg, pl = run(e, b)
    err = np.abs(lg.cpu().numpy() - g["logits"])
    perr = np.abs(pl.cpu().numpy() - g["pooled"])
    loss = e.loss_backward(b["target"]).item()
    gn = e.grad_norm().item()
    worst, wname, overall = _grad_sample_errors(e, g["grad_names"].tolist(), g["grad_counts"].tolist(), g["grad_samples"])
    _report("bf16x3_fwd full B=4 T=%d vs G2" % T, logits_max=err.max(), logits_mean=err.mean(), pooled_max=perr.max(), loss_rel=abs(loss - g["loss"]) / abs(g["loss"]),
            grad_norm_rel=abs(gn - g["grad_norm"]) / g["grad_norm"], grad_samples_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    assert err.max() <= 1e-3, err.max()            # the north-star bound
    assert perr.max() <= 1e-3, perr.max()
    assert abs(loss - g["loss"]) < 1e-4 * abs(g["loss"])          # the loss is a function of the forward pass alone
    # gradients: the bf16 mode's gates (test_bf16_full_config_vs_golden)
    assert abs(gn - g["grad_norm"]) < 1.5e-3 * g["grad_norm"]
    assert overall < 1.8e-2, overall
    assert worst < 0.115, (wname, worst)


def test_mixed_cfg3_b256_fwd_bwd_vs_oracle():
    """bf16x3_fwd at BASELINE's size (B=256, T=20, packed rows, dropout off for parity) against the CPU oracle: logits inside 1e-3,
    gradients at the gates of test_bf16_cfg3_b256_fwd_bwd_vs_oracle."""
    o = _b256_oracle()
    b = o["b"]
    e = make_engine(FULL, "bf16x3_fwd")
    d = dev(b)
    e.ensure_shape(256, 20, 36)
    e.sync_weights()
    lengths = np.ascontiguousarray(b["lengths"], dtype=np.int32)
    lg, pl = e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], train=False, seed=0, lengths=lengths)
    err = np.abs(lg.cpu().numpy() - o["logits"])
    loss = e.loss_backward(d["target"]).item()
    gn = e.grad_norm().item()
    worst, wname, overall = _sampled_grad_errors(e, o["grads"])
    _report("bf16x3_fwd full B=256 T=20 packed vs oracle", logits_max=err.max(), logits_mean=err.mean(), loss_rel=abs(loss - o["loss"]) / abs(o["loss"]),
            grad_norm_rel=abs(gn - o["grad_norm"]) / o["grad_norm"], grad_samples_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    assert err.max() <= 1e-3, err.max()            # the north-star bound, at the benchmarked size
    assert abs(loss - o["loss"]) < 2e-5 * abs(o["loss"])
    assert abs(gn - o["grad_norm"]) < 4e-4 * o["grad_norm"]
    assert overall < 1.25e-2, overall
    assert worst < 4.8e-2, (wname, worst)


def test_mixed_train_steps_follow_the_bf16x3_run():
    """Three optimizer steps (dropout off) of the bf16x3_fwd engine next to the bf16x3 engine from the same weights.  BertAdam's update is
    m / (sqrt(v) + eps) without bias correction: sign-like, up to 0.1 / sqrt(0.001) = 3.2 lr per element in the first steps, so an element whose
    tiny gradient changes sign under bf16 rounding moves the other way - the two runs can differ by ~6 lr per element and step; the logits of
    the fourth forward stay close."""
    B, T, O = 6, 12, 10
    raw = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=5, min_len=2)
    b = dev(raw)
    res = {}
    lr = 1e-4
    for prec in ("bf16x3", "bf16x3_fwd"):
        e = make_engine(MED, prec)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        for i in range(3):
            run(e, b)
            e.loss_backward(b["target"])
            e.adam_step(lr, max_norm=5.0)
        lg, _ = run(e, b)
        res[prec] = (lg.clone(), e.params.clone())
    dl = float((res["bf16x3"][0] - res["bf16x3_fwd"][0]).abs().max())
    dp = float((res["bf16x3"][1] - res["bf16x3_fwd"][1]).abs().max())
    _report("bf16x3_fwd vs bf16x3 after 3 steps (medium config)", logits_diff=dl, weights_diff=dp)
    assert dp <= 3 * 2 * 3.2 * lr * 1.05, dp
    assert dl < 1e-2, dl


@pytest.mark.parametrize("precision,tol,gtol", [("f32", 1e-4, 2e-3), ("bf16x3", 2e-4, 2e-3), ("bf16x3_fwd", 2e-4, 3.5e-2)])
def test_roi_mixup_train_step_vs_oracle(precision, tol, gtol):
    """BASELINE config 4's step (tasks/gqa_mixup_vis.py:134-181, 250-259) end to end: RoIMixup('mixup_v1', Beta(1, 5)) doubles a B = 8
    loader batch on the device (the six constructions themselves are pinned bit-exactly to the reference's by test_roi_mixup_entry_vs_golden),
    `sent = sent + sent`, then ONE engine step on the 16 rows: logits, loss (BCE x NA over 2B rows) and every gradient tensor against the
    oracle on the same doubled batch; packed language rows as bench.py --mixup runs them."""
    import random
    from rgqa_amd.mixup import RoIMixup
    B, T, O = 8, 12, 10
    raw = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=91, min_len=2, uq_frac=0.25)
    d = dev(raw)
    random.seed(5); np.random.seed(5)
    mixer = RoIMixup("mixup_v1", alpha=1.0, beta=5.0)
    f2, b2, t2 = mixer(d["feats"], d["boxes"], d["target"], list(range(B)))
    assert f2.shape[0] == 2 * B and float((t2[B:] - d["target"]).abs().max()) > 0          # targets of the mixed rows are scaled by prop
    ids2, mask2, seg2 = (torch.cat([d[k], d[k]], 0).contiguous() for k in ("input_ids", "input_mask", "segment_ids"))
    doubled = dict(feats=f2.cpu().numpy(), boxes=b2.cpu().numpy(), target=t2.cpu().numpy(), input_ids=ids2.cpu().numpy(),
                   input_mask=mask2.cpu().numpy(), segment_ids=seg2.cpu().numpy())
    lg_r, pl_r, loss_r, Pr = oracle_run(MED, doubled)
    e = make_engine(MED, precision)
    e.ensure_shape(2 * B, T, O)
    e.sync_weights()
    lens = np.ascontiguousarray(np.tile(raw["lengths"], 2), dtype=np.int32)
    lg, pl = e.forward(f2, b2, ids2, mask2, seg2, train=False, lengths=lens)
    lerr = float((lg.cpu() - lg_r).abs().max())
    loss = e.loss_backward(t2).item()
    gworst = 0.0
    for sp in e.specs:
        got = e.view(e.grads, sp).cpu()
        ref = Pr[sp.name].grad
        if ref is None or sp.dead:
            assert float(got.abs().max()) == 0.0, sp.name
            continue
        den = float(ref.norm())
        if den < 1e-8:
            continue
        gworst = max(gworst, float((got - ref).norm()) / den)
    _report("RoI-mixup step (medium config, 16 rows) %s vs oracle" % precision, logits_max=lerr, loss_rel=abs(loss - loss_r) / abs(loss_r), worst_tensor_rel=gworst)
    assert lerr < tol and abs(loss - loss_r) < 50 * tol and gworst < gtol


@pytest.mark.parametrize("precision,ltol,gtol", [("bf16x3", 1e-3, 3e-3), ("bf16", 1.0e-1, 6e-2)])
def test_roi_mixup_step_at_config4_size_vs_oracle(precision, ltol, gtol):
    """BASELINE config 4 at its per-GPU size on the FULL architecture: a loader batch of 256 QA pairs doubled by
    RoIMixup('mixup_v1', Beta(1, 5)) to 512 model rows (tasks/gqa_mixup_vis.py:134-181), packed language rows as bench.py runs them.
    Forward: the logits of ALL 512 rows and the loss (BCE x NA over 512 rows) against the CPU oracle's forward on the same doubled batch
    (bf16x3: inside the north star's 1e-3; bf16: the headline mode's gates).  Backward: samples are independent, so a backward pass whose
    incoming gradient is non-zero on 6 spread rows only - exactly those rows' share (sigmoid(z) - t) / 512 of the step's dlogits - must give
    the oracle's gradients of the 6-row sub-batch scaled by 6 / 512: every kernel runs at the 512-row size, every tensor is compared."""
    import random
    from oracle import lxmert_ref as R
    from rgqa_amd.mixup import RoIMixup
    B, T, O = 256, 20, 36
    raw = synth.synth_batch(B, T, seed=4242, uq_frac=0.25)
    d = dev(raw)
    random.seed(11); np.random.seed(11)
    f2, b2, t2 = RoIMixup("mixup_v1", alpha=1.0, beta=5.0)(d["feats"], d["boxes"], d["target"], list(range(B)))
    assert f2.shape == (2 * B, O, 2048) and t2.shape == (2 * B, 1842)
    ids2, mask2, seg2 = (torch.cat([d[k], d[k]], 0).contiguous() for k in ("input_ids", "input_mask", "segment_ids"))
    lens = np.ascontiguousarray(np.tile(raw["lengths"], 2), dtype=np.int32)
    e = make_engine(FULL, precision)
    e.ensure_shape(2 * B, T, O)
    e.sync_weights()
    lg, _ = e.forward(f2, b2, ids2, mask2, seg2, train=False, lengths=lens)
    lg = lg.clone()
    loss = e.loss_backward(t2).item()
    gn_full = e.grad_norm().item()
    assert np.isfinite(loss) and np.isfinite(gn_full) and gn_full > 0
    # ---- the oracle's forward on all 512 rows (no graph: ~10 s on the GPU box's host cores)
    cfg = R.RefConfig(**FULL)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    Pn = synth.fill_state_dict(R.param_shapes(cfg))
    cpu = dict(feats=f2.cpu(), boxes=b2.cpu(), ids=ids2.cpu(), mask=mask2.cpu(), seg=seg2.cpu(), target=t2.cpu())
    with torch.no_grad():
        Pd = {k: torch.from_numpy(v) for k, v in Pn.items()}
        lg_r, _ = R.gqa_forward(Pd, cfg, cpu["feats"], cpu["boxes"], cpu["ids"], cpu["mask"], cpu["seg"])
        loss_r = R.bce_loss(lg_r, cpu["target"]).item()
    err = (lg.cpu() - lg_r).abs()
    # ---- backward restricted to 6 rows (3 loader rows, 3 mixed rows)
    pick = [0, 101, 255, 256, 300, 511]
    dl = torch.zeros(2 * B, 1842, device="cuda")
    dl[pick] = (torch.sigmoid(lg[pick]) - t2[pick]) / (2 * B)
    e.forward(f2, b2, ids2, mask2, seg2, train=False, lengths=lens)
    e.backward(dl)
    P6 = {k: torch.from_numpy(v).requires_grad_(True) for k, v in Pn.items()}
    lg6, _ = R.gqa_forward(P6, cfg, cpu["feats"][pick], cpu["boxes"][pick], cpu["ids"][pick], cpu["mask"][pick], cpu["seg"][pick])
    R.bce_loss(lg6, cpu["target"][pick]).backward()
    scale = len(pick) / (2.0 * B)
    num = den = 0.0
    worst, wname = 0.0, ""
    for sp in e.specs:
        ref = P6[sp.name].grad
        got = e.view(e.grads, sp).float().cpu()
        if ref is None or sp.dead:
            assert float(got.abs().max()) == 0.0, sp.name
            continue
        ref = ref * scale
        dd, rr = float((got - ref).norm()), float(ref.norm())
        num += dd * dd
        den += rr * rr
        if rr > 1e-7 and dd / rr > worst:
            worst, wname = dd / rr, sp.name
    overall = (num / den) ** 0.5
    _report("RoI-mixup step at config 4's size (512 rows, full config) %s vs oracle" % precision, logits_max=float(err.max()), logits_mean=float(err.mean()),
            loss_rel=abs(loss - loss_r) / abs(loss_r), grads_rel=overall, worst_tensor_rel=worst)
    print("   worst tensor:", wname)
    assert float(err.max()) < ltol, float(err.max())
    assert abs(loss - loss_r) < (1e-4 if precision == "bf16" else 2e-5) * abs(loss_r)
    assert overall < gtol / 3 and worst < gtol, (overall, wname, worst)


def test_dropout_train_mode_is_deterministic_and_consistent():
    """Train mode (dropout 0.1 regenerated from (seed, site, index) in backward): same seed -> bit-identical results,
    different seed -> different; the analytic gradient agrees with a finite difference of the loss along a direction."""
    B, T, O = 4, 8, 6
    b = dev(synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=9, min_len=2))
    e = make_engine(MED, "f32", dropout=0.1)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    l1 = run(e, b, True, 123)[0].clone()
    loss1 = e.loss_backward(b["target"]).item()
    g1 = e.grads.clone()
    l2 = run(e, b, True, 123)[0].clone()
    e.loss_backward(b["target"])
    assert torch.equal(l1, l2)
    # every gradient is bit-reproducible, the three embedding tables included (round 6: summed per table row in a fixed order, no float atomics)
    assert torch.equal(g1, e.grads)
    assert loss1 == e._io["loss"].item()
    l3 = run(e, b, True, 124)[0].clone()
    assert not torch.equal(l1, l3)
    # directional finite difference (same seed => same masks => differentiable function of the weights)
    torch.manual_seed(0)
    dirn = torch.zeros_like(e.params)
    for sp in e.specs:
        if not sp.dead and "embeddings" not in sp.name:
            e.view(dirn, sp).copy_(torch.randn(sp.shape, device="cuda") * 0.02)
    ana = float((g1.double() * dirn.double()).sum())
    eps = 1e-2
    base = e.params.clone()
    vals = []
    for sgn in (1, -1):
        e.params.copy_(base + sgn * eps * dirn)
        run(e, b, True, 123)
        vals.append(e.loss_backward(b["target"]).item())
    e.params.copy_(base)
    num = (vals[0] - vals[1]) / (2 * eps)
    assert abs(num - ana) / max(1e-6, abs(ana)) < 2e-2, (num, ana)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "bf16x3_fwd"])
@pytest.mark.parametrize("packed", [False, True])
def test_paired_attention_launch_is_bit_identical(precision, packed, monkeypatch):
    """Round 3: the two attention problems of a stage (language | vision self-attention, the two cross-attention directions) go out as ONE
    launch when their tile shapes are the GQA ones (T in 17..32, 33..48 regions).  Same per-block code on the same operands: logits and
    every gradient are bit-identical to the two-launch path (rgqa_debug_set key 16 = 0),
    train mode, padded and packed rows."""
    B, T, O = 5, 20, 36
    cfgd = dict(MED, l_layers=2, x_layers=2, r_layers=1)
    raw = synth.synth_batch(B, T, O=O, F=cfgd["feat_dim"], NA=cfgd["num_answers"], vocab=cfgd["vocab_size"], seed=77, min_len=3)
    b = dev(raw)
    lens = [int(v) for v in raw["input_mask"].sum(1)] if packed else None
    outs = {}
    from rgqa_amd import _lib
    L = _lib.load()
    for pair in ("0", "1"):
        assert L.rgqa_debug_set(16, int(pair)) == 0
        e = make_engine(cfgd, precision, dropout=0.1)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        lg, _ = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=99, lengths=lens)
        lg = lg.clone()
        e.loss_backward(b["target"])
        outs[pair] = (lg, e.grads.clone())
    L.rgqa_debug_set(16, 1)
    assert torch.equal(outs["0"][0], outs["1"][0])
    assert torch.equal(outs["0"][1], outs["1"][1])
    assert float(outs["1"][1].abs().max()) > 0


def test_side_stream_wgrad_matches_serial():
    """The deferred per-layer weight-gradient launches run on a side stream against double-buffered gradient sets;
    forcing them onto the main stream (rgqa_debug_set key 2) must give bit-identical gradients, repeatedly."""
    from rgqa_amd import _lib
    L = _lib.load()
    B, T, O = 64, 20, 36
    b = dev(synth.synth_batch(B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=5, min_len=3))
    e = make_engine(FULL, "bf16", dropout=0.1)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    first = 0      # (until round 6 the float-atomic embedding tables were left out)

    def grads(serial):
        assert L.rgqa_debug_set(2, serial) == 0
        try:
            run(e, b, True, 77)
            e.loss_backward(b["target"])
            torch.cuda.synchronize()
            return e.grads[first:].clone()
        finally:
            L.rgqa_debug_set(2, 0)

    ref = grads(1)
    assert float(ref.abs().max()) > 0
    for _ in range(3):
        assert torch.equal(grads(0), ref)


def test_dgrad_on_the_weights_as_they_lie_gives_the_same_gradients():
    """rgqa_debug_set key 14 = 1: the bf16 engine's dgrad GEMMs take the [K, N] operand form on the forward copy of the weights and only two
    transposed copies are re-made at a weight sync (9 MB instead of 410).  Gradients after a whole backward pass - the [CLS]-row split-K
    launches, the last cross layer's partial q / kv projections, the input gradient of the visual projection (which keeps its transposed
    copy) included - must equal the default regime's bit for bit, over two passes with a re-make of the copies in between."""
    from rgqa_amd import _lib
    L = _lib.load()
    B, T, O = 48, 20, 36
    raw = synth.synth_batch(B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=9, min_len=3)
    b = dev(raw)
    lens = np.ascontiguousarray(raw["lengths"], dtype=np.int32)
    res = {}
    try:
        for nn in (0, 1):
            assert L.rgqa_debug_set(14, nn) == 0
            e = make_engine(FULL, "bf16", dropout=0.1)
            e.ensure_shape(B, T, O)
            e.sync_weights()
            dfe = torch.zeros(B * O, FULL["feat_dim"], device="cuda")
            e.set_input_grads(dfe, None)
            out = []
            for step in range(2):
                e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=31 + step, lengths=lens)
                e.loss_backward(b["target"])
                out.append((e.grads.clone(), dfe.clone()))
                e.adam_step(0.0, clip=False)      # learning rate 0: the weights stay (an update would carry the tables' rounding into step 2), the copies are re-made
            torch.cuda.synchronize()
            res[nn] = (out, e.params.clone())
            e.set_input_grads(None, None)
    finally:
        L.rgqa_debug_set(14, 0)
    for (g0, d0), (g1, d1) in zip(res[0][0], res[1][0]):
        assert float(g0.abs().max()) > 0 and float(d0.abs().max()) > 0
        assert torch.equal(g0, g1) and torch.equal(d0, d1)      # the embedding tables too (deterministic since round 6)
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("prec", ["bf16", "f32", "bf16x3_fwd"])
def test_optimizer_pass_beside_the_next_forward_is_the_serial_step(prec):
    """Engine.adam_step(overlap=True) (round 5, the default): BertAdam runs on a stream of its own, gradient segment by gradient segment in forward
    order, while the caller's stream goes straight on to the next forward pass, whose layers wait for the event of their segment; the
    transposed copies follow behind the last segment and the next backward waits for them.  Same kernels on the same ranges: parameters, Adam
    moments and both operand copies equal the serial step's bit for bit after four steps, and an arena read through the engine's attributes right after adam_step -
    no synchronisation - already sees the finished update (the properties join the update stream)."""
    cfg = FULL if prec != "f32" else MED
    B, T, O = (24, 20, 36) if prec != "f32" else (6, 12, 10)
    raw = synth.synth_batch(B, T, O=O, F=cfg["feat_dim"], NA=cfg["num_answers"], vocab=cfg["vocab_size"], seed=12, min_len=3)
    b = dev(raw)
    lens = np.ascontiguousarray(raw["lengths"], dtype=np.int32)
    res = {}
    for ov in (False, True):
        e = make_engine(cfg, prec, dropout=0.1)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        e.enable_segment_sumsq(True)
        assert e.num_weight_segments() > 0
        early = None
        for step in range(4):
            e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=3 + step, lengths=lens)
            e.loss_backward(b["target"])
            e.adam_step(1e-3, max_norm=5.0, overlap=ov)
            assert (e._upd_done is not None) == ov
            if step == 1:
                early = e.params.clone()            # read through the property: joins the update stream first
                torch.cuda.synchronize()
                assert torch.equal(early, e.params)
        torch.cuda.synchronize()
        res[ov] = [t.clone() for t in (e.params, e.adam_m, e.adam_v)] + ([e.params_lp.clone(), e.params_lp_t.clone()] if prec == "bf16" else [])
    for a, c in zip(res[False], res[True]):
        assert torch.equal(a, c)      # every precision, the embedding tables included: nothing in a train step depends on arrival order (round 6)


@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16x3", "bf16x3_fwd"])
def test_train_steps_are_bit_reproducible(prec):
    """a train step is bit-reproducible in every precision.  Until round 6 the gradients of the three embedding tables were scatter-added
    with float atomics (csrc/embed.hip): the 64 [CLS] rows of a batch met in one table row in arrival order, the last bits of the sums changed from
    run to run, and BertAdam's normalised update amplified that over a few steps.  Now every table row is summed by one workgroup in an order that
    depends on the batch alone, and the loss is folded in a fixed order: two RUNS (two engines) of three train steps - dropout on, packed rows, the
    optimizer pass beside the next forward, every question carrying the same [CLS] / [SEP] ids - agree bit for bit in logits, loss, gradients,
    parameters, Adam moments and operand copies."""
    cfg = FULL if prec != "f32" else MED
    B, T, O = (64, 20, 36) if prec != "f32" else (6, 12, 10)
    raw = synth.synth_batch(B, T, O=O, F=cfg["feat_dim"], NA=cfg["num_answers"], vocab=cfg["vocab_size"], seed=41, min_len=3)
    raw["input_ids"][: B // 2, 3] = raw["input_ids"][0, 3]          # one more id shared by half the batch, at the same position
    raw["segment_ids"][:, 2] = 1                                      # token type 1 rows (type 0 is padding_idx in LXMERT: no gradient)
    b = dev(raw)
    lens = np.ascontiguousarray(raw["lengths"], dtype=np.int32)
    runs = []
    for _ in range(2):
        e = make_engine(cfg, prec, dropout=0.1)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        e.enable_segment_sumsq(True)
        out = []
        for step in range(3):
            lg = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=11 + step, lengths=lens)[0].clone()
            loss = e.loss_backward(b["target"]).clone()
            out += [lg, loss, e.grads.clone()]
            e.adam_step(1e-3, max_norm=5.0)
        out += [e.params.clone(), e.adam_m.clone(), e.adam_v.clone()]
        if e.params_lp is not None:
            out += [e.params_lp.clone(), e.params_lp_t.clone()]
        torch.cuda.synchronize()
        runs.append(out)
        del e
    word = [sp for sp in make_engine(cfg, prec).specs if "word_embeddings" in sp.name][0]
    g = runs[0][2][word.offset:word.offset + word.numel].view(word.shape)
    assert float(g[101].abs().max()) > 0 and float(g[0].abs().max()) == 0.0      # the [CLS] row received its B contributions; padding_idx got none
    for i, (a, c) in enumerate(zip(*runs)):
        assert torch.equal(a, c), i


def test_layernorm_backward_reads_the_split_sums_in_place():
    """Round 6, bf16x3_fwd: the pre-LayerNorm sums no longer get a bf16 image from the projections' epilogues - the LayerNorm backward kernel reads the hi
    parts of the split-f32 tensor in place (norm.hip ln_bwd16_kernel<.., ZSF>; rgqa_debug_set key 21 = 0 restores the image).  hi = bf16(x) is exactly what
    the image held: logits and every gradient are bit-identical, packed and padded rows, the [CLS]-row tail included."""
    from rgqa_amd import _lib
    L = _lib.load()
    B, T, O = 24, 20, 36
    raw = synth.synth_batch(B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=23, min_len=3)
    b = dev(raw)
    lens = np.ascontiguousarray(raw["lengths"], dtype=np.int32)
    res = {}
    try:
        for inplace in (0, 1):
            assert L.rgqa_debug_set(21, inplace) == 0
            assert L.rgqa_debug_set(22, 0 if inplace == 0 else 12) == 0      # ... and the late start of the short blocks of a persistent GEMM launch (key 22) changes no bit either
            e = make_engine(FULL, "bf16x3_fwd", dropout=0.1)
            e.ensure_shape(B, T, O)
            e.sync_weights()
            out = []
            for packed in (True, False):
                lg = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=7, lengths=lens if packed else None)[0].clone()
                e.loss_backward(b["target"])
                out += [lg, e.grads.clone()]
            torch.cuda.synchronize()
            res[inplace] = out
    finally:
        L.rgqa_debug_set(21, 1)
        L.rgqa_debug_set(22, 8)
    assert float(res[1][1].abs().max()) > 0
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)


def test_rebind_right_after_an_overlapped_update(prec="bf16"):
    """ensure_shape() re-plans (or frees) the workspace, which holds the transpose descriptor table the LAST kernel of an optimizer pass
    still running on the update stream reads.  adam_step(overlap) -> ensure_shape(bigger) -> forward / backward must give what the serial order gives:
    ensure_shape joins the update first."""
    B, T, O = 24, 20, 36
    raw = synth.synth_batch(2 * B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=19, min_len=3)
    big = dev(raw)
    small = {k: v[:B].contiguous() for k, v in big.items()}
    res = {}
    for ov in (False, True):
        e = make_engine(FULL, prec, dropout=0.1)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        for step in range(2):
            e.forward(small["feats"], small["boxes"], small["input_ids"], small["input_mask"], small["segment_ids"], train=True, seed=3 + step)
            e.loss_backward(small["target"])
            e.adam_step(1e-3, max_norm=5.0, overlap=ov)
        if not ov:
            torch.cuda.synchronize()
        # no synchronisation in the overlapped arm: the re-bind itself must order the workspace's re-use behind the update
        e.ensure_shape(2 * B, T, O)
        lg = e.forward(big["feats"], big["boxes"], big["input_ids"], big["input_mask"], big["segment_ids"], train=True, seed=9)[0].clone()
        e.loss_backward(big["target"])
        torch.cuda.synchronize()
        res[ov] = (lg, e.grads.clone(), e.params.clone(), e.params_lp_t.clone())
    for a, c in zip(res[False], res[True]):
        assert torch.equal(a, c)


@pytest.mark.parametrize("B,varlen", [(48, True), (48, False), (256, True), (3, True)])
def test_layernorm_inside_the_projection_launch_is_the_separate_kernel(B, varlen):
    """Round 5: the LayerNorm behind every attention-output / FFN-output projection of the bf16 engine is done inside the projection's launch by the
    workgroup that finishes a row block's last tile (rgqa_debug_set key 19 = 1, opt-in; gemm256_dev.h nt256_ln_after_tile).  Same arithmetic in the
    same order as ln_fwd16_kernel: logits and every gradient equal the separate launches' bit for bit - three passes, so that a ticket left
    behind by a pass (the last arriver re-zeroes it) would show; row counts that end inside a row block and inside a half-wave trip included."""
    from rgqa_amd import _lib
    L = _lib.load()
    T, O = 20, 36
    raw = synth.synth_batch(B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=17, min_len=3)
    b = dev(raw)
    lens = np.ascontiguousarray(raw["lengths"], dtype=np.int32) if varlen else None
    res = {}
    try:
        for fuse in (0, 1):
            assert L.rgqa_debug_set(19, fuse) == 0
            e = make_engine(FULL, "bf16", dropout=0.1)
            e.ensure_shape(B, T, O)
            e.sync_weights()
            out = []
            for step in range(3):
                lg = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=5 + step, lengths=lens)[0].clone()
                e.loss_backward(b["target"])
                out.append((lg, e.grads.clone()))
            torch.cuda.synchronize()
            res[fuse] = out
    finally:
        L.rgqa_debug_set(19, 0)
    for (l0, g0), (l1, g1) in zip(res[0], res[1]):
        assert float(g0.abs().max()) > 0
        assert torch.equal(l0, l1)
        assert torch.equal(g0, g1)


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "f32"])
def test_merged_wgrad_launches_give_the_same_gradients(prec):
    """The weight-gradient problems of 1, 2, 3 or 4 backward periods in one launch (rgqa_debug_set key 6): the kernel per output tile is the
    same whatever shares the launch, so the gradients - including an accumulating second backward - are bit-identical."""
    from rgqa_amd import _lib
    L = _lib.load()
    cfg = FULL if prec != "f32" else MED
    B, T, O = (32, 20, 36) if prec != "f32" else (6, 12, 10)
    b = dev(synth.synth_batch(B, T, O=O, F=cfg["feat_dim"], NA=cfg["num_answers"], vocab=cfg["vocab_size"], seed=5, min_len=3))
    e = make_engine(cfg, prec, dropout=0.1)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    first = 0

    def grads(merge):
        assert L.rgqa_debug_set(6, merge) == 0
        try:
            run(e, b, True, 77)
            e.loss_backward(b["target"])
            one = e.grads[first:].clone()
            e.loss_backward(b["target"], accumulate=True)
            torch.cuda.synchronize()
            return one, e.grads[first:].clone()
        finally:
            L.rgqa_debug_set(6, 0)

    ref1, ref2 = grads(1)
    assert float(ref1.abs().max()) > 0 and float((ref2 - 2 * ref1).abs().max()) <= 1e-5 * float(ref1.abs().max())
    for merge in (2, 3, 4):
        g1, g2 = grads(merge)
        assert torch.equal(g1, ref1) and torch.equal(g2, ref2), merge


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("tag,T,prec,tol", [("small", 5, "f32", 2e-5), ("small", 8, "f32", 2e-5), ("full", 20, "f32", 1e-4), ("full", 20, "bf16x3", 1e-4),
                                            ("full", 20, "bf16", 3e-2)])
def test_cross_attention_probabilities_vs_golden(golden_dir, tag, T, prec, tol, packed):
    """rgqa_engine_get_cross_attention vs the reference's lxrt_vis `output_attention=True` vectors (g8): both directions
    of every stored cross layer; packed language rows give the same probabilities on the real tokens, zeros elsewhere
    (a padded KEY has probability exactly 0 in the reference too); the last layer's vision-query direction - dead in mode 'x',
    skipped by the forward pass - is projected on demand."""
    g = np.load(os.path.join(golden_dir, "g8_xatt.npz"))
    cfgd = SMALL if tag == "small" else FULL
    raw = small_batch(T) if tag == "small" else full_batch(T)
    b = dev(raw)
    B, O = raw["feats"].shape[0], raw["feats"].shape[1]
    lens = [int(v) for v in raw["input_mask"].sum(1)]
    e = make_engine(cfgd, prec)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens if packed else None)
    pre = "%s_T%d." % (tag, T)
    seen = 0
    for k in [k for k in g.files if k.startswith(pre + "x")]:
        layer, direction = int(k[len(pre) + 1]), k[-3:]
        got = e.cross_attention(layer, direction).cpu().numpy()
        ref = g[k]
        assert got.shape == ref.shape
        for i, n in enumerate(lens):
            if direction == "l2v":       # queries = language tokens
                np.testing.assert_allclose(got[i, :, :n], ref[i, :, :n], rtol=0, atol=tol, err_msg=k)
                if packed:
                    assert not got[i, :, n:].any()
                else:
                    np.testing.assert_allclose(got[i, :, n:], ref[i, :, n:], rtol=0, atol=tol, err_msg=k)
            else:                        # keys = language tokens: padded columns are 0 on both sides
                np.testing.assert_allclose(got[i], ref[i], rtol=0, atol=tol, err_msg=k)
                assert not got[i, :, :, n:].any()
        seen += 1
    assert seen >= 4
    with pytest.raises(RuntimeError):
        e.cross_attention(cfgd["x_layers"], "l2v")


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_config2_forward_b256_logits_vs_cpu(precision):
    """BASELINE config 2: forward-only inference at B=256 (full 9/5/5 architecture) in the two precisions inside the north star's bound
    (f32: exact FMA arithmetic, observed ~1e-5; bf16x3: the fast path, observed ~1e-4); samples are independent, so the CPU oracle is
    evaluated on a spread of 6 of the 256 rows and must agree within 1e-3."""
    from oracle import lxmert_ref as R
    B, T = 256, 20
    b = synth.synth_batch(B, T, seed=2024)
    e = make_engine(FULL, precision)
    d = dev(b)
    e.ensure_shape(B, T, 36)
    e.sync_weights()
    lg, pl = run(e, d)
    pick = [0, 1, 77, 128, 200, 255]
    cfg = R.RefConfig(**FULL)
    P = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(R.param_shapes(cfg)).items()}
    with torch.no_grad():
        lr, pr = R.gqa_forward(P, cfg, torch.from_numpy(b["feats"][pick]), torch.from_numpy(b["boxes"][pick]),
                               torch.from_numpy(b["input_ids"][pick]), torch.from_numpy(b["input_mask"][pick]))
    err = float((lg[pick].cpu() - lr).abs().max())
    _report("config 2 forward B=256 %s vs CPU oracle" % precision, logits_max=err, pooled_max=float((pl[pick].cpu() - pr).abs().max()))
    assert err <= 1e-3, err
    assert float((pl[pick].cpu() - pr).abs().max()) <= 1e-3
    assert torch.isfinite(lg).all()


# ---------------------------------------------------------------------------------------------- unpadded language rows
def _grads_by_name(e):
    return {sp.name: e.view(e.grads, sp).clone() for sp in e.specs}


@pytest.mark.parametrize("precision,ltol,gtol", [("f32", 2e-5, 2e-4), ("bf16x3", 2e-4, 1e-3), ("bf16", 6e-2, 8e-2), ("bf16x3_fwd", 2e-4, 8e-2)])
def test_varlen_matches_padded(precision, ltol, gtol):
    """rgqa_engine_set_lengths packs the language rows to the real tokens.  Padded positions are masked keys with probability
    exactly 0 and the pooler reads token 0, so logits, loss and every gradient must agree with the padded pass (f32: to
    rounding of the differently-tiled sums; bf16: to bf16 rounding), and switching back restores the padded result bit for bit."""
    B, T, O = 6, 12, 7
    raw = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=21, min_len=2)
    lengths = raw["lengths"].astype(np.int32)
    assert lengths.min() < T and (raw["input_mask"].sum(1) == lengths).all()
    b = dev(raw)
    e = make_engine(MED, precision)
    e.ensure_shape(B, T, O)
    e.sync_weights()

    def run_pass(lens):
        lg, pl = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)
        lg, pl = lg.clone(), pl.clone()
        loss = e.loss_backward(b["target"]).item()
        return lg, pl, loss, _grads_by_name(e)

    lg0, pl0, loss0, g0 = run_pass(None)
    lg1, pl1, loss1, g1 = run_pass(lengths)
    assert float((lg1 - lg0).abs().max()) <= ltol * max(1.0, float(lg0.abs().max()))
    assert float((pl1 - pl0).abs().max()) <= ltol
    assert abs(loss1 - loss0) <= ltol * max(1.0, abs(loss0))
    for name, ref in g0.items():
        tol = gtol * float(ref.abs().max()) + 1e-7
        assert float((g1[name] - ref).abs().max()) <= tol, name
    # packed activations are the valid rows of the padded ones
    n = int(lengths.sum())
    rows = np.concatenate([np.arange(T * i, T * i + lengths[i]) for i in range(B)])
    a1 = e.activation("x0_lang", n).cpu().numpy()
    c1 = e.activation("x1_lang", B).cpu().numpy()            # the final language output exists for the [CLS] rows only
    e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"])
    a0 = e.activation("x0_lang", B * T).cpu().numpy()[rows]
    c0 = e.activation("x1_lang", B).cpu().numpy()
    np.testing.assert_allclose(a1, a0, rtol=0, atol=(1e-4 if precision == "f32" else 0.1))
    np.testing.assert_allclose(c1, c0, rtol=0, atol=(1e-4 if precision == "f32" else 0.1))
    lg2, pl2, loss2, g2 = run_pass(None)
    assert torch.equal(lg2, lg0) and torch.equal(pl2, pl0)
    first = min(sp.offset for sp in e.specs if "embeddings.LayerNorm" in sp.name)
    for sp in e.specs:
        if sp.offset >= first:
            assert torch.equal(g2[sp.name], g0[sp.name]), sp.name


@pytest.mark.parametrize("precision", ["f32", "bf16", "bf16x3", "bf16x3_fwd"])
@pytest.mark.parametrize("train", [False, True])
def test_cls_only_tail_matches_full_rows(precision, train):
    """Only token 0 of the last language FFN is consumed (modeling.py:575-581): running that sub-block on the B [CLS] rows
    (default) gives the logits and every gradient of the all-rows computation (rgqa_debug_set key 8 = 0), in both layouts.  Train
    mode runs with attention dropout only: the hidden-dropout draws of the tail's own site are indexed by the row number inside the
    launch, which the compact layout changes (any draw is a valid dropout mask; it is the same in forward and backward)."""
    from rgqa_amd import _lib
    L = _lib.load()
    B, T, O = 6, 12, 7
    raw = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=33, min_len=2)
    b = dev(raw)
    lens = raw["lengths"].astype(np.int32)
    from rgqa_amd.engine import Engine
    e = Engine(precision=precision, hidden_dropout=0.0, attn_dropout=0.1 if train else 0.0, **MED).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    e.ensure_shape(B, T, O)
    e.sync_weights()
    res = {}
    try:
        for mode in (0, 1):
            assert L.rgqa_debug_set(8, mode) == 0
            for packed in (False, True):
                lg, _ = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=train, seed=5, lengths=lens if packed else None)
                lg = lg.clone()
                loss = e.loss_backward(b["target"]).item()
                res[(mode, packed)] = (lg, loss, _grads_by_name(e))
    finally:
        L.rgqa_debug_set(8, -1)
    tol = 1e-6       # observed: logits identical, gradients within 1.2e-7 of their largest entry (different GEMM tile shapes / row counts)
    for packed in (False, True):
        lg0, loss0, g0 = res[(0, packed)]
        lg1, loss1, g1 = res[(1, packed)]
        assert float((lg1 - lg0).abs().max()) <= tol * max(1.0, float(lg0.abs().max()))
        assert abs(loss1 - loss0) <= tol * max(1.0, abs(loss0))
        worst = 0.0
        for name, ref in g0.items():
            den = float(ref.abs().max())
            if den > 0:
                worst = max(worst, float((g1[name] - ref).abs().max()) / den)
        _report("cls tail vs all rows %s train=%s packed=%s" % (precision, train, packed), logits_max=float((lg1 - lg0).abs().max()), grad_rel_max=worst)
        assert worst <= 1e-6


@pytest.mark.parametrize("T", [5, 8])
def test_varlen_f32_small_vs_golden(golden_dir, T):
    """The reference's own outputs (fixtures generated by running it) reproduced from the packed layout."""
    g = np.load(os.path.join(golden_dir, "g1_small_T%d.npz" % T))
    e = make_engine(SMALL, "f32")
    raw = small_batch(T)
    b = dev(raw)
    e.ensure_shape(3, T, 6)
    e.sync_weights()
    lens = raw["input_mask"].sum(1).astype(np.int32)
    lg, pl = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)
    np.testing.assert_allclose(lg.cpu().numpy(), g["logits"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(pl.cpu().numpy(), g["pooled"], rtol=0, atol=1e-4)
    loss = e.loss_backward(b["target"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    dead = set(g["dead"].tolist())
    for sp in e.specs:
        if sp.name in dead:
            continue
        ref = g["grad." + sp.name]
        np.testing.assert_allclose(e.view(e.grads, sp).cpu().numpy(), ref, rtol=2e-3, atol=1e-6 + 2e-4 * np.abs(ref).max(), err_msg=sp.name)


def test_set_lengths_rejects_bad_input():
    from rgqa_amd import _lib
    e = make_engine(MED, "f32")
    e.ensure_shape(4, 8, 6)
    L = e.lib
    import ctypes as C
    bad_n = np.array([3, 4, 5], dtype=np.int32)
    assert L.rgqa_engine_set_lengths(e.h, C.c_void_p(bad_n.ctypes.data), 3) != 0
    assert b"lengths" in L.rgqa_last_error_string()
    for v in (0, 9):
        bad = np.array([3, v, 5, 8], dtype=np.int32)
        assert L.rgqa_engine_set_lengths(e.h, C.c_void_p(bad.ctypes.data), 4) != 0
    ok = np.array([3, 8, 1, 2], dtype=np.int32)
    assert L.rgqa_engine_set_lengths(e.h, C.c_void_p(ok.ctypes.data), 4) == 0
    assert L.rgqa_engine_set_lengths(e.h, None, 0) == 0


@pytest.mark.parametrize("precision,tol", [("f32", 2e-4), ("bf16", 6e-2)])
def test_input_gradients_vs_golden(golden_dir, precision, tol):
    """dL/dfeats and dL/dboxes (rgqa_engine_set_input_grads; what the reference's ODIN scorer differentiates, tasks/gqa_odin.py:97-121)
    against the reference's own autograd result stored in the G1 fixture; parameter gradients are unchanged by asking for them."""
    T = 8
    g = np.load(os.path.join(golden_dir, "g1_small_T%d.npz" % T))
    e = make_engine(SMALL, precision)
    raw = small_batch(T)
    b = dev(raw)
    e.ensure_shape(3, T, 6)
    e.sync_weights()
    run(e, b)
    e.loss_backward(b["target"])
    g0 = e.grads.clone()
    dfeats = torch.full((3 * 6, SMALL["feat_dim"]), 7.0, device="cuda")
    dboxes = torch.full((3 * 6, 4), 7.0, device="cuda")
    e.set_input_grads(dfeats, dboxes)
    run(e, b)
    e.loss_backward(b["target"])
    e.set_input_grads(None, None)
    first = min(sp.offset for sp in e.specs if "embeddings.LayerNorm" in sp.name)
    assert torch.equal(e.grads[first:], g0[first:])
    rf, rb = g["dfeats"].reshape(18, -1), g["dboxes"].reshape(18, 4)
    np.testing.assert_allclose(dfeats.cpu().numpy(), rf, rtol=0, atol=tol * np.abs(rf).max())
    np.testing.assert_allclose(dboxes.cpu().numpy(), rb, rtol=0, atol=tol * np.abs(rb).max())
    assert np.abs(rf).max() > 0 and np.abs(rb).max() > 0


def test_input_gradients_bf16x3_vs_f32_engine():
    """dL/dfeats, dL/dboxes in the split-f32 precision (the ODIN scorer's pass, tasks/gqa_odin.py:97-121) against the exact-f32 engine on the
    medium config (head size 64: the MFMA kernels), packed rows: within 1e-3 of the largest entry."""
    B, T, O = 6, 12, 10
    b = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=5, min_len=2)
    d = dev(b)
    lens = [int(v) for v in b["input_mask"].sum(1)]
    out = {}
    for prec in ("f32", "bf16x3"):
        e = make_engine(MED, prec)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        dfeats = torch.zeros(B * O, MED["feat_dim"], device="cuda")
        dboxes = torch.zeros(B * O, 4, device="cuda")
        e.set_input_grads(dfeats, dboxes)
        e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], lengths=lens)
        e.loss_backward(d["target"])
        e.set_input_grads(None, None)
        out[prec] = (dfeats.cpu().numpy(), dboxes.cpu().numpy())
    for got, ref, name in ((out["bf16x3"][0], out["f32"][0], "dfeats"), (out["bf16x3"][1], out["f32"][1], "dboxes")):
        assert np.abs(ref).max() > 0
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print("bf16x3 %s vs f32 engine: max err / max |ref| = %.2e" % (name, err))
        assert err < 1e-3, (name, err)


SWEEP = [   # (B, T, O, l, x, r, heads, hidden, answers, packed)
    (1, 3, 1, 1, 1, 1, 1, 64, 8, True),
    (2, 20, 36, 2, 1, 0, 2, 128, 9, True),       # no r-layers: visn goes straight into the cross layers; odd answer count
    (5, 7, 5, 0, 2, 2, 2, 128, 64, False),       # no l-layers
    (3, 33, 17, 1, 1, 3, 2, 128, 130, True),     # T > 32: three query tiles; r deeper than l
    (7, 2, 64, 2, 2, 1, 4, 256, 24, True),       # shortest possible questions ([CLS][SEP]) on a maximal RoI count
]


@pytest.mark.parametrize("precision,tol,gtol", [("f32", 2e-4, 3e-3), ("bf16x3", 3e-4, 3e-3), ("bf16", 2e-2, 4.5e-2)])     # bf16 observed: logits <= 1.0e-2, worst tensor <= 2.2e-2
@pytest.mark.parametrize("shape", SWEEP, ids=lambda s: "B%dT%dO%d_l%dx%dr%d_h%d" % (s[0], s[1], s[2], s[3], s[4], s[5], s[6]))
def test_shape_sweep_vs_oracle(shape, precision, tol, gtol):
    """Edge shapes of the engine (single sample, no l- or r-layers, T beyond two query tiles, 2-token questions, 64 RoIs, answer
    counts that are not multiples of 8) against the oracle: logits, loss and every gradient, padded or packed language rows."""
    B, T, O, l, x, r, heads, hidden, na, packed = shape
    F = 64 if precision == "bf16x3" else 48          # split-f32 rows are whole 128-byte lines: feature size a multiple of 32
    cfgd = dict(vocab_size=300, hidden=hidden, heads=heads, inter=2 * hidden, max_pos=64, type_vocab=2, l_layers=l, x_layers=x, r_layers=r,
                feat_dim=F, pos_dim=4, num_answers=na)
    b = synth.synth_batch(B, T, O=O, F=F, NA=na, vocab=300, seed=100 + B, min_len=2)
    lg_r, pl_r, loss_r, Pr = oracle_run(cfgd, b)
    e = make_engine(cfgd, precision)
    d = dev(b)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    lens = b["lengths"].astype(np.int32) if packed else None
    lg, pl = e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], lengths=lens)
    lerr = float((lg.cpu() - lg_r).abs().max())
    assert lerr < tol * max(1.0, float(lg_r.abs().max()))
    assert float((pl.cpu() - pl_r).abs().max()) < tol
    loss = e.loss_backward(d["target"])
    assert abs(loss.item() - loss_r) < 50 * tol * max(1.0, abs(loss_r))
    gworst = 0.0
    for sp in e.specs:
        got = e.view(e.grads, sp).cpu()
        ref = Pr[sp.name].grad
        if ref is None or sp.dead:
            assert float(got.abs().max()) == 0.0, sp.name
            continue
        den = float(ref.norm())
        if den < 1e-8:      # mathematically zero gradients (key biases: softmax is shift-invariant): rounding noise only
            assert float(got.norm()) < 1e-3 * gtol, sp.name
            continue
        gworst = max(gworst, float((got - ref).norm()) / den)
        assert float((got - ref).norm()) / den < gtol, (sp.name, float((got - ref).norm()) / den)
    _report("shape sweep %s %s vs oracle" % ("B%dT%dO%d_l%dx%dr%d_h%d" % shape[:7], precision), logits_max=lerr, worst_tensor_rel=gworst)


def test_training_converges_bf16_like_f32():
    """End to end through the C ABI: 80 train steps (dropout 0.1, clip 5, BertAdam with linear warm-up) on one fixed batch with packed
    language rows.  Both precisions drive the BCE x NA loss down by more than 3x, and the bf16 trajectory stays near the f32 one."""
    B, T, O = 16, 12, 9
    raw = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=77, min_len=3, uq_frac=0.0)
    lens = raw["lengths"].astype(np.int32)
    b = dev(raw)
    curves = {}
    for precision in ("f32", "bf16", "bf16x3"):
        e = make_engine(MED, precision, dropout=0.1)
        e.ensure_shape(B, T, O)
        e.sync_weights()
        losses = []
        for step in range(80):
            e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=1000 + step, lengths=lens)
            losses.append(e.loss_backward(b["target"]).item())
            x = step / 200.0
            e.adam_step(2e-3 * (x / 0.1 if x < 0.1 else max((x - 1.0) / (0.1 - 1.0), 0.0)), max_norm=5.0)
        assert all(np.isfinite(losses)), precision
        curves[precision] = losses
        # medians: once the toy model has memorised the batch, an occasional dropout mask (hidden + attention dropout together) sends the
        # train-mode loss of ONE step from 3.6 to 70..140 and back - a property of that mask on those weights, reproduced exactly by the
        # f32 engine from the same weights and seed in both layouts and absent in eval mode (measured in round 2); which step it
        # hits depends on the last bits of the trajectory, so a mean over the final steps is not a stable statistic
        assert np.median(losses[-9:]) < np.median(losses[:3]) / 3.0, (precision, losses[:3], losses[-9:])
    f = np.median(curves["f32"][-11:])
    for other in ("bf16", "bf16x3"):
        h = np.median(curves[other][-11:])
        assert abs(f - h) < 0.35 * max(f, h), (other, f, h)


def test_split_k_visual_projection_wgrad():
    """B*O >= 2048 rows: the visual projection's weight gradient is computed as S split-K partials folded in a fixed order
    (bf16 engine).  Checked against the oracle and for run-to-run bit-reproducibility."""
    B, T, O = 64, 8, 36
    b = synth.synth_batch(B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=31, min_len=2)
    lg_r, pl_r, loss_r, Pr = oracle_run(MED, b)
    e = make_engine(MED, "bf16")
    d = dev(b)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    name = "lxrt_encoder.model.bert.encoder.visn_fc.visn_fc.weight"
    sp = [x for x in e.specs if x.name == name][0]
    got = []
    for _ in range(2):
        run(e, d)
        e.loss_backward(d["target"])
        got.append(e.view(e.grads, sp).clone())
    assert torch.equal(got[0], got[1])
    ref = Pr[name].grad
    assert float((got[0].cpu() - ref).norm() / ref.norm()) < 8e-2


def test_segment_sumsq_matches_full_norm():
    """rgqa_engine_set_grad_sumsq_slots: the per-segment sums backward leaves behind add up to the sum of squares of the live gradient
    ranges (what clip_grad_norm_ measures), in both stream configurations; an optimizer step taken from them equals the step taken from
    the full-arena reduction up to the f32 rounding of the norm."""
    from rgqa_amd import _lib
    L = _lib.load()
    B, T, O = 32, 20, 36
    raw = synth.synth_batch(B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=15, min_len=3)
    lens = [int(v) for v in raw["input_mask"].sum(1)]
    b = dev(raw)
    e = make_engine(FULL, "bf16", dropout=0.1)
    e.ensure_shape(B, T, O)
    e.sync_weights()
    e.enable_segment_sumsq(True)
    for serial in (0, 1):
        assert L.rgqa_debug_set(2, serial) == 0
        try:
            e._seg_sumsq.fill_(-1.0)
            e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], train=True, seed=3, lengths=lens)
            e.loss_backward(b["target"])
            torch.cuda.synchronize()
        finally:
            L.rgqa_debug_set(2, 0)
        assert float(e._seg_sumsq.min()) >= 0.0
        full = float(e.grad_norm().item()) ** 2
        np.testing.assert_allclose(float(e._seg_sumsq.double().sum()), full, rtol=1e-5)
    p0 = e.params.clone()
    e.adam_step(1e-3, max_norm=0.5)                      # uses the slots (valid after loss_backward)
    p_slots = e.params.clone()
    e.params.copy_(p0); e.adam_m.zero_(); e.adam_v.zero_()
    e.invalidate_segment_sumsq()
    e.adam_step(1e-3, max_norm=0.5)                      # full-arena reduction
    assert float((e.params - p_slots).abs().max()) <= 1e-6 * float(p0.abs().max()) + 1e-9
    assert float((p_slots - p0).abs().max()) > 0


def test_full_size_batch_independence_and_roi_permutation():
    """Size-independent properties at BASELINE's full size (B=256, T=20, 36 RoIs, 9/5/5 layers, bf16, eval):
    (1) samples are independent - a sample's logits in the batch of 256 equal its logits in a batch of 3 (same per-row arithmetic
        whatever the tile shapes; packed language rows in both);
    (2) the encoder has no RoI order: permuting a sample's 36 (feature, box) pairs changes only the order of the attention sums."""
    B, T, O = 256, 20, 36
    raw = synth.synth_batch(B, T, O=O, F=FULL["feat_dim"], NA=FULL["num_answers"], vocab=FULL["vocab_size"], seed=77, min_len=3)
    lens = [int(v) for v in raw["input_mask"].sum(1)]
    b = dev(raw)
    e = make_engine(FULL, "bf16")
    e.ensure_shape(B, T, O)
    e.sync_weights()
    full = e.forward(b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)[0].clone()
    assert bool(torch.isfinite(full).all())
    pick = [0, 131, 255]
    sub = {k: v[pick].contiguous() for k, v in b.items()}
    small = e.forward(sub["feats"], sub["boxes"], sub["input_ids"], sub["input_mask"], sub["segment_ids"], lengths=[lens[i] for i in pick])[0].clone()
    scale = float(full.abs().max())
    assert float((small - full[pick]).abs().max()) <= 2e-3 * scale          # observed: bit-equal or last-bit differences of bf16 activations
    # RoI permutation of every sample
    perm = torch.stack([torch.randperm(O, generator=torch.Generator().manual_seed(i)) for i in range(B)]).cuda()
    feats_p = torch.gather(b["feats"], 1, perm[:, :, None].expand(-1, -1, b["feats"].shape[2])).contiguous()
    boxes_p = torch.gather(b["boxes"], 1, perm[:, :, None].expand(-1, -1, 4)).contiguous()
    permuted = e.forward(feats_p, boxes_p, b["input_ids"], b["input_mask"], b["segment_ids"], lengths=lens)[0]
    assert float((permuted - full).abs().max()) <= 3e-2 * scale
    assert float((permuted - full).abs().mean()) <= 3e-3 * scale


