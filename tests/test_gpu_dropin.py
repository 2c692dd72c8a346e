"""The drop-in Python surface (tasks.gqa_model.GQAModel, lxrt.entry.LXRTEncoder, lxrt.optimization.BertAdam) driven the
way the reference's trainer drives it (tasks/gqa_conf.py:150-202), checked against the oracle on a real MI355X."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(vocab_size=80, hidden=128, heads=2, inter=256, max_pos=64, type_vocab=2, l_layers=2, x_layers=2, r_layers=1,
           feat_dim=64, pos_dim=4, num_answers=17)
SENTS = ["What color is the dog?", "Is the man to the left of the woman?", "who is holding the bottle", "unaffable", "?",
         "Is there a dog in front of the table behind the man on the right side of the photo to the left of the woman holding the red bottle"]


@pytest.fixture()
def env(golden_dir, monkeypatch):
    monkeypatch.setenv("RGQA_BERT_VOCAB", os.path.join(golden_dir, "g4_vocab.txt"))
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    import rgqa_amd.lxrt.modeling as M
    monkeypatch.setattr(M.VISUAL_CONFIG, "visual_feat_dim", CFG["feat_dim"])
    monkeypatch.setattr(M.LXRTFeatureExtraction, "from_pretrained", classmethod(
        lambda cls, name, **kw: cls(M.BertConfig(CFG["vocab_size"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"],
                                                 intermediate_size=CFG["inter"], max_position_embeddings=CFG["max_pos"],
                                                 hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), **kw)))
    yield M
    sys.path.remove(os.path.join(ROOT, "dropin"))


def build(precision, T=20):
    from tasks.gqa_model import GQAModel
    from rgqa_amd import synth
    os.environ["RGQA_PRECISION"] = precision
    args = types.SimpleNamespace(llayers=CFG["l_layers"], xlayers=CFG["x_layers"], rlayers=CFG["r_layers"], from_scratch=False)
    m = GQAModel(CFG["num_answers"], max_seq_length=T, model_args=args)
    filled = synth.fill_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()})     # on the host, BEFORE .cuda(), as gqa_conf.py:97-110 does
    return m.cuda(), filled


def oracle(filled, feats, boxes, ids, mask, target):
    from oracle import lxmert_ref as R
    cfg = R.RefConfig(**CFG)
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in filled.items()}
    lg, pooled = R.gqa_forward(P, cfg, feats, boxes, ids, mask)
    loss = R.bce_loss(lg, target)
    loss.backward()
    return lg.detach(), pooled.detach(), loss.item(), P


def batch(T=20):
    from rgqa_amd import synth
    B = len(SENTS)
    b = synth.synth_batch(B, T, O=7, F=CFG["feat_dim"], NA=CFG["num_answers"], vocab=CFG["vocab_size"], seed=21)
    return torch.from_numpy(b["feats"]), torch.from_numpy(b["boxes"]), torch.from_numpy(b["target"])


def test_gqa_model_train_steps_match_oracle(env, golden_dir):
    """zero_grad -> model(feats, boxes, sent) -> BCE*NA -> backward -> clip_grad_norm_(5.) -> BertAdam.step, three times
    (f32 precision), against the oracle's functional model + BertAdamRef on identical weights and inputs."""
    from oracle import lxmert_ref as R
    from lxrt.optimization import BertAdam
    T = 20
    m, filled = build("f32", T)
    feats, boxes, target = batch(T)
    vocab = {w.rstrip("\n"): i for i, w in enumerate(open(os.path.join(golden_dir, "g4_vocab.txt"), encoding="utf-8"))}
    ids, mask, _ = R.sents_to_features(SENTS, T, vocab)
    ids, mask = torch.tensor(ids), torch.tensor(mask)
    optim = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=20)
    Pref = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in filled.items()}
    ref_opt = R.BertAdamRef(list(Pref.values()), lr=1e-3, warmup=0.1, t_total=20)
    cfg = R.RefConfig(**CFG)
    bce = torch.nn.BCEWithLogitsLoss()
    m.eval()    # parity runs use dropout off (the reference's dropout stream cannot be reproduced); grads still flow
    for step in range(3):
        optim.zero_grad()
        logit = m(feats.cuda(), boxes.cuda(), SENTS)
        assert logit.dim() == 2 and logit.shape == (len(SENTS), CFG["num_answers"])
        loss = bce(logit, target.cuda()) * logit.size(1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.)
        optim.step()
        batch_ref = dict(feats=feats, boxes=boxes, input_ids=ids, input_mask=mask, target=target)
        loss_ref = R.train_step(Pref, cfg, batch_ref, ref_opt)
        assert abs(loss.item() - loss_ref) < 1e-3 * max(1.0, abs(loss_ref)), (step, loss.item(), loss_ref)
    sd = m.state_dict()
    dead = 0
    for k, p in Pref.items():
        got = sd[k].cpu()
        if ".x_layers.1.visn_" in k:       # never receive gradients in mode 'x': untouched by the optimizer in both
            dead += 1
            assert torch.equal(got, torch.from_numpy(filled[k])), k
            continue
        np.testing.assert_allclose(got.numpy(), p.detach().numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    assert dead == 16
    named = dict(m.named_parameters())
    assert all(named[k].grad is None for k in named if ".x_layers.1.visn_" in k)


@pytest.mark.parametrize("variant", ["mce_loss", "sample_pair"])
def test_trainer_variants_through_the_dropin_model(env, golden_dir, variant):
    """VERDICT r5 #8 / missing #6: the two switches of the reference trainer that change what surrounds the model (tasks/gqa_conf.py): `--mceLoss`
    (:193-196: CrossEntropyLoss(ignore_index=-1) over the returned logits x NA - torch autograd over the drop-in's output) and `--sample_pair`
    (:155-170: the batch is doubled - features and boxes repeated, a negative question per sample, all-zero targets for the second half).  One step
    of each through GQAModel (f32) against the oracle's functional model on identical weights and inputs: logits, loss, every live gradient."""
    from oracle import lxmert_ref as R
    T = 20
    m, filled = build("f32", T)
    feats, boxes, target = batch(T)
    sent = list(SENTS)
    if variant == "sample_pair":
        perm = [(j + 2) % len(SENTS) for j in range(len(SENTS))]                      # the in-batch negative question of sample j
        feats, boxes = feats.repeat(2, 1, 1), boxes.repeat(2, 1, 1)                    # gqa_conf.py:166-167
        target = torch.cat([target, torch.zeros_like(target)], 0)                      # :168
        sent = sent + [SENTS[j] for j in perm]                                         # :169
    vocab = {w.rstrip("\n"): i for i, w in enumerate(open(os.path.join(golden_dir, "g4_vocab.txt"), encoding="utf-8"))}
    ids, mask, _ = R.sents_to_features(sent, T, vocab)
    ids, mask = torch.tensor(ids), torch.tensor(mask)

    def loss_of(logit, tgt):
        if variant == "mce_loss":
            _, cls = tgt.max(1)                                                        # :194
            return torch.nn.CrossEntropyLoss(ignore_index=-1)(logit, cls) * logit.size(1)
        return torch.nn.BCEWithLogitsLoss()(logit, tgt) * logit.size(1)

    cfg = R.RefConfig(**CFG)
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in filled.items()}
    lg_r, _ = R.gqa_forward(P, cfg, feats, boxes, ids, mask)
    loss_r = loss_of(lg_r, target)
    loss_r.backward()
    m.eval()          # dropout off (the reference's dropout stream cannot be reproduced); gradients still flow
    m.zero_grad()
    logit = m(feats.cuda(), boxes.cuda(), sent)
    assert logit.shape == (len(sent), CFG["num_answers"])
    np.testing.assert_allclose(logit.detach().cpu().numpy(), lg_r.detach().numpy(), rtol=0, atol=1e-4)
    loss = loss_of(logit, target.cuda())
    loss.backward()
    assert abs(loss.item() - loss_r.item()) < 1e-4 * max(1.0, abs(loss_r.item()))
    got_norm = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 5.))
    ref_norm = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in P.values() if q.grad is not None)))
    assert abs(got_norm - ref_norm) < 1e-3 * ref_norm
    coef = min(1.0, 5.0 / (ref_norm + 1e-6))
    checked = 0
    for k, p in m.named_parameters():
        if ".x_layers.1.visn_" in k:
            assert p.grad is None
            continue
        ref = P[k].grad
        np.testing.assert_allclose(p.grad.cpu().numpy(), (ref * coef).numpy(), rtol=2e-3, atol=1e-6 + 2e-4 * float(ref.abs().max()) * coef, err_msg=k)
        checked += 1
    assert checked > 60


def test_gqa_model_bf16_forward_and_grads(env, golden_dir):
    m, filled = build("bf16", 30)
    from oracle import lxmert_ref as R
    feats, boxes, target = batch(30)
    vocab = {w.rstrip("\n"): i for i, w in enumerate(open(os.path.join(golden_dir, "g4_vocab.txt"), encoding="utf-8"))}
    ids, mask, _ = R.sents_to_features(SENTS, 30, vocab)
    lg_r, _, loss_r, Pr = oracle(filled, feats, boxes, torch.tensor(ids), torch.tensor(mask), target)
    m.eval()
    with torch.no_grad():
        lg0 = m(feats.cuda(), boxes.cuda(), SENTS)
    print("drop-in bf16: logits max err %.3e" % float((lg0.cpu() - lg_r).abs().max()))
    assert not lg0.requires_grad and float((lg0.cpu() - lg_r).abs().max()) < 1.3e-2      # observed 6.4e-3
    lg = m(feats.cuda(), boxes.cuda(), SENTS)
    assert lg.requires_grad and torch.equal(lg, lg0)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(lg, target.cuda()) * lg.size(1)
    loss.backward()
    w = "lxrt_encoder.model.bert.encoder.x_layers.0.visual_attention.att.query.weight"
    g = dict(m.named_parameters())[w].grad.cpu()
    print("drop-in bf16: grad rel err %.3e" % float((g - Pr[w].grad).norm() / Pr[w].grad.norm()))
    assert float((g - Pr[w].grad).norm() / Pr[w].grad.norm()) < 3e-2       # observed 1.4e-2
    # gradient accumulation: a second backward without zero_grad doubles the gradient
    lg2 = m(feats.cuda(), boxes.cuda(), SENTS)
    (torch.nn.functional.binary_cross_entropy_with_logits(lg2, target.cuda()) * lg2.size(1)).backward()
    g2 = dict(m.named_parameters())[w].grad.cpu()
    assert float((g2 - 2 * g).norm() / g.norm()) < 1e-2


def test_encoder_alone_under_a_foreign_head(env):
    """LXRTEncoder used the way tasks/vqa_model.py / nlvr2_model.py use it: pooled output feeds a caller-owned torch head;
    gradients flow back through rgqa_engine_backward_pooled."""
    from lxrt.entry import LXRTEncoder
    from rgqa_amd import synth
    from oracle import lxmert_ref as R
    os.environ["RGQA_PRECISION"] = "f32"
    args = types.SimpleNamespace(llayers=CFG["l_layers"], xlayers=CFG["x_layers"], rlayers=CFG["r_layers"], from_scratch=False)
    enc = LXRTEncoder(args, max_seq_length=20)
    filled = synth.fill_state_dict({"lxrt_encoder.model." + k: tuple(v.shape) for k, v in enc.model.state_dict().items()})
    enc.model.load_state_dict({k[len("lxrt_encoder.model."):]: torch.from_numpy(v) for k, v in filled.items()})
    enc = enc.cuda().eval()
    head = torch.nn.Linear(CFG["hidden"], 5).cuda()
    feats, boxes, _ = batch(20)
    x = enc(SENTS, (feats.cuda(), boxes.cuda()))
    assert x.shape == (len(SENTS), CFG["hidden"]) and x.requires_grad
    (head(x) ** 2).sum().backward()
    assert head.weight.grad is not None
    # oracle: same encoder, same head
    cfg = R.RefConfig(**CFG)
    full = dict(filled)
    for k, shp in R.param_shapes(cfg).items():
        if k.startswith("logit_fc."):
            full[k] = np.zeros(shp, dtype=np.float32)
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in full.items()}
    vocab = enc.tokenizer.vocab
    ids, mask, _ = R.sents_to_features(SENTS, 20, vocab)
    _, _, pooled = R.encoder_forward(P, cfg, torch.tensor(ids), torch.zeros(len(SENTS), 20, dtype=torch.long), torch.tensor(mask), feats, boxes)
    np.testing.assert_allclose(x.detach().cpu().numpy(), pooled.detach().numpy(), rtol=0, atol=2e-5)
    hw, hb = head.weight.detach().cpu(), head.bias.detach().cpu()
    ((pooled @ hw.t() + hb) ** 2).sum().backward()
    w = "lxrt_encoder.model.bert.encoder.layer.0.attention.self.value.weight"
    got = dict(enc.model.named_parameters())[w[len("lxrt_encoder.model."):]].grad.cpu()
    np.testing.assert_allclose(got.numpy(), P[w].grad.numpy(), rtol=2e-3, atol=1e-6 + 2e-4 * float(P[w].grad.abs().max()))
    # save / load round trip (entry.py:122-152)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rgqa_gpu_test")
    enc.save(path)
    before = {k: v.clone() for k, v in enc.model.state_dict().items()}
    with torch.no_grad():
        for p in enc.model.parameters():
            p.add_(1.0)
    enc.load(path)
    after = enc.model.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    x2 = enc(SENTS, (feats.cuda(), boxes.cuda()))
    assert torch.allclose(x2, x)            # bf16/f32 weight copies were refreshed after the in-place load
    os.remove(path + "_LXRT.pth")


def test_trainer_step_with_the_update_beside_the_next_forward(env, monkeypatch):
    """The drop-in BertAdam.step hands the update to the engine's pass beside the next forward (round 5; RGQA_ADAM_OVERLAP=0 keeps it on the
    step's stream) when it covers every live parameter: the unchanged trainer loop then ends in the same parameters bit for bit, a
    state_dict() taken right after optimizer.step() - no synchronisation, no forward in between - holds the finished update (the module's
    state_dict hook joins the update stream), and zero_grad(set_to_none=False) cannot overtake the update that still reads the gradients."""
    import lxrt.entry  # noqa: F401
    from lxrt.optimization import BertAdam
    feats, boxes, target = batch(20)
    res = {}
    for ov in ("0", "1"):
        monkeypatch.setenv("RGQA_ADAM_OVERLAP", ov)
        m, _ = build("bf16", 20)
        m.train()
        eng = m.lxrt_encoder.model._binding.engine
        assert eng.adam_overlap == (ov == "1")
        optim = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=20)
        snaps = []
        for step in range(4):
            optim.zero_grad(set_to_none=(step % 2 == 0))
            logit = m(feats.cuda(), boxes.cuda(), SENTS)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 5.)
            optim.step()
            assert (eng._upd_done is not None) == (ov == "1")          # the pass really runs beside what follows
            if step == 2:
                snaps.append({k: v.clone() for k, v in m.state_dict().items()})
                torch.cuda.synchronize()
                snaps.append({k: v.clone() for k, v in m.state_dict().items()})
            if step == 3:       # ... and so does `.data`, the way loggers and checkpoint code usually reach a parameter (ArenaParameter.data joins too)
                early = {k: p.data.clone() for k, p in m.named_parameters()}
                torch.cuda.synchronize()
                assert all(torch.equal(early[k], p.detach()) for k, p in m.named_parameters())
        assert all(torch.equal(snaps[0][k], snaps[1][k]) for k in snaps[0])
        torch.cuda.synchronize()
        res[ov] = {k: v.detach().clone() for k, v in m.named_parameters()}
    for k in res["0"]:
        assert torch.equal(res["0"][k], res["1"][k]), k      # the embedding tables too: their gradients are summed in a fixed order (round 6)


@pytest.mark.parametrize("precision", ["f32", "bf16", "bf16x3", "bf16x3_fwd"])
def test_trainer_steps_are_bit_reproducible(env, precision):
    """VERDICT r5 #4, at the drop-in surface: two runs of three steps of the reference trainer's loop (tasks/gqa_conf.py:174-202: zero_grad, model(feats,
    boxes, sent), BCE x NA, backward, clip_grad_norm_, BertAdam.step) end in torch.equal parameters, losses and clip norms in all four precisions -
    the embedding tables' gradients are summed in a fixed order (csrc/embed.hip), nothing in a step depends on arrival order any more."""
    import lxrt.entry  # noqa: F401
    from lxrt.optimization import BertAdam
    feats, boxes, target = batch(20)
    runs = []
    for _ in range(2):
        m, _ = build(precision, 20)
        m.train()
        optim = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=20)
        rec = []
        for step in range(3):
            optim.zero_grad()
            logit = m(feats.cuda(), boxes.cuda(), SENTS)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)
            loss.backward()
            norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.)
            optim.step()
            rec += [logit.detach().clone(), loss.detach().clone(), norm.detach().clone()]
        torch.cuda.synchronize()
        runs.append((rec, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    for a, c in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, c)
    assert all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_trainer_step_fused_clip_and_operand_copies(env, precision, monkeypatch):
    """The unchanged trainer's `nn.utils.clip_grad_norm_` + `BertAdam.step` (tasks/gqa_conf.py:201-202) through the drop-in's fast paths:
    importing lxrt.entry routes torch.nn.utils.clip_grad_norm_ through lxrt.optimization.clip_grad_norm_ (norm from the sums backward
    left per gradient segment, one rescale kernel), and BertAdam.step lets the update kernel re-write the engine's operand copies.
    Against the same steps with torch's own clip: same norm, same parameters; and the operand copies the fused path maintains equal a
    full re-cast of the updated weights bit for bit (same logits before and after Engine.sync_weights)."""
    import lxrt.entry  # noqa: F401
    from lxrt.optimization import BertAdam, clip_grad_norm_ as fast_clip, _torch_clip_grad_norm_ as torch_clip
    feats, boxes, target = batch(20)
    res = {}
    for which in ("fast", "inplace", "torch"):
        # "fast": the default - the rescale is deferred and folded into BertAdam's update kernel; "inplace": RGQA_DEFER_CLIP=0, the fast norm with
        # the rescale kernel run at the clip call (rounds 3-4); "torch": torch's own clip_grad_norm_
        monkeypatch.setenv("RGQA_DEFER_CLIP", "0" if which == "inplace" else "1")
        m, _ = build(precision, 20)
        m.train()
        assert torch.nn.utils.clip_grad_norm_ is fast_clip      # routed while an engine-backed model is alive (installed by the model, not by the import)
        # (dropout is off in the config these models are built with: the two runs can be compared exactly)
        eng = m.lxrt_encoder.model._binding.engine
        optim = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=20)
        norms = []
        for step in range(3):
            optim.zero_grad()
            logit = m(feats.cuda(), boxes.cuda(), SENTS)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)
            loss.backward()
            clip = torch_clip if which == "torch" else torch.nn.utils.clip_grad_norm_
            norms.append(float(clip(m.parameters(), 0.5)))          # small max_norm: the rescale really happens
            assert (eng._pending_clip == 0.5) == (which == "fast")
            optim.step()
            assert eng._pending_clip is None                        # consumed by the update kernel
        m.eval()
        with torch.no_grad():
            lg_a = m(feats.cuda(), boxes.cuda(), SENTS).clone()
            in_sync = m.lxrt_encoder.model._binding.in_sync()
            m.lxrt_encoder.model._binding.engine.sync_weights()
            lg_b = m(feats.cuda(), boxes.cuda(), SENTS).clone()
        res[which] = (norms, {k: v.detach().clone() for k, v in m.named_parameters()}, lg_a, lg_b, in_sync)
    nf, pf, la, lb, sync_f = res["fast"]
    nt, pt, _, _, _ = res["torch"]
    ni, pi_, _, _, _ = res["inplace"]
    # folding the coefficient into the update = scaling in place: the same f32 arithmetic (g * coef, then the moments): bit for bit in both precisions
    # (round 5 held bf16x3 to rtol 1e-4 / atol 2e-6: two RUNS then differed by themselves - the embedding tables' gradients were scatter-added by float
    # atomics in arrival order, and BertAdam's normalised update amplified the last-bit differences; round 6 sums every table row in a fixed order)
    assert nf == ni and all(torch.equal(pf[k], pi_[k]) for k in pf)
    assert sync_f                                   # the fused optimizer path left the copies current: no re-cast happened at the forward
    assert torch.equal(la, lb)                      # ... and they are exactly what a full re-cast produces
    np.testing.assert_allclose(nf, nt, rtol=2e-5)
    assert min(nf) > 0.5                            # every step was clipped
    worst = max(float((pf[k] - pt[k]).abs().max()) / max(1e-12, float(pt[k].abs().max())) for k in pt)
    print("fused clip vs torch clip (%s): norms %s vs %s, worst relative parameter difference %.2e" % (precision, nf, nt, worst))
    assert worst < 1e-4


def test_deferred_clip_is_seen_by_every_reader_of_grad(env):
    """`clip_grad_norm_` leaves the clip coefficient with the engine (BertAdam.step folds it in), yet whoever READS a `.grad`
    after the clip sees the scaled values, as after torch's in-place clip (the parameters are ArenaParameters: their `.grad` property
    materialises the pending rescale first).  Also: an optimizer that steps only SOME parameters cannot fold (the others' gradients would stay
    unscaled): the rescale is materialised; a second clip measures the scaled gradients; a backward that accumulates onto clipped gradients
    materialises first; a backward that overwrites them drops the pending rescale."""
    import lxrt.entry  # noqa: F401
    from lxrt.optimization import BertAdam
    from rgqa_amd.engine import raw_grad
    from rgqa_amd.lxrt.modeling import ArenaParameter
    feats, boxes, target = batch(20)
    m, _ = build("bf16", 20)
    m.train()
    eng = m.lxrt_encoder.model._binding.engine
    params = list(m.parameters())
    assert all(isinstance(p, ArenaParameter) and isinstance(p, torch.nn.Parameter) for p in params)
    assert set(dict(m.named_parameters())) == set(m.state_dict())

    def fwd_bwd(zero=True):
        if zero:
            m.zero_grad()
        logit = m(feats.cuda(), boxes.cuda(), SENTS)
        (torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)).backward()

    fwd_bwd()
    live = [p for p in params if raw_grad(p) is not None]
    before = [raw_grad(p).clone() for p in live]
    norm = float(torch.nn.utils.clip_grad_norm_(params, 0.5))
    want = float(sum(float((g.double() ** 2).sum()) for g in before) ** 0.5)
    assert abs(norm - want) < 1e-4 * want and norm > 0.5
    assert eng._pending_clip == 0.5 and torch.equal(raw_grad(live[3]), before[3])       # nothing has moved yet
    coef = 0.5 / (norm + 1e-6)
    seen = live[3].grad                                                                    # a reader: the rescale happens now, for every gradient
    assert eng._pending_clip is None
    for p, g0 in zip(live, before):
        assert torch.allclose(raw_grad(p), g0 * coef, rtol=1e-5, atol=1e-12)
    assert seen is raw_grad(live[3])
    # a second clip measures the scaled gradients (norm = 0.5), deferred again; an optimizer over HALF of the parameters cannot fold it
    n2 = float(torch.nn.utils.clip_grad_norm_(params, 0.25))
    assert abs(n2 - 0.5) < 1e-3 and eng._pending_clip == 0.25
    half = BertAdam(live[: len(live) // 2], lr=1e-3, warmup=0.1, t_total=20)
    half.step()
    assert eng._pending_clip is None
    c2 = 0.25 / (n2 + 1e-6)
    for p, g0 in zip(live, before):
        assert torch.allclose(raw_grad(p), g0 * (coef * c2), rtol=2e-5, atol=1e-12)          # every gradient scaled in place, stepped or not
    # accumulation onto clipped gradients: materialise, then add
    fwd_bwd()
    g1 = [raw_grad(p).clone() for p in live]
    n3 = float(torch.nn.utils.clip_grad_norm_(params, 0.5))
    assert eng._pending_clip == 0.5
    fwd_bwd(zero=False)                                                                    # same batch, no dropout in this config: adds g1 again
    assert eng._pending_clip is None
    c3 = 0.5 / (n3 + 1e-6)
    for p, g in zip(live, g1):
        assert torch.allclose(raw_grad(p), g * c3 + g, rtol=2e-2, atol=1e-6)               # (bf16 backward: the second pass rounds on its own)
    # a backward that overwrites the gradients drops a pending rescale
    torch.nn.utils.clip_grad_norm_(params, 0.5)
    assert eng._pending_clip == 0.5
    fwd_bwd()
    assert eng._pending_clip is None
    for p, g in zip(live, g1):
        assert torch.allclose(raw_grad(p), g, rtol=2e-2, atol=1e-6)


def test_clip_patch_leaves_foreign_models_to_torch(env):
    """while a drop-in model is alive `torch.nn.utils.clip_grad_norm_` is routed through lxrt.optimization.clip_grad_norm_.
    The contract: a parameter set that is not exactly one engine's arena views goes to torch's own implementation untouched - a
    foreign model in the same process (CPU or GPU), a mix of an rgqa model's and foreign parameters, a generator argument, norm_type != 2,
    error_if_nonfinite - and gives torch's results."""
    import gc
    import lxrt.entry  # noqa: F401
    import rgqa_amd.lxrt.optimization as O      # (the implementation module: the drop-in `lxrt.optimization` re-exports its names at import time)
    from lxrt.optimization import clip_grad_norm_ as fast_clip, _torch_clip_grad_norm_ as torch_clip
    # the routing is a scope the drop-in MODEL installs and its finaliser removes - importing lxrt.entry alone rebinds nothing
    gc.collect()
    owners0 = O._CLIP_OWNERS
    if owners0 == 0:
        assert torch.nn.utils.clip_grad_norm_ is torch_clip
    scope, _ = build("bf16", 20)
    assert O._CLIP_OWNERS == owners0 + 1 and torch.nn.utils.clip_grad_norm_ is fast_clip
    del scope
    gc.collect()
    assert O._CLIP_OWNERS == owners0
    if owners0 == 0:
        assert torch.nn.utils.clip_grad_norm_ is torch_clip      # the last engine-backed model is gone: torch's own function is back
    keep, _ = build("bf16", 20)                                   # alive for the rest of the test: the routed name is what is exercised below
    assert torch.nn.utils.clip_grad_norm_ is fast_clip
    for dev_ in ("cpu", "cuda"):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3)).to(dev_)
        net(torch.randn(5, 8, device=dev_)).pow(2).sum().backward()
        ref = [p.grad.clone() for p in net.parameters()]
        want = float(torch_clip(net.parameters(), 1e9))                                                          # torch's norm (no scaling at 1e9)
        got = float(torch.nn.utils.clip_grad_norm_(net.parameters(), 0.1))                                        # a generator, as the trainers pass
        assert abs(got - want) < 1e-6 * max(1.0, want)
        coef = min(1.0, 0.1 / (got + 1e-6))
        for p, r in zip(net.parameters(), ref):
            assert torch.allclose(p.grad, r * coef, rtol=1e-6, atol=1e-12)
        assert float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0, norm_type=1.0)) > 0               # other norms: torch's code
    # an rgqa model's parameters TOGETHER with a foreign module's: not one arena -> torch's implementation over all of them, in place
    feats, boxes, target = batch(20)
    m, _ = build("bf16", 20)
    m.train()
    eng = m.lxrt_encoder.model._binding.engine
    extra = torch.nn.Linear(4, 4).cuda()
    extra(torch.randn(2, 4, device="cuda")).sum().backward()
    logit = m(feats.cuda(), boxes.cuda(), SENTS)
    (torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)).backward()
    allp = list(m.parameters()) + list(extra.parameters())
    ref = [None if p.grad is None else p.grad.detach().clone() for p in allp]
    want = float(sum(float((r.double() ** 2).sum()) for r in ref if r is not None) ** 0.5)
    got = float(torch.nn.utils.clip_grad_norm_(allp, 0.5))
    assert abs(got - want) < 1e-4 * want and eng._pending_clip is None              # nothing deferred: torch scaled every tensor in place
    coef = 0.5 / (got + 1e-6)
    for p, r in zip(allp, ref):
        if r is not None:
            assert torch.allclose(p.grad, r * coef, rtol=1e-4, atol=1e-10)


def test_clip_fast_path_rejects_a_foreign_gradient_tensor(env):
    """after one recognised call the patched clip_grad_norm_ re-validates EVERY parameter's .grad, not only the first: a gradient
    tensor that was re-bound afterwards (here: a scaled clone in place of the arena view) sends the call to torch's implementation, which
    clips the tensors it was given - the foreign one included - instead of rescaling the engine's arena behind their back."""
    import lxrt.entry  # noqa: F401
    from lxrt.optimization import clip_grad_norm_ as fast_clip
    feats, boxes, target = batch(20)
    m, _ = build("bf16", 20)
    m.train()
    params = list(m.parameters())

    def fwd_bwd():
        m.zero_grad()
        logit = m(feats.cuda(), boxes.cuda(), SENTS)
        (torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)).backward()

    fwd_bwd()
    n0 = float(fast_clip(params, 1e9))                           # recognised: the engine's fast path, now cached
    fwd_bwd()
    victim = [p for p in params if p.grad is not None and p.grad.numel() > 1000][-1]
    victim.grad = victim.grad.clone() * 3.0                       # a foreign tensor in place of the arena view
    ref = [None if p.grad is None else p.grad.detach().clone() for p in params]
    want = float(sum(float((r.double() ** 2).sum()) for r in ref if r is not None) ** 0.5)
    got = float(fast_clip(params, 0.25))
    assert abs(got - want) <= 1e-4 * want and got > n0 * (1 + 1e-4)          # the norm counts the foreign (x 3) tensor
    coef = 0.25 / (got + 1e-6)
    assert torch.allclose(victim.grad, ref[next(i for i, p in enumerate(params) if p is victim)] * coef, rtol=1e-4, atol=1e-9)      # ... and it was clipped in place


def test_train_mode_runs_and_is_seeded(env):
    """model.train(): dropout 0.1 active (reference BertConfig defaults); different forward calls draw different masks."""
    import rgqa_amd.lxrt.modeling as M
    from tasks.gqa_model import GQAModel
    os.environ["RGQA_PRECISION"] = "bf16"
    args = types.SimpleNamespace(llayers=2, xlayers=2, rlayers=1, from_scratch=True)
    M.LXRTFeatureExtraction.from_pretrained = classmethod(
        lambda cls, name, **kw: cls(M.BertConfig(CFG["vocab_size"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"],
                                                 intermediate_size=CFG["inter"], max_position_embeddings=CFG["max_pos"]), **kw))
    m = GQAModel(CFG["num_answers"], max_seq_length=20, model_args=args).cuda()
    feats, boxes, target = batch(20)
    m.train()
    a = m(feats.cuda(), boxes.cuda(), SENTS)
    b = m(feats.cuda(), boxes.cuda(), SENTS)
    assert not torch.equal(a, b)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(b, target.cuda()) * b.size(1)
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    with pytest.raises(RuntimeError, match="newer forward"):
        a.sum().backward()
    m.eval()
    c = m(feats.cuda(), boxes.cuda(), SENTS)
    d = m(feats.cuda(), boxes.cuda(), SENTS)
    assert torch.equal(c, d)


def test_odin_style_input_gradients(env, golden_dir):
    """What tasks/gqa_odin.py:97-121 does: feats / boxes with requires_grad, a loss on the logits, backward, then the SIGN of
    feats.grad / boxes.grad perturbs the inputs.  Checked against the oracle's autograd (f32)."""
    from oracle import lxmert_ref as R
    T = 20
    m, filled = build("f32", T)
    feats, boxes, target = batch(T)
    vocab = {w.rstrip("\n"): i for i, w in enumerate(open(os.path.join(golden_dir, "g4_vocab.txt"), encoding="utf-8"))}
    ids, mask, _ = R.sents_to_features(SENTS, T, vocab)
    ids, mask = torch.tensor(ids), torch.tensor(mask)
    m.eval()
    f = feats.cuda().requires_grad_(True)
    bx = boxes.cuda().requires_grad_(True)
    logit = m(f, bx, SENTS)
    labels = logit.detach().argmax(1)
    loss = torch.nn.functional.cross_entropy(logit / 1000.0, labels)       # temperature-scaled CE on the predicted label (:105-112)
    loss.backward()
    assert f.grad is not None and bx.grad is not None and f.grad.shape == f.shape and bx.grad.shape == bx.shape
    cfg = R.RefConfig(**CFG)
    P = {k: torch.from_numpy(v.copy()) for k, v in filled.items()}
    fr, br = feats.clone().requires_grad_(True), boxes.clone().requires_grad_(True)
    lg, _ = R.gqa_forward(P, cfg, fr, br, ids, mask)
    torch.nn.functional.cross_entropy(lg / 1000.0, lg.detach().argmax(1)).backward()
    np.testing.assert_allclose(f.grad.cpu().numpy(), fr.grad.numpy(), rtol=0, atol=2e-3 * float(fr.grad.abs().max()))
    np.testing.assert_allclose(bx.grad.cpu().numpy(), br.grad.numpy(), rtol=0, atol=2e-3 * float(br.grad.abs().max()))
    big = fr.grad.abs() > 0.05 * fr.grad.abs().max()
    assert torch.equal(torch.ge(f.grad.cpu(), 0)[big], torch.ge(fr.grad, 0)[big])      # the sign ODIN uses, away from zero


def test_lxrt_vis_output_attention(env):
    """The reference's visualisation variant (lxrt_vis/entry.py:109-121): forward(sents, feats, output_attention=True) returns
    (pooled, (l2v_atts, v2l_atts), input_ids); probabilities checked against the oracle's restatement of
    lxrt_vis/modeling.py:320-350 on the real tokens, with and without packed language rows."""
    from lxrt_vis.entry import LXRTEncoder
    import rgqa_amd.lxrt_vis.modeling as MV
    from rgqa_amd import synth
    from oracle import lxmert_ref as R
    os.environ["RGQA_PRECISION"] = "f32"
    args = types.SimpleNamespace(llayers=CFG["l_layers"], xlayers=CFG["x_layers"], rlayers=CFG["r_layers"], from_scratch=False)
    assert issubclass(LXRTEncoder.MODEL_CLASS, MV.LXRTFeatureExtraction)
    enc = LXRTEncoder(args, max_seq_length=20)
    filled = synth.fill_state_dict({"lxrt_encoder.model." + k: tuple(v.shape) for k, v in enc.model.state_dict().items()})
    enc.model.load_state_dict({k[len("lxrt_encoder.model."):]: torch.from_numpy(v) for k, v in filled.items()})
    enc = enc.cuda().eval()
    feats, boxes, _ = batch(20)
    cfg = R.RefConfig(**CFG)
    full = dict(filled)
    for k, shp in R.param_shapes(cfg).items():
        if k.startswith("logit_fc."):
            full[k] = np.zeros(shp, dtype=np.float32)
    P = {k: torch.from_numpy(v.copy()) for k, v in full.items()}
    ids, mask, _ = R.sents_to_features(SENTS, 20, enc.tokenizer.vocab)
    trace = {}
    with torch.no_grad():
        _, _, pooled = R.encoder_forward(P, cfg, torch.tensor(ids), torch.zeros(len(SENTS), 20, dtype=torch.long), torch.tensor(mask), feats, boxes, trace)
    lens = [int(sum(m)) for m in mask]
    for varlen in ("1", "0"):
        os.environ["RGQA_VARLEN"] = varlen
        try:
            with torch.no_grad():
                out, (l2v, v2l), input_ids = enc(SENTS, (feats.cuda(), boxes.cuda()), output_attention=True)
                out0, (n1, n2), _ = enc(SENTS, (feats.cuda(), boxes.cuda()))
        finally:
            os.environ.pop("RGQA_VARLEN")
        assert n1 == [None] * CFG["x_layers"] and n2 == [None] * CFG["x_layers"] and torch.equal(out0, out)
        assert input_ids.tolist() == [list(r) for r in ids]
        np.testing.assert_allclose(out.cpu().numpy(), pooled.numpy(), rtol=0, atol=2e-5)
        assert len(l2v) == len(v2l) == CFG["x_layers"]
        for i in range(CFG["x_layers"]):
            a, c = l2v[i].cpu().numpy(), v2l[i].cpu().numpy()
            ra, rc = trace["x%d_l2v" % i].numpy(), trace["x%d_v2l" % i].numpy()
            assert a.shape == ra.shape and c.shape == rc.shape
            for b_, n in enumerate(lens):
                np.testing.assert_allclose(a[b_, :, :n], ra[b_, :, :n], rtol=0, atol=2e-5)
                np.testing.assert_allclose(c[b_], rc[b_], rtol=0, atol=2e-5)


def _patch_small_config(golden_dir):
    """what the `env` fixture does, for a spawned rank process"""
    os.environ["RGQA_BERT_VOCAB"] = os.path.join(golden_dir, "g4_vocab.txt")
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    sys.path.insert(0, ROOT)
    import rgqa_amd.lxrt.modeling as M
    M.VISUAL_CONFIG.visual_feat_dim = CFG["feat_dim"]
    M.LXRTFeatureExtraction.from_pretrained = classmethod(
        lambda cls, name, **kw: cls(M.BertConfig(CFG["vocab_size"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"],
                                                 intermediate_size=CFG["inter"], max_position_embeddings=CFG["max_pos"],
                                                 hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), **kw))


def _trainer_loop(m, feats, boxes, sents, target, steps):
    """tasks/gqa_conf.py:174-202, unchanged"""
    from lxrt.optimization import BertAdam
    optim = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=20)
    bce = torch.nn.BCEWithLogitsLoss()
    m.train()       # (dropout is 0 in the config these tests build: train mode is deterministic; under torch.distributed the exchange runs in train mode only)
    for _ in range(steps):
        optim.zero_grad()
        logit = m(feats.cuda(), boxes.cuda(), sents)
        loss = bce(logit, target.cuda()) * logit.size(1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.)
        optim.step()
    return {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def _dropin_dp_worker(rank, world, port, golden_dir, q, mode="allreduce", precision="f32"):
    import torch.distributed as dist
    _patch_small_config(golden_dir)
    os.environ["RGQA_DP_MODE"] = mode
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    m, _ = build(precision, 20)
    m.lxrt_encoder.multi_gpu()
    feats, boxes, target = batch(20)
    n = len(SENTS) // world
    sl = slice(rank * n, (rank + 1) * n)
    sd = _trainer_loop(m, feats[sl], boxes[sl], SENTS[sl], target[sl], 2)
    torch.cuda.synchronize()
    q.put((rank, sd))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,precision", [("allreduce", "f32"), ("sharded", "f32"), ("sharded", "bf16"), ("sharded", "bf16x3_fwd")])
def test_unchanged_trainer_is_data_parallel_under_torch_distributed(env, golden_dir, mode, precision):
    """the reference's train loop, untouched, launched as two processes (gloo here, RCCL with one GPU per rank): two ranks on
    half the batch each == one rank on the whole batch.  `allreduce`: gradients are averaged inside backward(), every rank steps every parameter.
    `sharded` (RGQA_DP_MODE=sharded, round 6): backward() reduce-scatters, `clip_grad_norm_` returns the global norm from the owners' shares,
    this package's `BertAdam.step()` updates the 1/N of the arena the rank owns and gathers the weights (bf16 / bf16x3_fwd engines: beside the next
    forward pass), `state_dict()` gathers the f32 masters."""
    import torch.multiprocessing as mp
    m, _ = build(precision, 20)
    feats, boxes, target = batch(20)
    ref = _trainer_loop(m, feats, boxes, SENTS, target, 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 500) + 3 * (["allreduce", "sharded"].index(mode) * 4 + ["f32", "bf16", "bf16x3_fwd"].index(precision))
    procs = [ctx.Process(target=_dropin_dp_worker, args=(r, 2, port, golden_dir, q, mode, precision), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r, sd = q.get(timeout=300)
            res[r] = sd
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    worst, tot, cnt = 0.0, 0.0, 0
    for k in ref:
        assert np.array_equal(res[0][k], res[1][k]), k              # replicas identical (sharded: after state_dict()'s gather)
        d = np.abs(res[0][k] - ref[k])
        worst, tot, cnt = max(worst, float(d.max())), tot + float(d.sum()), cnt + d.size
        if precision == "f32":
            # BCEWithLogitsLoss takes the mean over the LOCAL batch: the average of the two half-batch means == the full-batch mean
            np.testing.assert_allclose(res[0][k], ref[k], rtol=2e-4, atol=2e-5, err_msg=k)
    print("drop-in DP %s / %s, 2 ranks vs 1: |param diff| max %.3e mean %.3e" % (mode, precision, worst, tot / cnt))
    if precision != "f32":
        # bf16 operands / a bf16 payload round differently under the other batch split; BertAdam moves an element whose tiny gradient changes sign by up to
        # 3.2 lr per step whatever its size (tests/test_gpu_dp.py): the MEAN is the meaningful bound
        assert worst < 1.3e-2 and tot / cnt < 2e-5, (worst, tot / cnt)


@pytest.mark.parametrize("precision", ["f32", "bf16", "bf16x3"])
def test_parameter_changes_after_a_forward_reach_the_operand_copies(env, precision, tmp_path):
    """What a trainer does to a model that has already run (gqa_conf.py:96-110, entry.py:126-152; foreign optimizers): in-place updates of
    the device parameters, `load_state_dict` on the device, `LXRTEncoder.save` / `.load`.  Every one must reach the engine's low-precision /
    split-f32 / transposed operand copies before the next forward."""
    m, _ = build(precision)
    m.eval()
    feats, boxes, _ = batch()
    f, b = feats.cuda(), boxes.cuda()
    with torch.no_grad():
        lg1 = m(f, b, SENTS).float().cpu()
        for p in m.parameters():
            p.mul_(1.05)                       # in place, after the engine has made its copies
        lg2 = m(f, b, SENTS).float().cpu()
        assert float((lg2 - lg1).abs().max()) > 1e-3
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        m2, _ = build(precision)
        m2.eval()
        m2(f, b, SENTS)
        m2.load_state_dict(sd)                 # checkpoint load on the device, after a forward
        assert torch.equal(m2(f, b, SENTS).float().cpu(), lg2)
        m.lxrt_encoder.save(str(tmp_path / "ck"))
        m3, _ = build(precision)
        m3.eval()
        m3(f, b, SENTS)
        m3.lxrt_encoder.load(str(tmp_path / "ck"))
        m3.logit_fc.load_state_dict(m.logit_fc.state_dict())
        assert torch.equal(m3(f, b, SENTS).float().cpu(), lg2)
