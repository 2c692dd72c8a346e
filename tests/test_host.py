"""CPU-only checks: host logic of the drop-in layer, the C-ABI surface, argument validation, and the N>1 gradient
exchange over gloo (world_size 2).  No compute call is made on the HIP library here."""
import json
import os
import re
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rgqa_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "rgqa.h")).read()
    declared = set(re.findall(r"\b(rgqa_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), "include/rgqa.h declares %s but librgqa_hip.so does not export it" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert lib.rgqa_version() >= 100


def test_ctypes_signatures_match_header_arity():
    """Every SIGNATURES entry has exactly as many argtypes as the prototype in include/rgqa.h has parameters (a short list
    still 'works' on x86-64 by accident of the calling convention), and pointer / integer / float classes agree."""
    import ctypes as C
    from rgqa_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rgqa.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = dict(re.findall(r"\b(rgqa_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S))
    assert len(protos) >= 25
    for name, args in _lib.SIGNATURES.items():
        assert name in protos, "%s has a ctypes signature but no prototype in include/rgqa.h" % name
        params = [a.strip() for a in protos[name].replace("\n", " ").split(",")]
        if params == ["void"] or params == [""]:
            params = []
        assert len(params) == len(args), "%s: header has %d parameters, ctypes signature %d" % (name, len(params), len(args))
        for prm, ct in zip(params, args):
            is_ptr = "*" in prm or "[" in prm
            ct_ptr = ct in (C.c_void_p, C.c_char_p) or hasattr(ct, "contents") or getattr(ct, "_type_", None) is not None and not isinstance(ct._type_, str)
            assert is_ptr == bool(ct_ptr), "%s: parameter '%s' vs ctypes %s" % (name, prm, ct)
            if not is_ptr:
                want_float = re.search(r"\b(float|double)\b", prm) is not None
                assert want_float == (ct in (C.c_float, C.c_double)), "%s: parameter '%s' vs ctypes %s" % (name, prm, ct)


def test_engine_layout_matches_reference_state_dict_contract():
    """Engine parameter table == the reference's GQAModel state_dict keys/shapes, full 9/5/5 config;
    fused-QKV contiguity and 64-element alignment hold; the dead range is exactly x_layers.4.visn_*."""
    from oracle import lxmert_ref as R
    from rgqa_amd.engine import Engine
    e = Engine(precision="bf16")
    ref = R.param_shapes(R.RefConfig())
    got = {sp.name: sp.shape for sp in e.specs}
    assert set(got) == set(ref)
    assert all(tuple(ref[k]) == tuple(got[k]) for k in ref)
    assert len(got) == 449 + 6
    by = {sp.name: sp for sp in e.specs}
    for sp in e.specs:
        assert sp.offset % 64 == 0
        if sp.name.endswith("query.weight"):
            k, v = by[sp.name.replace("query", "key")], by[sp.name.replace("query", "value")]
            assert k.offset == sp.offset + sp.numel and v.offset == k.offset + k.numel
    dead = [sp for sp in e.specs if sp.dead]
    assert len(dead) == 16 and all(".x_layers.4.visn_" in sp.name for sp in dead)
    assert sum(sp.numel for sp in dead) == 7087872
    b, en = e.dead_range
    assert all(b <= sp.offset and sp.offset + sp.numel <= en for sp in dead)
    assert sum(sp.numel for sp in e.specs if not sp.dead) == 204864306     # parameters with gradients


def test_engine_create_rejects_bad_configs():
    from rgqa_amd.engine import Engine
    with pytest.raises(RuntimeError, match="not a multiple of the number of attention"):
        Engine(hidden=768, heads=7)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        Engine(hidden=96, heads=2)
    with pytest.raises(RuntimeError):
        Engine(hidden_dropout=1.5)
    e = Engine()
    with pytest.raises(RuntimeError, match="no CPU path"):
        e.allocate("cpu")


def test_tokenizer_and_features_match_reference_vectors(golden_dir):
    from rgqa_amd.lxrt.tokenization import BertTokenizer
    from rgqa_amd.lxrt.entry import convert_sents_to_features
    g = json.load(open(os.path.join(golden_dir, "g4_tokenizer.json"), encoding="utf-8"))
    tok = BertTokenizer(os.path.join(golden_dir, "g4_vocab.txt"), do_lower_case=True)
    for T in (20, 30):
        for _ in range(2):   # second pass is served by the per-string cache
            feats = convert_sents_to_features(g["sentences"], T, tok)
            assert [f.input_ids for f in feats] == g["T%d" % T]["input_ids"]
            assert [f.input_mask for f in feats] == g["T%d" % T]["input_mask"]
            assert [f.segment_ids for f in feats] == g["T%d" % T]["segment_ids"]
    assert tok.tokenize("unaffable") == ["un", "##aff", "##able"]
    assert tok.convert_ids_to_tokens(tok.convert_tokens_to_ids(["[CLS]", "dog", "[SEP]"])) == ["[CLS]", "dog", "[SEP]"]
    with pytest.raises(ValueError):
        BertTokenizer("/nonexistent/vocab.txt")
    assert BertTokenizer.from_pretrained("/nonexistent-dir-xyz") is None      # reference returns None on a failed fetch


def test_native_tokenizer_matches_reference_vectors_and_python_rules(golden_dir):
    """rgqa_tokenizer_encode (csrc/tokenizer.hip) against (a) the reference-generated G4 ids / masks for every sentence it
    accepts, (b) the Python implementation of the same rules (itself pinned by G4) on generated ASCII text that exercises
    control bytes, runs of whitespace, punctuation, the never-split specials, over-long words, truncation and [UNK]s."""
    import random
    from rgqa_amd.lxrt.tokenization import BertTokenizer, NativeBatchEncoder
    from rgqa_amd.lxrt.entry import convert_sents_to_features
    g = json.load(open(os.path.join(golden_dir, "g4_tokenizer.json"), encoding="utf-8"))
    tok = BertTokenizer(os.path.join(golden_dir, "g4_vocab.txt"), do_lower_case=True)
    enc = NativeBatchEncoder(tok)
    sents = g["sentences"]
    for T in (20, 30):
        ids = np.zeros((len(sents), T), dtype=np.int64)
        mask = np.zeros_like(ids)
        lens, needs = enc.encode(sents, T, ids, mask)
        assert 0 < int(needs.sum()) < len(sents) // 2          # the accented sentences go back to Python, the rest is native
        for i in range(len(sents)):
            if needs[i]:
                assert not sents[i].isascii()
                continue
            assert ids[i].tolist() == g["T%d" % T]["input_ids"][i], sents[i]
            assert mask[i].tolist() == g["T%d" % T]["input_mask"][i]
            assert lens[i] == sum(g["T%d" % T]["input_mask"][i])
    words = [w for w in tok.vocab if w.isascii() and not w.startswith("##") and not w.startswith("[")]
    rnd = random.Random(7)
    pool = words + ["[SEP]", "[CLS]", "[UNK]", "[MASK]", "[PAD]", "qzxv", "Is", "THE", "x" * 101, "un" + "able" * 30, "it's", "left-most", "(red)",
                    "a\x01b", "tab\there", "what?!", "...", "[sep]", "#", "##able", "3.5", "re_do", "end."]
    seps = [" ", "  ", "\t", "\n", " \r\n ", "\x0b", "\x1f", ""]
    gen = []
    for _ in range(1500):
        k = rnd.randint(0, 26)
        txt = rnd.choice(["", " ", "\n"]) + "".join(rnd.choice(pool) + rnd.choice(seps) for _ in range(k))
        if rnd.random() < 0.3:
            txt = txt.upper() if rnd.random() < 0.5 else txt.title()
        gen.append(txt)
    gen += ["", " ", "\x01", "?", "a" * 100, "a" * 101 + " dog"]
    T = 20
    ids = np.zeros((len(gen), T), dtype=np.int64)
    mask = np.zeros_like(ids)
    lens, needs = enc.encode(gen, T, ids, mask)
    assert int(needs.sum()) == 0
    feats = convert_sents_to_features(gen, T, tok)
    for i, f in enumerate(feats):
        assert ids[i].tolist() == f.input_ids, repr(gen[i])
        assert mask[i].tolist() == f.input_mask, repr(gen[i])
        assert lens[i] == sum(f.input_mask)
    # a sentence with an embedded NUL cannot cross a C string: handed back
    l2, n2 = enc.encode(["a\x00b", "caf\u00e9"], T, np.zeros((2, T), dtype=np.int64), np.zeros((2, T), dtype=np.int64))
    assert n2.tolist() == [1, 1]
    with pytest.raises(RuntimeError):
        tok_bad = BertTokenizer(os.path.join(golden_dir, "g4_vocab.txt"))
        tok_bad.vocab_file = "/nonexistent/vocab.txt"
        NativeBatchEncoder(tok_bad)


def test_bertadam_validation_and_schedules():
    from rgqa_amd.lxrt.optimization import BertAdam, warmup_linear, warmup_constant, warmup_cosine
    p = [torch.nn.Parameter(torch.zeros(3))]
    for kw in (dict(lr=-1.0), dict(lr=1e-3, schedule="nope"), dict(lr=1e-3, warmup=1.5), dict(lr=1e-3, b1=1.0), dict(lr=1e-3, b2=-0.1), dict(lr=1e-3, e=-1.0)):
        with pytest.raises(ValueError):
            BertAdam(p, **kw)
    assert warmup_linear(0.05, 0.1) == pytest.approx(0.5)
    assert warmup_linear(0.1, 0.1) == pytest.approx(1.0)
    assert warmup_linear(0.55, 0.1) == pytest.approx(0.5)
    assert warmup_linear(1.2, 0.1) == 0
    assert warmup_constant(0.5, 0.1) == 1.0 and warmup_cosine(0.05, 0.1) == pytest.approx(0.5)
    opt = BertAdam(p, lr=1e-3, warmup=0.1, t_total=10)
    assert opt.get_lr() == [0]
    p[0].grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        opt.step()


def test_dropin_model_surface(golden_dir, monkeypatch):
    """tasks.gqa_model.GQAModel / lxrt.entry.LXRTEncoder import under the reference's module names, carry the reference's
    state_dict keys, and refuse to compute without a GPU."""
    monkeypatch.setenv("RGQA_BERT_VOCAB", os.path.join(golden_dir, "g4_vocab.txt"))
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        from tasks.gqa_model import GQAModel, GQAModel_maha, MAX_GQA_LENGTH
        from lxrt.entry import LXRTEncoder, convert_sents_to_features  # noqa: F401
        from lxrt.modeling import BertLayerNorm, GeLU, VISUAL_CONFIG, BertConfig  # noqa: F401
        from lxrt.optimization import BertAdam  # noqa: F401
        import rgqa_amd.lxrt.modeling as M
    finally:
        sys.path.pop(0)
    assert MAX_GQA_LENGTH == 30
    monkeypatch.setattr(M.VISUAL_CONFIG, "visual_feat_dim", 64)
    monkeypatch.setattr(M.LXRTFeatureExtraction, "from_pretrained", classmethod(
        lambda cls, name, **kw: cls(M.BertConfig(80, hidden_size=128, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64), **kw)))
    args = types.SimpleNamespace(llayers=2, xlayers=2, rlayers=1, from_scratch=True)
    m = GQAModel(17, model_args=args)
    from oracle import lxmert_ref as R
    ref = R.param_shapes(R.RefConfig(vocab_size=80, hidden=128, heads=2, inter=256, max_pos=64, l_layers=2, x_layers=2, r_layers=1, feat_dim=64, num_answers=17))
    sd = m.state_dict()
    assert set(sd) == set(ref) and all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert isinstance(m.logit_fc[2], BertLayerNorm) and m.logit_fc[2].eps == 1e-12 and isinstance(m.logit_fc[1], GeLU)
    assert m.lxrt_encoder.dim == 128 and m.lxrt_encoder.max_seq_length == 30
    # init_bert_weights semantics: LN = 1/0, Linear bias = 0, weights ~ N(0, 0.02)
    w = sd["lxrt_encoder.model.bert.encoder.layer.0.attention.self.query.weight"]
    assert abs(float(w.std()) - 0.02) < 2e-3 and float(sd["logit_fc.2.weight"].min()) == 1.0 and float(sd["logit_fc.0.bias"].abs().max()) == 0.0
    # state_dict round trip + `module.` prefix stripping of LXRTEncoder.load (entry.py:126-152)
    enc_sd = {"module." + k: v.clone() + 1 for k, v in m.lxrt_encoder.model.state_dict().items()}
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rgqa_host_test")
    torch.save(enc_sd, path + "_LXRT.pth")
    m.lxrt_encoder.load(path)
    assert torch.allclose(m.state_dict()["lxrt_encoder.model.bert.pooler.dense.bias"], enc_sd["module.bert.pooler.dense.bias"])
    os.remove(path + "_LXRT.pth")
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(2, 36, 64), torch.zeros(2, 36, 4), ["what is this", "is it red?"])
    with pytest.raises(NotImplementedError):
        M.LXRTFeatureExtraction(M.BertConfig(80, hidden_size=128, num_attention_heads=2, intermediate_size=256), mode="lxr")
    with pytest.raises(ValueError):
        M.BertConfig(3.5)


class _TorchOps:
    """the exchange's two local kernels (rgqa_cast_bf16, rgqa_sum_bf16_parts) restated with torch for the CPU rehearsal"""

    def cast_bf16(self, dst, src):
        dst.copy_(src)

    def sum_parts(self, dst, parts, stride, nparts, sq_ws=None, sq_out=None):
        acc = torch.zeros_like(dst)
        for r in range(nparts):
            acc += parts[r * stride:r * stride + dst.numel()].float()
        dst.copy_(acc)
        if sq_out is not None:
            sq_out += (acc.double() ** 2).sum().float()


def _fake_engine(n, rank, precision, mod=64):
    base = (torch.arange(n) % mod).float()
    return types.SimpleNamespace(grads=base * (rank + 1), params=torch.ones(n), params_lp=torch.ones(n, dtype=torch.bfloat16), adam_m=None, adam_v=None,
                                 precision=precision, live_ranges=lambda: [(0, 296), (360, n)], lib=None, h=None, _sharded_owner=None)


def _dp_worker(rank, world, port, q, mod=64):
    import torch.distributed as dist
    from rgqa_amd.parallel import GradAllReduce, ShardedExchange
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    n = 1000
    out = {}
    if world > 2:       # bench.py's pre-flight check of every collective the exchanges need, at this world size (all-to-all and in-place all-gather included)
        import bench
        for mode, prec in (("sharded", "f32"), ("allreduce", "f32")):
            bench.dp_selfcheck(dist, mode, torch.device("cpu"), prec)
    for name, bf16 in (("allreduce", False), ("allreduce_bf16", True)):
        eng = _fake_engine(n, rank, "f32", mod)
        GradAllReduce(eng, dist, bucket_mb=1, bf16=bf16, ops=_TorchOps()).all_reduce()
        out[name] = eng.grads.numpy().copy()

    class CpuSharded(ShardedExchange):      # the two HIP calls of step() restated with torch; everything else is the product code
        def _local_sumsq(self, lo, hi, s):
            self._sumsq += (self.e.grads[lo:hi] ** 2).sum()

        def _local_adam(self, lo, hi, lr_t, b1, b2, eps, wd, clip, max_norm, prescale, s):
            e = self.e
            e.params[lo:hi] -= lr_t * prescale * e.grads[lo:hi]
            e.params_lp[lo:hi] = e.params[lo:hi].bfloat16()

        def _after_weights(self, s):
            pass

    for prec in ("bf16", "f32"):
        eng = _fake_engine(n, rank, prec, mod)
        local = eng.grads.clone()
        ex = CpuSharded(eng, dist, ops=_TorchOps())
        ex.chunks = __import__("rgqa_amd.parallel", fromlist=["shard_layout"]).shard_layout(eng.live_ranges(), world, 256)    # several chunks, two of them ragged
        ex.smax = max(c[2] for c in ex.chunks)
        ex.events = [-1] * len(ex.chunks)
        ex.exchange()
        mine = [__import__("rgqa_amd.parallel", fromlist=["owned"]).owned(c, rank) for c in ex.chunks]
        ex.step(0.5)
        assert eng._sharded_owner is ex          # Engine.adam_step refuses to run on a sharded optimizer state
        ex.gather_master()
        out["sharded_" + prec] = dict(grads=eng.grads.numpy().copy(), local=local.numpy().copy(), mine=mine, params=eng.params.numpy().copy(),
                                      params_lp=eng.params_lp.float().numpy().copy(), sumsq=float(ex._sumsq), chunks=ex.chunks)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_exchange_modes_gloo(world):
    """DP exchange on CPU/gloo at world size 2 and at the size the driver's node has, 8 (the shard arithmetic had
    only ever run with two ranks): all-reduce (f32 and bf16 payload) sums the live ranges and leaves the dead range alone; the sharded mode
    leaves each rank the SUM over its own 1/N of every chunk - ragged chunks, parts aligned to 8 elements and, at world 8, owners whose part of
    a ragged chunk is EMPTY included - takes the global norm from one scalar all-reduce, and after step() every rank holds the same updated
    weights.  At world 8 the workers also run bench.py's collective self-check (all-to-all + in-place all-gather) first."""
    import torch.multiprocessing as mp
    from rgqa_amd.parallel import bucket_ranges, shard_layout, owned
    assert bucket_ranges([(0, 10), (20, 25)], 4) == [(0, 4), (4, 8), (8, 10), (20, 24), (24, 25)]
    assert shard_layout([(0, 100)], 4, 64) == [(0, 64, 16), (64, 100, 16)]
    assert [owned((64, 100, 16), r) for r in range(4)] == [(64, 80), (80, 96), (96, 100), (100, 100)]
    # the layout the workers use at world 8: the chunk (256, 296) is cut into parts of 8 (5 rounded up to the alignment): ranks 5..7 own nothing of it
    lay8 = shard_layout([(0, 296), (360, 1000)], 8, 256)
    assert lay8 == [(0, 256, 32), (256, 296, 8), (360, 616, 32), (616, 872, 32), (872, 1000, 16)]
    assert [owned(lay8[1], r) for r in range(8)] == [(256, 264), (264, 272), (272, 280), (280, 288), (288, 296), (296, 296), (296, 296), (296, 296)]
    mod = 64 if world == 2 else 8          # integers whose sums over the ranks (x 36 at world 8) stay exact in bf16: every check below is an equality
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, mod), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=240) for _ in range(world))
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    n = 1000
    tot = world * (world + 1) // 2          # sum over ranks of (rank + 1)
    base = (torch.arange(n) % mod).float()
    live = torch.zeros(n, dtype=torch.bool)
    live[:296] = True
    live[360:] = True
    for r in range(world):
        for mode in ("allreduce", "allreduce_bf16"):
            g = torch.from_numpy(res[r][mode])
            assert torch.equal(g[live], base[live] * tot), mode
            assert torch.equal(g[~live], base[~live] * (r + 1)), mode
        for prec in ("bf16", "f32"):
            o = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in res[r]["sharded_" + prec].items()}
            assert len(o["chunks"]) == 5 and any((b - a) != world * s for a, b, s in o["chunks"])
            own = torch.zeros(n, dtype=torch.bool)
            for lo, hi in o["mine"]:
                own[lo:hi] = True
            assert torch.equal(o["grads"][own], base[own] * tot)               # reduced where this rank is the owner
            assert torch.equal(o["grads"][~own], o["local"][~own])             # untouched elsewhere (dead range included)
            assert abs(o["sumsq"] - float(((base[live] * tot) ** 2).sum())) < 1e-3 * o["sumsq"]
            want = torch.ones(n)
            want[live] -= 0.5 * (1.0 / world) * tot * base[live]
            assert torch.equal(o["params"], want)                               # after gather_master every rank has every range
            if prec == "bf16":
                assert torch.equal(o["params_lp"][live], want[live].bfloat16().float())
    for prec in ("bf16", "f32"):
        owns = [set(map(tuple, res[r]["sharded_" + prec]["mine"])) for r in range(world)]
        cover = torch.zeros(n, dtype=torch.int32)
        for r in range(world):
            for lo, hi in owns[r]:
                cover[lo:hi] += 1
        assert torch.equal(cover, live.int())                                   # the owners tile the live ranges exactly once
        if world == 8:
            assert any(hi == lo for lo, hi in res[7]["sharded_" + prec]["mine"])   # rank 7 owns an EMPTY part of the ragged chunk


def test_bench_launches_its_own_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus N` (the driver's command; WORLD_SIZE unset) starts N rank processes itself: rendezvous over gloo on
    the CPU, one JSON line from rank 0 with n_ranks_seen == N; a failing rank makes the parent exit non-zero."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # nothing but the JSON line on stdout (gloo's rendezvous banner goes to stderr)
    out = json.loads(lines[0])
    assert out["n_ranks_seen"] == 2 and out["n_gpus"] == 2 and out["sum"] == 2.0
    assert out["dp_mode"] == "sharded" and out["dp_fallback"] is None
    # a rank that fails under EVERY exchange mode: the parent exits non-zero (after its one fallback attempt) and prints no JSON line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--launch-check-fail-rank", "1"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_falls_back_to_allreduce_when_the_sharded_exchange_fails():
    """the first multi-GPU run must not come back empty.  A rank that fails only under RGQA_DP_MODE=sharded (the default)
    makes the supervising parent start a FRESH set of rank processes with RGQA_DP_MODE=allreduce: exit code 0, one JSON line, with
    `dp_mode` = allreduce and the reason in `dp_fallback`.  An explicitly requested mode is not second-guessed."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RGQA_DP_MODE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--launch-check-fail-rank", "1", "--launch-check-fail-mode", "sharded"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_ranks_seen"] == 2 and out["dp_mode"] == "allreduce" and "sharded exchange failed" in out["dp_fallback"]
    assert "fresh set of rank processes" in r.stderr
    # the same launch as ONE rank of a launcher (WORLD_SIZE set by torch.distributed.run): each rank supervises its own child
    port = 29900 + (os.getpid() % 500) * 4
    procs = [subprocess.Popen(cmd, env=dict(env, WORLD_SIZE="2", RANK=str(k), LOCAL_RANK=str(k), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = [pr.communicate(timeout=240) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [o[1][-1500:] for o in outs]
    got = [json.loads(l) for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(got) == 1 and got[0]["dp_mode"] == "allreduce" and got[0]["dp_fallback"] and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    r = subprocess.run(cmd, env=dict(env, RGQA_DP_MODE="allreduce"), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and json.loads(r.stdout.strip())["dp_fallback"] is None


def test_synth_batch_contract():
    from rgqa_amd import synth
    b = synth.synth_batch(8, 20, seed=1)
    assert b["feats"].shape == (8, 36, 2048) and b["feats"].min() >= 0 and b["boxes"].shape == (8, 36, 4)
    assert (b["boxes"][..., 0] <= b["boxes"][..., 2]).all() and (b["boxes"][..., 1] <= b["boxes"][..., 3]).all()
    ids, m = b["input_ids"], b["input_mask"]
    assert (ids[:, 0] == 101).all() and ((ids != 0) == (m == 1)).all()
    for r in range(8):
        L = int(m[r].sum())
        assert ids[r, L - 1] == 102 and 5 <= L <= 20
    assert set(np.unique(b["target"])) <= {0.0, 1.0} and (b["target"].sum(1) <= 1).all()
    b2 = synth.synth_batch(8, 20, seed=1)
    assert all(np.array_equal(b[k], b2[k]) for k in b)


def test_gradient_segments_cover_live_arena_once():
    """DP overlap contract: the engine's gradient segments (completion order) tile the live arena exactly once, never touch
    the dead range, and merge into few large buckets."""
    from rgqa_amd.engine import Engine
    from rgqa_amd.parallel import merge_segments
    e = Engine()
    segs = e.grad_segments()
    evs = [ev for _, _, ev in segs]
    assert evs == sorted(evs) and evs[0] == 0 and len(set(evs)) == 1 + 5 + 9 + 1
    cover = sorted((b, en) for b, en, _ in segs)
    assert cover[0][0] == 0 and all(a[1] <= b[0] for a, b in zip(cover, cover[1:]))
    db, de = e.dead_range
    holes = [(a[1], b[0]) for a, b in zip(cover, cover[1:]) if a[1] != b[0]]
    assert holes == [(db, de)] and cover[-1][1] == e.arena_elems
    buckets = merge_segments(segs, 64 * (1 << 20) // 4)
    assert sum(b[1] - b[0] for b in buckets) == sum(s[1] - s[0] for s in segs)
    assert len(buckets) <= 12 and [b[2] for b in buckets] == sorted(b[2] for b in buckets)
    flat = sorted((b[0], b[1]) for b in buckets)
    assert all(a[1] <= b[0] for a, b in zip(flat, flat[1:]))
    # every bucket waits for the LAST event among the segments it contains
    for b0, b1, ev in buckets:
        assert ev == max(s[2] for s in segs if b0 <= s[0] and s[1] <= b1)


def test_bench_profiling_section_has_no_collective():
    """bench.py profiles kernels on rank 0 only, after the timed region, while the other ranks wait at a barrier: the profiled
    steps must not contain the gradient all-reduce (that deadlocked every N>1 launch with the default --profile-steps)."""
    import inspect
    import bench
    src = inspect.getsource(bench.main)
    prof = src[src.index("e.profile(True)"):src.index("e.profile(False)")]
    assert "step(exchange=False)" in prof and "step()" not in prof
    assert "if c is not None and exchange:" in src


def test_committed_bench_line_keeps_the_contract():
    """The JSON line the driver parses (profiles/rNN_bench_n1.json = bench.py's stdout on an MI355X): every field of the contract, the
    metric / workload BASELINE.json names, roofline consistency (achieved / peak = frac; flops per launch / mean launch duration = achieved)
    and the cpu_baseline object.  A host test: it guards the line's SHAPE against edits of bench.py between GPU runs."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_n1.json")))
    assert files
    d = json.loads(open(files[-1]).readline())
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "QA-pairs/s"      # BASELINE's name without its ", 1/2/4/8 MI355X" tail (n_gpus says which)
    # (round 6: the headline is the precision whose logits are inside the north star's bound - the bf16 step is the named leg `config3_bf16`)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "bf16x3_fwd" and d["vs_baseline"] is None
    p = d["parity_in_run"]
    assert p["precision"] == d["dtype"] and p["within_bound"] is True and p["logits_max_err"] <= p["bound"] == 1e-3
    assert d["config3_bf16"]["precision"] == "bf16" and d["config3_bf16"]["within_bound"] is False
    assert d["seq30"]["seq_len"] == 30 and d["seq30"]["ms_per_step"] > d["ms_per_step"]
    assert all(d["other_workloads"][k]["within_logits_bound"] for k in ("roi_mixup_b256", "butd_b256"))
    assert abs(d["roofline"]["mfma_issued"]["achieved"] - 3 * d["roofline"]["achieved"]) < 0.05
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["gflop_per_launch"] / r["avg_launch_us"] * 1e3) / r["achieved"] < 5e-3      # GFLOP / us = 1000 TFLOP/s
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.9
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    t = d["tolerance_compliant"]
    assert t["precision"] == "bf16x3" and t["within_bound"] is True and t["logits_max_err"] <= t["bound"] == 1e-3
