"""how far apart are two RUNS of the drop-in trainer on a bf16x3 engine?  (test_trainer_step_fused_clip_and_operand_copies[bf16x3] compares two runs)
python3 tools/x3_repro.py [trials]   RGQA_ADAM_OVERLAP=0/1 from the environment"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dropin"))
import numpy as np, torch
os.environ["RGQA_BERT_VOCAB"] = os.path.join(ROOT, "tests", "golden", "g4_vocab.txt")
import rgqa_amd.lxrt.modeling as M
from rgqa_amd import synth
import tests.test_gpu_dropin as T
CFG = T.CFG
M.VISUAL_CONFIG.visual_feat_dim = CFG["feat_dim"]
M.LXRTFeatureExtraction.from_pretrained = classmethod(lambda cls, name, **kw: cls(M.BertConfig(CFG["vocab_size"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"],
    intermediate_size=CFG["inter"], max_position_embeddings=CFG["max_pos"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0), **kw))
import lxrt.entry
from lxrt.optimization import BertAdam
feats, boxes, target = T.batch(20)
def run(defer):
    os.environ["RGQA_DEFER_CLIP"] = defer
    m, _ = T.build("bf16x3", 20)
    m.train()
    optim = BertAdam(list(m.parameters()), lr=1e-3, warmup=0.1, t_total=20)
    for step in range(3):
        optim.zero_grad()
        logit = m(feats.cuda(), boxes.cuda(), T.SENTS)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, target.cuda()) * logit.size(1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
        optim.step()
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in m.named_parameters()}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for t in range(n):
    a, b = run("1"), run("0" if t % 2 == 0 else "1")
    worst = (0.0, "", 0.0)
    bad = 0
    for k in a:
        d = (a[k] - b[k]).abs()
        bad += int((d > 2e-6 * b[k].abs() + 1e-8).sum())
        if float(d.max()) > worst[0]:
            worst = (float(d.max()), k, float(b[k].abs().max()))
    print("trial %d (fast vs %s): worst |diff| %.3e in %s (|p| max %.2e); elements outside rtol 2e-6 / atol 1e-8: %d" % (t, "inplace" if t % 2 == 0 else "fast", worst[0], worst[1], worst[2], bad), flush=True)
