"""ORACLE (test infrastructure only) — CPU PyTorch restatement of the reference's BUTD GQA model (src/butd/butd.py),
functional over a {state_dict key: tensor} mapping; pinned by tests/golden/g7_butd.npz generated from the reference
itself (oracle/gen_golden.py butd).  BASELINE config 5 / SURVEY.md §8 A23."""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class ButdConfig:
    ntoken: int = 3000          # dictionary size; padding index = ntoken (butd/preprocess.py:27-29)
    emb_dim: int = 300
    hidden: int = 1024
    v_dim: int = 2048
    pos_dim: int = 4
    num_answers: int = 1842
    max_len: int = 40           # MAX_GQA_LENGTH (butd.py:6)


def param_shapes(c):
    H, D = c.hidden, c.v_dim + c.pos_dim
    out = {"w_emb.emb.weight": (c.ntoken + 1, c.emb_dim),
           "q_enc.rnn.weight_ih_l0": (3 * H, c.emb_dim), "q_enc.rnn.weight_hh_l0": (3 * H, H),
           "q_enc.rnn.bias_ih_l0": (3 * H,), "q_enc.rnn.bias_hh_l0": (3 * H,)}

    def wn(name, o, i):
        out[name + ".bias"] = (o,)
        out[name + ".weight_g"] = ()
        out[name + ".weight_v"] = (o, i)
    wn("att.image_proj.mlp.0", H, D)
    wn("att.question_proj.mlp.0", H, H)
    wn("att.linear", 1, H)
    wn("q_project.mlp.0", H, H)
    wn("img_project.mlp.0", H, D)
    wn("ans_classifier.0", 2 * H, H)
    wn("ans_classifier.3", c.num_answers, 2 * H)
    return out


def wn_linear(x, P, name):
    """weight_norm(nn.Linear, dim=None): W = g * V / ||V||_F with a scalar g (butd.py:17, 85, 170-178)."""
    v, g = P[name + ".weight_v"], P[name + ".weight_g"]
    return F.linear(x, v * (g / v.norm()), P[name + ".bias"])


def gru_last(x, P):
    """nn.GRU(emb, hidden, 1, batch_first=True) from h0 = 0, returning output[:, -1] (butd.py:48-66). Gate order r, z, n."""
    Wih, Whh = P["q_enc.rnn.weight_ih_l0"], P["q_enc.rnn.weight_hh_l0"]
    bih, bhh = P["q_enc.rnn.bias_ih_l0"], P["q_enc.rnn.bias_hh_l0"]
    B, L, _ = x.shape
    H = Whh.shape[1]
    h = x.new_zeros(B, H)
    for t in range(L):
        gi = F.linear(x[:, t], Wih, bih)
        gh = F.linear(h, Whh, bhh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
    return h


def butd_forward(P, c, feat, pos, toks, want_att=False):
    """GQABUTD.forward (butd.py:195-221), dropout off."""
    w = F.embedding(toks, P["w_emb.emb.weight"], padding_idx=c.ntoken)
    q = gru_last(w, P)
    img = torch.cat([feat, pos], dim=2)
    ip = torch.relu(wn_linear(img, P, "att.image_proj.mlp.0"))
    qp = torch.relu(wn_linear(q, P, "att.question_proj.mlp.0")).unsqueeze(1)
    logits = wn_linear(ip * qp, P, "att.linear")
    att = torch.softmax(logits, dim=1)
    img_enc = (img * att).sum(dim=1)
    q_repr = torch.relu(wn_linear(q, P, "q_project.mlp.0"))
    img_repr = torch.relu(wn_linear(img_enc, P, "img_project.mlp.0"))
    joint = q_repr * img_repr
    out = wn_linear(torch.relu(wn_linear(joint, P, "ans_classifier.0")), P, "ans_classifier.3")
    return (out, att) if want_att else out


def tokenize(sentences, word2idx, max_len=40):
    """Dictionary.tokenize (preprocess.py:31-44) + front padding (butd.py:180-193)."""
    pad = len(word2idx)
    rows = []
    for s in sentences:
        s = s.lower().replace(",", "").replace(".", "").replace("?", "").replace("'s", " 's")
        t = [word2idx.get(w, pad) for w in s.split()][:max_len]
        rows.append([pad] * (max_len - len(t)) + t)
    return rows
