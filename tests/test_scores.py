"""Test-time scoring of logits: oracle vs the golden vectors produced from the reference's torch
expressions (CPU), and the fused HIP kernel vs both (GPU)."""
import os

import numpy as np
import pytest

from oracle import score_ref as R


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_scores.npz"))


def test_oracle_scores_match_reference_vectors(g9):
    x = g9["logit"]
    for temp in (1.0, 1000.0):
        s, l = R.sigmoid_max(x, temp)
        np.testing.assert_allclose(s, g9["max_score_T%g" % temp], rtol=2e-7, atol=0)
        np.testing.assert_array_equal(l, g9["label_T%g" % temp])
    e = R.energy(x)
    assert np.isinf(e[2]) and np.isinf(g9["energy"][2])         # the naive softplus overflows exactly as the reference's
    fin = np.isfinite(g9["energy"])
    np.testing.assert_allclose(e[fin], g9["energy"][fin], rtol=2e-6)
    for k in (2, 5):
        v, i = R.topk(x, k)
        np.testing.assert_array_equal(v, g9["topk%d_values" % k])
        # torch leaves the order of equal values unspecified: compare the index SETS where values tie, exactly otherwise
        for r in range(x.shape[0]):
            if len(set(v[r])) == k:
                np.testing.assert_array_equal(i[r], g9["topk%d_indices" % k][r])
            else:
                np.testing.assert_array_equal(x[r, i[r]], x[r, g9["topk%d_indices" % k][r]])
        te = R.topk_energy(x, k)
        fin = np.isfinite(g9["topk%d_energy" % k])
        np.testing.assert_allclose(te[fin], g9["topk%d_energy" % k][fin], rtol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 2, 5])
def test_score_rows_kernel_vs_golden(g9, k):
    import torch
    from rgqa_amd import scoring
    x = torch.from_numpy(g9["logit"]).cuda()
    for temp in (1.0, 1000.0):
        s = scoring.score_rows(x, temp, k)
        np.testing.assert_allclose(s.max_score.cpu().numpy(), g9["max_score_T%g" % temp], rtol=1e-6, atol=0)
        np.testing.assert_array_equal(s.label.cpu().numpy(), g9["label_T%g" % temp])
    e = s.energy.cpu().numpy()
    fin = np.isfinite(g9["energy"])
    assert np.array_equal(np.isinf(e), ~fin)
    np.testing.assert_allclose(e[fin], g9["energy"][fin], rtol=1e-5)
    if k:
        v, i = s.topk_values.cpu().numpy(), s.topk_indices.cpu().numpy()
        np.testing.assert_array_equal(v, g9["topk%d_values" % k])
        rv, ri = R.topk(g9["logit"], k)                            # ties: ascending index, as the oracle defines
        np.testing.assert_array_equal(i, ri)
        te = s.topk_energy.cpu().numpy()
        fin = np.isfinite(g9["topk%d_energy" % k])
        np.testing.assert_allclose(te[fin], g9["topk%d_energy" % k][fin], rtol=1e-5)
        assert np.array_equal(np.isinf(te), ~fin)


@pytest.mark.gpu
def test_score_rows_large_batch_and_strided_vs_oracle():
    """B not a multiple of the 4 rows per block, a row stride > NA (the engine's padded logits), NA not a multiple of 64."""
    import torch
    from rgqa_amd import scoring, synth
    B, NA, ld = 259, 1842, 1856
    full = synth.uniform("scores.big", (B, ld), -9.0, 7.0)
    x = torch.from_numpy(full).cuda()[:, :NA]
    s = scoring.score_rows(x, 2.0, 3)
    ref = full[:, :NA]
    ms, lab = R.sigmoid_max(ref, 2.0)
    np.testing.assert_allclose(s.max_score.cpu().numpy(), ms, rtol=1e-6)
    np.testing.assert_array_equal(s.label.cpu().numpy(), lab)
    np.testing.assert_allclose(s.energy.cpu().numpy(), R.energy(ref), rtol=1e-5)
    v, i = R.topk(ref, 3)
    np.testing.assert_array_equal(s.topk_values.cpu().numpy(), v)
    np.testing.assert_array_equal(s.topk_indices.cpu().numpy(), i)
    np.testing.assert_allclose(s.topk_energy.cpu().numpy(), R.topk_energy(ref, 3), rtol=1e-5)


@pytest.mark.gpu
def test_score_rows_rejects_bad_arguments():
    import torch
    from rgqa_amd import scoring
    x = torch.zeros(4, 10, device="cuda")
    with pytest.raises(RuntimeError):
        scoring.score_rows(x, 1.0, 11)
    with pytest.raises(RuntimeError):
        scoring.score_rows(x, 0.0, 0)
    with pytest.raises(ValueError):
        scoring.score_rows(x.double(), 1.0, 0)
