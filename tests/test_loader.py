"""Input path: TSV wire format -> binary store -> device batches, against the oracle and the golden vectors the
reference's own utils.load_obj_tsv produced (g10)."""
import os

import numpy as np
import pytest

from oracle import loader_ref as LR

NA = 23


@pytest.fixture(scope="module")
def g10(golden_dir):
    return np.load(os.path.join(golden_dir, "g10_loader.npz"))


@pytest.fixture()
def tsv(tmp_path):
    return LR.write_synthetic_tsv(str(tmp_path / "syn_obj36.tsv"), n_images=5, O=36, F=64, seed=3)


def test_oracle_decodes_tsv_like_the_reference(g10, tsv):
    imgs = LR.load_obj_tsv(tsv)
    assert [im["img_id"] for im in imgs] == g10["img_ids"].tolist()
    np.testing.assert_array_equal(np.array([[im["img_h"], im["img_w"]] for im in imgs]), g10["img_hw"])
    np.testing.assert_array_equal(np.stack([im["boxes"] for im in imgs]), g10["boxes_raw"])
    np.testing.assert_array_equal(np.stack([im["features"] for im in imgs]), g10["features"])
    assert len(LR.load_obj_tsv(tsv, topk=2)) == 2
    data, ans2label = LR.synthetic_questions([im["img_id"] for im in imgs], NA)
    by_id = {im["img_id"]: im for im in imgs}
    for i, d in enumerate(data):
        f, b, t = LR.getitem(by_id[d["img_id"]], d["label"], ans2label, NA)
        np.testing.assert_array_equal(b, g10["boxes_norm"][i])
        np.testing.assert_array_equal(t, g10["target"][i])
    assert g10["boxes_norm"].min() >= 0 and g10["boxes_norm"].max() <= 1 + 1e-5       # the reference asserts this (gqa_data.py:199-200)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_store_round_trip(g10, tsv, tmp_path, dtype):
    from rgqa_amd import data
    meta = data.tsv_to_store(tsv, str(tmp_path / "store"), dtype=dtype)
    assert meta["O"] == 36 and meta["F"] == 64 and meta["img_ids"] == g10["img_ids"].tolist()
    st = data.FeatureStore(str(tmp_path / "store"))
    assert len(st) == 5 and "n000102" in st and "nope" not in st
    np.testing.assert_array_equal(st.boxes, g10["boxes_raw"])
    np.testing.assert_array_equal(st.img_hw, g10["img_hw"])
    if dtype == "f32":
        np.testing.assert_array_equal(st.feats, g10["features"])
    else:
        np.testing.assert_array_equal(st.feats, g10["features"].astype(np.float16))
    rows = [3, 0, 3]
    f = np.empty((3, 36, 64), dtype=st.np_dtype); b = np.empty((3, 36, 4), dtype=np.float32); hw = np.empty((3, 2), dtype=np.int32)
    st.gather(rows, f, b, hw)
    np.testing.assert_array_equal(b, g10["boxes_raw"][rows])
    np.testing.assert_array_equal(hw, g10["img_hw"][rows])
    assert len(list(data.iter_obj_tsv(tsv, topk=2))) == 2
    with pytest.raises(ValueError):
        data.tsv_to_store(tsv, str(tmp_path / "bad"), dtype="f64")


def test_store_rejects_ragged_images(tmp_path):
    from rgqa_amd import data
    a = LR.write_synthetic_tsv(str(tmp_path / "a.tsv"), n_images=2, O=36, F=64)
    b = LR.write_synthetic_tsv(str(tmp_path / "b.tsv"), n_images=1, O=20, F=64)
    with pytest.raises(ValueError):
        data.tsv_to_store([a, b], str(tmp_path / "s"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_device_batches_match_reference_vectors(g10, tsv, tmp_path, dtype):
    """DeviceBatcher output vs the golden: boxes and targets bit-exact (IEEE f32 division / plain stores); features bit-exact from
    the f32 store and equal to the f16-rounded features from the f16 store. Several batches through both staging sets."""
    import torch
    from rgqa_amd import data
    data.tsv_to_store(tsv, str(tmp_path / "store"), dtype=dtype)
    st = data.FeatureStore(str(tmp_path / "store"))
    qs, ans2label = LR.synthetic_questions(g10["img_ids"].tolist(), NA)
    db = data.DeviceBatcher(st, ans2label, NA, max_batch=8)
    row_of = {k: i for i, k in enumerate(g10["img_ids"].tolist())}
    ref_f = g10["features"] if dtype == "f32" else g10["features"].astype(np.float16).astype(np.float32)
    for lo, hi in ((0, 5), (5, 12), (0, 8), (11, 12)):
        chunk = qs[lo:hi]
        qid, feats, boxes, sents, target = db.batch(chunk)
        torch.cuda.synchronize()
        assert qid == [d["question_id"] for d in chunk] and sents == [d["sent"] for d in chunk]
        assert feats.dtype == torch.float32 and feats.shape == (hi - lo, 36, 64) and boxes.shape == (hi - lo, 36, 4)
        np.testing.assert_array_equal(boxes.cpu().numpy(), g10["boxes_norm"][lo:hi])
        np.testing.assert_array_equal(target.cpu().numpy(), g10["target"][lo:hi])
        np.testing.assert_array_equal(feats.cpu().numpy(), ref_f[[row_of[d["img_id"]] for d in chunk]])
    qid, feats, boxes, sents, target = db.batch(qs[:3], with_target=False)
    assert target is None
    with pytest.raises(KeyError):
        db.batch([{"img_id": "missing", "question_id": "x", "sent": "?", "label": {}}])
    with pytest.raises(ValueError):
        db.batch(qs[:9])


@pytest.mark.gpu
def test_batch_prepare_full_size_properties():
    """BASELINE-size batch (256 x 36 x 2048, f16 store): f16 -> f32 expansion is exact, boxes land in [0,1], every target row holds
    exactly its labels."""
    import ctypes as C
    import torch
    from rgqa_amd import _lib, synth
    lib = _lib.load()
    B, O, F, NAF = 256, 36, 2048, 1842
    f16 = torch.from_numpy(np.maximum(synth.uniform("bp.f", (B, O, F), -2.0, 6.0), 0).astype(np.float16)).cuda()
    hw = torch.from_numpy(np.stack([300 + np.arange(B) % 200, 400 + np.arange(B) % 333], 1).astype(np.int32)).cuda()
    u = synth.uniform("bp.b", (B, O, 4), 0.0, 1.0)
    px = torch.from_numpy((u * np.stack([hw.cpu().numpy()[:, 1], hw.cpu().numpy()[:, 0]] * 2, 1)[:, None, :]).astype(np.float32)).cuda()
    offs = torch.arange(0, 2 * B + 1, 2, dtype=torch.int32).cuda()
    labs = torch.from_numpy(((np.arange(2 * B) * 7919) % NAF).astype(np.int32)).cuda()
    labs[5] = -1
    scs = torch.from_numpy(synth.uniform("bp.s", (2 * B,), 0.1, 1.0)).cuda()
    feats = torch.empty(B, O, F, device="cuda"); boxes = torch.empty(B, O, 4, device="cuda"); target = torch.full((B, NAF), 7.0, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.rgqa_batch_prepare(p(f16), 1, p(feats), p(px), p(hw), p(boxes), p(offs), p(labs), p(scs), p(target), NAF, B, O, F, NAF,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(feats, f16.float())
    assert float(boxes.min()) >= 0 and float(boxes.max()) <= 1 + 1e-5
    assert torch.equal(boxes[:, :, 0], px[:, :, 0] / hw[:, 1:2].float()) and torch.equal(boxes[:, :, 3], px[:, :, 3] / hw[:, 0:1].float())
    t = target.cpu().numpy(); l = labs.cpu().numpy(); s = scs.cpu().numpy()
    assert (t != 0).sum() == 2 * B - 1
    for b in (0, 2, 100, 255):
        for k in (2 * b, 2 * b + 1):
            if l[k] >= 0:
                assert t[b, l[k]] == s[k]
