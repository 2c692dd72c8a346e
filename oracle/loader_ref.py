"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's input path.

  * load_obj_tsv            - utils.py:16-54: base64-TSV of Faster-RCNN detections -> list of dicts
  * getitem                 - tasks/gqa_data.py:173-238 (LXMERT branch): box normalisation, soft-target build

Pinned by tests/golden/g10_loader.npz: oracle/gen_golden.py wrote a synthetic TSV (write_synthetic_tsv below), decoded it
with the reference's OWN utils.load_obj_tsv, and stored what it returned."""
import base64
import csv
import sys

import numpy as np

FIELDNAMES = ["img_id", "img_h", "img_w", "objects_id", "objects_conf", "attrs_id", "attrs_conf", "num_boxes", "boxes", "features"]


def load_obj_tsv(fname, topk=None):
    """utils.py:16-54: ints for img_h / img_w / num_boxes; the six array columns are base64 of raw little-endian buffers."""
    csv.field_size_limit(sys.maxsize)
    data = []
    with open(fname) as f:
        for item in csv.DictReader(f, FIELDNAMES, delimiter="\t"):
            for key in ("img_h", "img_w", "num_boxes"):
                item[key] = int(item[key])
            n = item["num_boxes"]
            for key, shape, dtype in (("objects_id", (n,), np.int64), ("objects_conf", (n,), np.float32), ("attrs_id", (n,), np.int64),
                                      ("attrs_conf", (n,), np.float32), ("boxes", (n, 4), np.float32), ("features", (n, -1), np.float32)):
                item[key] = np.frombuffer(base64.b64decode(item[key]), dtype=dtype).reshape(shape)
            data.append(item)
            if topk is not None and len(data) == topk:
                break
    return data


def getitem(img_info, label, ans2label, num_answers):
    """tasks/gqa_data.py:186-200 (boxes to 0..1 by image width / height, float32 in place) and :213-217 (soft target)."""
    boxes = img_info["boxes"].copy()
    feats = img_info["features"].copy()
    boxes[:, (0, 2)] /= img_info["img_w"]
    boxes[:, (1, 3)] /= img_info["img_h"]
    target = None
    if label is not None:
        target = np.zeros(num_answers, dtype=np.float32)
        for ans, score in label.items():
            if ans in ans2label:
                target[ans2label[ans]] = score
    return feats, boxes, target


def write_synthetic_tsv(path, n_images=5, O=36, F=64, seed=3):
    """A TSV in the reference's wire format with deterministic content (rgqa_amd.synth hashes): post-ReLU-like features,
    boxes inside the image."""
    from rgqa_amd import synth
    rows = []
    for i in range(n_images):
        h, w = 300 + 37 * i, 400 + 53 * i
        feats = np.maximum(synth.uniform("tsv.f%d.%d" % (seed, i), (O, F), -2.0, 6.0), 0).astype(np.float32)
        u = synth.uniform("tsv.b%d.%d" % (seed, i), (O, 4), 0.0, 1.0)
        x1 = np.minimum(u[:, 0], u[:, 2]) * w; x2 = np.maximum(u[:, 0], u[:, 2]) * w
        y1 = np.minimum(u[:, 1], u[:, 3]) * h; y2 = np.maximum(u[:, 1], u[:, 3]) * h
        boxes = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
        oid = (synth.hash_u32("tsv.o%d.%d" % (seed, i), O) % np.uint64(1600)).astype(np.int64)
        aid = (synth.hash_u32("tsv.a%d.%d" % (seed, i), O) % np.uint64(400)).astype(np.int64)
        oc = synth.uniform("tsv.oc%d.%d" % (seed, i), (O,), 0.0, 1.0)
        ac = synth.uniform("tsv.ac%d.%d" % (seed, i), (O,), 0.0, 1.0)
        enc = lambda a: base64.b64encode(np.ascontiguousarray(a).tobytes()).decode("ascii")
        rows.append(["n%06d" % (100 + i), str(h), str(w), enc(oid), enc(oc), enc(aid), enc(ac), str(O), enc(boxes), enc(feats)])
    with open(path, "w") as f:
        for r in rows:
            f.write("\t".join(r) + "\n")
    return path


def synthetic_questions(img_ids, num_answers, n=12):
    """GQADataset-style datum dicts (tasks/gqa_data.py:45-62) over the synthetic images, incl. an unlabeled one, an
    out-of-vocabulary answer, a two-answer soft label and a duplicated image."""
    ans2label = {"ans%d" % k: k for k in range(num_answers)}
    data = []
    for q in range(n):
        label = {"ans%d" % ((7 * q + 3) % num_answers): 1.0}
        if q % 4 == 1:
            label["ans%d" % ((5 * q + 1) % num_answers)] = 0.3
        if q % 5 == 2:
            label = {"never-seen-answer": 1.0}
        if q % 6 == 5:
            label = {}
        data.append({"img_id": img_ids[(q * 2) % len(img_ids)], "question_id": "q%04d" % q, "sent": "what is object %d doing?" % q, "label": label})
    return data, ans2label
