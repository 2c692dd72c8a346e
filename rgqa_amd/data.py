"""Input path: the reference keeps Faster-RCNN features as a base64 TSV that `utils.load_obj_tsv`
(utils.py:16-54) decodes into Python dicts held in host memory, and `GQATorchDataset.__getitem__` (tasks/gqa_data.py:173-238)
copies, normalises and labels sample by sample.  Here:

  * `tsv_to_store`    one-off conversion of the same TSV into a flat binary store: feats [N,O,F] f16 or f32, boxes [N,O,4] f32
                       (pixels, as in the TSV), image sizes and ids in a JSON side file - mmap-able, no decode at start-up;
  * `FeatureStore`    the mmap view; `gather` copies a batch's rows into pinned staging buffers;
  * `DeviceBatcher`   per batch: one pinned gather, one async host->device copy (f16: 147 KB/sample instead of 295 KB),
                       then `rgqa_batch_prepare` (csrc/loader.hip) expands features to f32, normalises the boxes and builds the
                       soft targets ON THE DEVICE.  Returns what the reference's DataLoader hands the trainer
                       (tasks/gqa_conf.py:150-153): ques_id, feats, boxes, sent, target.

No CPU fallback: batches are produced by the HIP library or not at all."""
import base64
import csv
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

from . import _lib

FIELDNAMES = ["img_id", "img_h", "img_w", "objects_id", "objects_conf", "attrs_id", "attrs_conf", "num_boxes", "boxes", "features"]   # utils.py:12-13


def iter_obj_tsv(fname, topk=None):
    """Rows of the reference's detection TSV (wire format of utils.py:16-54): img_id, img_h, img_w, num_boxes, boxes [n,4] f32
    (pixels), features [n,F] f32. The object / attribute columns are not needed on the GQA path and are skipped undecoded."""
    csv.field_size_limit(sys.maxsize)
    with open(fname) as f:
        for i, item in enumerate(csv.DictReader(f, FIELDNAMES, delimiter="\t")):
            if topk is not None and topk >= 0 and i >= topk:
                break
            n = int(item["num_boxes"])
            yield {"img_id": item["img_id"], "img_h": int(item["img_h"]), "img_w": int(item["img_w"]), "num_boxes": n,
                   "boxes": np.frombuffer(base64.b64decode(item["boxes"]), dtype=np.float32).reshape(n, 4),
                   "features": np.frombuffer(base64.b64decode(item["features"]), dtype=np.float32).reshape(n, -1)}


def tsv_to_store(tsv_paths, prefix, dtype="f16", topk=None):
    """Converts one or more detection TSVs into `<prefix>.feats.bin`, `<prefix>.boxes.bin`, `<prefix>.meta.json`. All images must
    carry the same number of boxes and feature width (36 x 2048 for the reference's GQA features). Returns the meta dict."""
    if isinstance(tsv_paths, str):
        tsv_paths = [tsv_paths]
    if dtype not in ("f16", "f32"):
        raise ValueError("dtype must be 'f16' or 'f32'")
    np_dt = np.float16 if dtype == "f16" else np.float32
    meta = {"img_ids": [], "img_h": [], "img_w": [], "O": None, "F": None, "dtype": dtype}
    with open(prefix + ".feats.bin", "wb") as ff, open(prefix + ".boxes.bin", "wb") as fb:
        for path in tsv_paths:
            for it in iter_obj_tsv(path, topk):
                O, F = it["features"].shape
                if meta["O"] is None:
                    meta["O"], meta["F"] = O, F
                if (O, F) != (meta["O"], meta["F"]) or it["boxes"].shape != (O, 4):
                    raise ValueError("%s: image %s has %dx%d features, the store holds %dx%d" % (path, it["img_id"], O, F, meta["O"], meta["F"]))
                ff.write(np.ascontiguousarray(it["features"], dtype=np_dt).tobytes())
                fb.write(np.ascontiguousarray(it["boxes"], dtype=np.float32).tobytes())
                meta["img_ids"].append(it["img_id"]); meta["img_h"].append(it["img_h"]); meta["img_w"].append(it["img_w"])
    with open(prefix + ".meta.json", "w") as f:
        json.dump(meta, f)
    return meta


class FeatureStore(object):
    def __init__(self, prefix, threads=None):
        self.threads = int(threads or os.environ.get("RGQA_LOADER_THREADS", 8))     # host threads of the batch gather
        with open(prefix + ".meta.json") as f:
            self.meta = json.load(f)
        self.O, self.F, self.dtype = self.meta["O"], self.meta["F"], self.meta["dtype"]
        self.N = len(self.meta["img_ids"])
        self.np_dtype = np.float16 if self.dtype == "f16" else np.float32
        self.feats = np.memmap(prefix + ".feats.bin", dtype=self.np_dtype, mode="r", shape=(self.N, self.O, self.F))
        self.boxes = np.memmap(prefix + ".boxes.bin", dtype=np.float32, mode="r", shape=(self.N, self.O, 4))
        self.img_hw = np.stack([np.asarray(self.meta["img_h"], dtype=np.int32), np.asarray(self.meta["img_w"], dtype=np.int32)], 1)
        self.row_of = {k: i for i, k in enumerate(self.meta["img_ids"])}

    def __len__(self):
        return self.N

    def __contains__(self, img_id):
        return img_id in self.row_of

    def gather(self, rows, feats_out, boxes_out, hw_out):
        """rows: store rows of a batch; the *_out arrays are (pinned) host buffers of shape [B,O,F], [B,O,4], [B,2]."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        if not (feats_out.flags.c_contiguous and feats_out.dtype == self.np_dtype and feats_out.shape == (len(rows), self.O, self.F)):
            raise ValueError("gather: feats_out must be a C-contiguous [%d,%d,%d] %s array" % (len(rows), self.O, self.F, self.dtype))
        lib = _lib.load()
        _lib.check(lib.rgqa_host_gather_rows(C.c_void_p(self.feats.ctypes.data), self.O * self.F * self.feats.itemsize, self.N,
                                             C.c_void_p(rows.ctypes.data), len(rows), C.c_void_p(feats_out.ctypes.data), self.threads))
        np.take(self.boxes, rows, axis=0, out=boxes_out)
        np.take(self.img_hw, rows, axis=0, out=hw_out)


class DeviceBatcher(object):
    """Builds engine-ready batches from GQADataset-style datum dicts ({'img_id', 'question_id', 'sent', 'label': {answer: score}},
    tasks/gqa_data.py:45-62) and a FeatureStore.  Everything that crosses PCIe goes from pinned staging buffers on a COPY stream of its
    own: the trainer's host thread runs ahead of the GPU, so the copies of batch i+1 proceed (on the DMA engines) while the compute
    stream still executes step i, and the compute stream only waits for the copy's event.  Two staging sets alternate."""

    LABEL_CAP = 16          # staged labels per sample (GQA: one answer per question; more grows the buffers)

    def __init__(self, store, ans2label, num_answers, max_batch, device="cuda:0"):
        self.store, self.ans2label, self.NA, self.maxB = store, ans2label, int(num_answers), int(max_batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatcher needs a CUDA/HIP device: batches are prepared by the HIP library (no CPU fallback)")
        self.lib = _lib.load()
        O, F = store.O, store.F
        tdt = torch.float16 if store.dtype == "f16" else torch.float32
        from . import streams
        # a stream shown to run beside the caller's (streams.py): the third of the device's set - the sharded exchange's second stream, which
        # the trainer loop that feeds from a DeviceBatcher (all-reduce exchange) does not use
        self.copy_stream = streams.pick(self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device()), 3)[2]
        self.sets = []
        for _ in range(2):
            s = {"feats_h": torch.empty(self.maxB, O, F, dtype=tdt).pin_memory(), "boxes_h": torch.empty(self.maxB, O, 4).pin_memory(),
                 "hw_h": torch.empty(self.maxB, 2, dtype=torch.int32).pin_memory(),
                 "feats_raw": torch.empty(self.maxB, O, F, dtype=tdt, device=self.device), "boxes_raw": torch.empty(self.maxB, O, 4, device=self.device),
                 "hw": torch.empty(self.maxB, 2, dtype=torch.int32, device=self.device), "event": None, "copied": None}
            self._label_buffers(s, self.maxB * self.LABEL_CAP)
            self.sets.append(s)
        self.turn = 0

    def _label_buffers(self, s, cap):
        s["lab_cap"] = cap
        s["offs_h"] = torch.empty(self.maxB + 1, dtype=torch.int32).pin_memory()
        s["labs_h"] = torch.empty(cap, dtype=torch.int32).pin_memory()
        s["scs_h"] = torch.empty(cap, dtype=torch.float32).pin_memory()
        s["offs"] = torch.empty(self.maxB + 1, dtype=torch.int32, device=self.device)
        s["labs"] = torch.empty(cap, dtype=torch.int32, device=self.device)
        s["scs"] = torch.empty(cap, dtype=torch.float32, device=self.device)

    def batch(self, data, with_target=True):
        """-> (ques_ids, feats [B,O,F] f32, boxes [B,O,4] f32, sents, target [B,NA] f32 or None), device tensors, like one
        iteration of the reference's DataLoader + the `.cuda()` calls of tasks/gqa_conf.py:153."""
        B = len(data)
        if B == 0 or B > self.maxB:
            raise ValueError("batch of %d samples (1..%d supported)" % (B, self.maxB))
        st = self.store
        try:
            rows = [st.row_of[d["img_id"]] for d in data]
        except KeyError as e:
            raise KeyError("image %s is not in the feature store" % e)
        s = self.sets[self.turn]
        self.turn ^= 1
        if s["copied"] is not None:
            s["copied"].synchronize()                    # the batch staged through this set two calls ago has left the pinned buffers
        st.gather(rows, s["feats_h"][:B].numpy(), s["boxes_h"][:B].numpy(), s["hw_h"][:B].numpy())
        nlab = 0
        if with_target:
            off, lab, sc = [0], [], []
            for d in data:
                for ans, score in d.get("label", {}).items():
                    lab.append(self.ans2label.get(ans, -1)); sc.append(score)
                off.append(len(lab))
            nlab = len(lab)
            if nlab > s["lab_cap"]:
                if s["event"] is not None:
                    s["event"].synchronize()
                self._label_buffers(s, 2 * nlab)
            s["offs_h"][:B + 1] = torch.tensor(off, dtype=torch.int32)
            if nlab:
                s["labs_h"][:nlab] = torch.tensor(lab, dtype=torch.int32)
                s["scs_h"][:nlab] = torch.tensor(sc, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            if s["event"] is not None:
                self.copy_stream.wait_event(s["event"])  # the device-side staging buffers of this set are no longer being read
            s["feats_raw"][:B].copy_(s["feats_h"][:B], non_blocking=True)
            s["boxes_raw"][:B].copy_(s["boxes_h"][:B], non_blocking=True)
            s["hw"][:B].copy_(s["hw_h"][:B], non_blocking=True)
            if with_target:
                s["offs"][:B + 1].copy_(s["offs_h"][:B + 1], non_blocking=True)
                if nlab:
                    s["labs"][:nlab].copy_(s["labs_h"][:nlab], non_blocking=True)
                    s["scs"][:nlab].copy_(s["scs_h"][:nlab], non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.copy_stream)
        s["copied"] = copied
        stream.wait_event(copied)
        feats = torch.empty(B, st.O, st.F, dtype=torch.float32, device=self.device)
        boxes = torch.empty(B, st.O, 4, dtype=torch.float32, device=self.device)
        target = torch.empty(B, self.NA, dtype=torch.float32, device=self.device) if with_target else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self.lib.rgqa_batch_prepare(p(s["feats_raw"]), 1 if st.dtype == "f16" else 0, p(feats), p(s["boxes_raw"]), p(s["hw"]), p(boxes),
                                               p(s["offs"]) if with_target else None, p(s["labs"]) if with_target else None, p(s["scs"]) if with_target else None,
                                               p(target), self.NA, B, st.O, st.F, self.NA, C.c_void_p(stream.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(stream)
        s["event"] = ev
        return [d["question_id"] for d in data], feats, boxes, [d["sent"] for d in data], target
