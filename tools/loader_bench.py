"""Throughput of the input path alone: synthetic f16/f32 store on local disk -> DeviceBatcher batches of 256.
usage: python tools/loader_bench.py [n_images] [dtype]"""
import os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgqa_amd import data

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
O, F, NA, B = 36, 2048, 1842, 256
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
prefix = os.path.join(d, "store")
rng = np.random.default_rng(0)
np_dt = np.float16 if dtype == "f16" else np.float32
with open(prefix + ".feats.bin", "wb") as f:
    blk = np.maximum(rng.standard_normal((256, O, F), dtype=np.float32), 0).astype(np_dt)
    for _ in range(N // 256):
        f.write(blk.tobytes())
with open(prefix + ".boxes.bin", "wb") as f:
    f.write((rng.random((N, O, 4), dtype=np.float32) * 300).tobytes())
json.dump({"img_ids": ["n%d" % i for i in range(N)], "img_h": [400] * N, "img_w": [500] * N, "O": O, "F": F, "dtype": dtype}, open(prefix + ".meta.json", "w"))
st = data.FeatureStore(prefix)
ans2label = {"a%d" % k: k for k in range(NA)}
db = data.DeviceBatcher(st, ans2label, NA, B)
qs = [{"img_id": "n%d" % int(i), "question_id": str(k), "sent": "what is this?", "label": {"a%d" % (k % NA): 1.0}} for k, i in enumerate(rng.integers(0, N, 64 * B))]
for it in range(2):
    t0 = time.perf_counter()
    for k in range(0, len(qs), B):
        db.batch(qs[k:k + B])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pass %d: %d batches of %d from the %s store: %.2f ms/batch = %.0f QA pairs/s (host gather + H2D + device prepare)" % (it, len(qs) // B, B, dtype, dt / (len(qs) // B) * 1e3, len(qs) / dt), flush=True)
import shutil; shutil.rmtree(d)
