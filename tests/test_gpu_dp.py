"""Data-parallel step == single-process step on the concatenated batch, two ranks sharing the one
GPU of the test box over gloo (RCCL needs one device per rank; the code path — segment events, side stream, bucketed
async all-reduce, grad_prescale — is the same one bench.py runs over RCCL)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MED = dict(vocab_size=512, hidden=128, heads=2, inter=256, max_pos=64, type_vocab=2, l_layers=3, x_layers=2, r_layers=2,
           feat_dim=64, pos_dim=4, num_answers=70)
B, T, O = 4, 12, 10


def _make(batch):
    from rgqa_amd.engine import Engine
    from rgqa_amd import synth
    e = Engine(precision="f32", hidden_dropout=0.0, attn_dropout=0.0, **MED).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    d = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
    e.ensure_shape(d["feats"].shape[0], T, O)
    e.sync_weights()
    return e, d


def _step(e, d, comm, world):
    e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], train=False)
    e.loss_backward(d["target"])
    if comm is not None:
        comm.all_reduce()
    e.adam_step(1e-3, max_norm=5.0, grad_prescale=1.0 / world)


def _full_batch():
    from rgqa_amd import synth
    return synth.synth_batch(2 * B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=31, min_len=2)


def _worker(rank, world, port, overlap, q):
    import torch.distributed as dist
    from rgqa_amd.parallel import GradAllReduce
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    full = _full_batch()
    shard = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
    e, d = _make(shard)
    comm = GradAllReduce(e, dist, bucket_mb=1, overlap=overlap)
    for _ in range(2):
        _step(e, d, comm, world)
    torch.cuda.synchronize()
    q.put((rank, e.params.cpu().numpy(), len(comm.buckets)))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_step_equals_single_rank_on_concatenated_batch(overlap):
    import torch.multiprocessing as mp
    e, d = _make(_full_batch())
    for _ in range(2):
        _step(e, d, None, 1)
    ref = e.params.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + (7 if overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res, nbs = {}, []
    try:
        for _ in range(2):
            r, params, nb = q.get(timeout=240)
            res[r] = params
            nbs.append(nb)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert all(nb >= (3 if overlap else 1) for nb in nbs)
    assert np.array_equal(res[0], res[1])                      # replicas stay bit-identical
    np.testing.assert_allclose(res[0], ref, rtol=2e-4, atol=2e-6)
