"""Worker of tests/test_sharded_c3_gpu.py (b): world_size-2 run of the sharded embedding on the REAL HIP kernels (both
ranks on the test box's one GPU, collectives over gloo) at the C3' table sizes -- the 26 Criteo-1TB vocabularies capped
at 1,000,000 rows (7.1 M rows in all), dim 128, bf16 tables with fp32 Adagrad accumulators, the ml_perf bag lengths,
B_local = 4096 -- compared with the ORACLE (oracle/krs_oracle.c: embed_bag_fwd / embed_bag_bwd_dense / apply_optimizer on
the UNSHARDED tables, fed the two ranks' batches one after the other).  Static-capacity exchange, partials in fp32."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import krs_oracle as ko  # noqa: E402
from tests.helpers import to_np  # noqa: E402

CRITEO = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938,
          155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import keras_rs_amd.layers as kl
    from keras_rs_amd.sharded import ShardedDistributedEmbedding

    V = [min(v, 1_000_000) for v in CRITEO]
    T, D, B, LR, ACC0 = 26, 128, 4096, 0.0034, 0.1
    opt = kl.Adagrad(learning_rate=LR, initial_accumulator_value=ACC0)
    tcs = [kl.TableConfig(f"cat_{t}", V[t], D, optimizer=opt, combiner="sum", placement="sparsecore") for t in range(T)]
    feats = {f"cat_{t:02d}_id": kl.FeatureConfig(f"cat_{t}", tcs[t], (B, HOTS[t]), (B, D)) for t in range(T)}
    layer = ShardedDistributedEmbedding(feats, dtype="bfloat16", partial_dtype="float32", exchange="static")
    layer.build(None)
    # the full tables, known to both ranks: bf16 values drawn on the host from one seed
    rng = np.random.default_rng(1337)
    full = [ko.f32_to_bf16_bits(rng.uniform(-0.05, 0.05, (V[t], D)).astype(np.float32)) for t in range(T)]
    g = layer._sgroups[0]
    with torch.no_grad():
        for t in range(T):
            mine = torch.from_numpy(full[t][rank::world].view(np.int16)).view(torch.bfloat16)
            layer.shard.data[g.row_off[t]: g.row_off[t] + mine.shape[0]].copy_(mine)
    ids = [{f"cat_{t:02d}_id": np.random.default_rng(1338 + 100 * r + t).integers(0, V[t], (B, HOTS[t])).astype(np.int32)
            for t in range(T)} for r in range(world)]
    grads = [(np.random.default_rng(1339 + r).uniform(-1, 1, (B, T * D)).astype(np.float32)) for r in range(world)]
    gq = [ko.f32_to_bf16_bits(x) for x in grads]                      # the gradient the layer sees is bf16

    out = layer(ids[rank])
    views = [out[k] for k in feats]
    gt = torch.from_numpy(gq[rank].view(np.int16)).view(torch.bfloat16).cuda()
    torch.autograd.backward(views, [gt[:, i * D:(i + 1) * D] for i in range(T)])
    torch.cuda.synchronize()
    layer.check_ids(wait=True)
    assert layer.last_exchange["mode"] == "static" and layer.overflow_steps == 0
    got = np.concatenate([to_np(v) for v in views], axis=1)            # bf16 bits [B, T*D]
    after = layer.get_embedding_tables()                                # collective: every rank takes part

    if rank == 0:
        tabs = ko.make_tables(full)
        f = ko.make_features(list(range(T)), ["sum"] * T, [t * D for t in range(T)], hots=HOTS, batch=B)
        flat = np.concatenate([ids[0][k].reshape(-1) for k in feats])
        exp = np.zeros((B, T * D), np.uint16)
        ko.embed_bag_fwd_raw(tabs, ko.BF16, f, flat, None, None, B, D, exp)
        # forward: the layer adds per-owner fp32 partials, the oracle sums a bag in one pass: equal up to the
        # rounding of a different fp32 association -> at most one bf16 ulp on a few elements
        a, b = ko.bf16_bits_to_f32(got), ko.bf16_bits_to_f32(exp)
        np.testing.assert_allclose(a, b, rtol=2.0 ** -7, atol=1e-6)
        assert (got == exp).mean() > 0.99, (got == exp).mean()
        # backward: dense [V, D] gradient of BOTH ranks' batches (rank 0's bags, then rank 1's: the order in
        # which the owner receives them), then the reference's Adagrad on the touched rows
        B2 = world * B
        f2 = ko.make_features(list(range(T)), ["sum"] * T, [t * D for t in range(T)], hots=HOTS, batch=B2)
        flat2 = np.concatenate([np.concatenate([ids[r][k] for r in range(world)], 0).reshape(-1) for k in feats])
        g2 = np.concatenate(gq, axis=0)
        dense = [np.zeros((V[t], D), np.float32) for t in range(T)]
        ko.embed_bag_bwd_dense(ko.make_tables(dense), f2, flat2, None, None, None, g2, B2, D)
        worst = 0.0
        for t in range(T):
            touched = np.zeros(V[t], np.uint8)
            touched[np.concatenate([ids[r][f"cat_{t:02d}_id"].reshape(-1) for r in range(world)])] = 1
            tab, acc = full[t].copy(), np.full((V[t], D), ACC0, np.float32)
            ko.apply_optimizer(tab, acc, dense[t], touched, LR, "adagrad")
            gott = to_np(after[f"cat_{t}"])
            same = (gott == tab).mean()
            assert same > 0.9999, (t, same)            # bf16 rows: bit-equal up to rare 1-ulp association effects
            np.testing.assert_allclose(ko.bf16_bits_to_f32(gott), ko.bf16_bits_to_f32(tab), rtol=2.0 ** -7, atol=1e-6)
            assert not np.array_equal(gott, full[t]) or V[t] < 4
            worst = max(worst, 1.0 - same)
        print("SHARDED_C3P_OK worst mismatch fraction %.2e" % worst)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
