"""Worker of tests/test_sharded_gloo.py: world_size-2 gloo run of the sharded embedding exchange on
CPU, with the compute kernels replaced by the CPU oracle (test infrastructure) so that only the
bucketise / all-to-all / permutation plumbing of keras_rs_amd/sharded.py is under test."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import krs_oracle as ko  # noqa: E402


class OracleShardKernels:
    """The kernel calls of keras_rs_amd.sharded on the CPU oracle (K6 route / unpack / combine, K1, K2)."""

    def route(self, desc, ids, offsets, weights, batch, n_shards, emit_w, err_flag=None):
        r = ko.shard_route(desc.view(ko.SHARD_FEATURE_DT), np.ascontiguousarray(ids.numpy()),
                           None if offsets is None else np.ascontiguousarray(offsets.numpy()),
                           None if weights is None else weights.numpy(), batch, n_shards, emit_w)
        self.flags = r["flags"]
        return dict(packed=torch.from_numpy(r["packed"]), seg_grow=torch.from_numpy(r["seg_grow"]),
                    bag_seg=torch.from_numpy(r["bag_seg"]), counts=torch.from_numpy(r["counts"].copy()))

    def unpack(self, packed, lookups, segments, weighted):
        rows, w, off = ko.shard_unpack(packed.numpy(), lookups, segments, weighted)
        return torch.from_numpy(rows.copy()), (None if w is None else torch.from_numpy(w.copy())), torch.from_numpy(off)

    def route_static(self, desc, ids, offsets, weights, batch, n_shards, emit_w, cap_l, cap_s, err_flag=None):
        r = ko.shard_route_static(desc.view(ko.SHARD_FEATURE_DT), np.ascontiguousarray(ids.numpy()),
                                  None if offsets is None else np.ascontiguousarray(offsets.numpy()),
                                  None if weights is None else weights.numpy(), batch, n_shards, emit_w, cap_l, cap_s)
        self.flags = r["flags"]
        return dict(packed=torch.from_numpy(r["packed"]), seg_grow=torch.from_numpy(r["seg_grow"]),
                    bag_seg=torch.from_numpy(r["bag_seg"]), counts=torch.from_numpy(r["counts"].copy()))

    def unpack_static(self, packed, cap_l, cap_s, weighted):
        rows, w, off, stats = ko.shard_unpack_static(packed.numpy(), cap_l, cap_s, weighted)
        return (torch.from_numpy(rows), None if w is None else torch.from_numpy(w), torch.from_numpy(off),
                torch.from_numpy(stats))

    def combine(self, partials, bag_seg, batch, n_feats, dim, out):
        res = ko.shard_combine(np.ascontiguousarray(partials.numpy()), bag_seg.numpy(), batch, n_feats, dim)
        out.copy_(torch.from_numpy(res))
        return out

    def gather_rows(self, table, rows):
        t = table.detach().numpy()
        return torch.from_numpy(t[rows.numpy()].copy()) if rows.numel() else torch.zeros((0, t.shape[1]))

    def pool_segments(self, table, rows, offsets, weights, out_dtype):
        t = np.ascontiguousarray(table.detach().numpy())
        n_seg = offsets.numel() - 1
        out = np.zeros((max(n_seg, 1), t.shape[1]), np.float32)
        if n_seg:
            f = ko.make_features([0], ["sum"], [0])
            ko.embed_bag_fwd_raw(ko.make_tables([t]), ko.F32, f, np.ascontiguousarray(rows.numpy()), offsets.numpy(),
                                 None if weights is None else np.ascontiguousarray(weights.numpy()), n_seg, t.shape[1], out)
        return torch.from_numpy(out[:n_seg])

    def apply_segments(self, table, slot, rows, offsets, weights, seg_grads, lr, kind, hyper=None, grad_scale=1.0):
        t = table.numpy()
        n_seg = offsets.numel() - 1
        if rows.numel() == 0 or n_seg == 0:
            return
        dense = np.zeros_like(t)
        f = ko.make_features([0], ["sum"], [0])
        r = np.ascontiguousarray(rows.numpy())
        ko.embed_bag_bwd_dense(ko.make_tables([dense]), f, r, offsets.numpy(),
                               None if weights is None else np.ascontiguousarray(weights.numpy()), None,
                               np.ascontiguousarray(seg_grads.numpy() * np.float32(grad_scale)), n_seg, t.shape[1])
        touched = np.zeros(t.shape[0], np.uint8)
        touched[r[: int(offsets[-1])]] = 1     # (the static form pads `rows` behind the last segment with -1)
        ko.apply_optimizer(t, None if slot is None else slot.numpy(), dense, touched, lr, kind, hyper)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import keras_rs_amd.layers as kl
    from keras_rs_amd.sharded import ShardedDistributedEmbedding

    kind = sys.argv[1]
    use_w = len(sys.argv) < 3 or sys.argv[2] in ("w", "wragged")
    ragged = len(sys.argv) >= 3 and sys.argv[2].endswith("ragged")
    opt = {"sgd": kl.SGD(0.1), "adagrad": kl.Adagrad(0.1, 0.1), "adam": kl.Adam(0.1, 0.9, 0.999, 1e-7),
           "ftrl": kl.Ftrl(0.1, -0.5, 0.1, 0.01, 0.02, 0.3)}[kind]
    V, D, B = [37, 10, 64], 8, 6
    mode = sys.argv[3] if len(sys.argv) > 3 else "exact"
    if mode == "static_shrink":
        B = 600           # enough lookups for the settled capacity to sit below the generous first guess
    combs = ["sum", "mean", "sqrtn", "sum"]
    tcs = [kl.TableConfig(f"t{i}", V[i], D, optimizer=opt, combiner="sum", placement="sparsecore") for i in range(3)]
    feats = {}
    hots = [1, 3, 2, 4]
    tix = [0, 1, 2, 0]  # feature 3 shares table 0
    for i in range(4):
        tc = tcs[tix[i]]
        # the combiner is a property of the table in the reference; give shared-table features the table's
        feats[f"f{i}"] = kl.FeatureConfig(f"f{i}", tc, (B, hots[i]), (B, D))
    for i, tc in enumerate(tcs):
        tc.combiner = combs[i]
    exchange = sys.argv[3] if len(sys.argv) > 3 else "exact"
    kw = {}
    if exchange == "static_tiny":
        # a capacity the first steps overflow: lookups are dropped (flag raised), the running statistics grow the
        # capacity on every rank at the same step, and from then on the result is the exact one
        exchange, kw = "static", dict(capacity=(4, 4))
    elif exchange == "static_cfg":
        exchange, kw = "static", dict(capacity="table_config")
    elif exchange == "static_prefetch":
        exchange = "static"
    elif exchange == "static_shrink":
        exchange, kw = "static", dict(capacity_headroom=4.0, capacity_settle_steps=3)
    # (static_prefetch: a lead of whole feature slots on ranks > 0, so that the slab-gradient path of the backward runs)
    lead = D * rank if mode == "static_prefetch" else 3 * rank
    layer = ShardedDistributedEmbedding(feats, kernels=OracleShardKernels(), device="cpu", slab_lead_cols=lead,
                                        exchange=exchange, **kw)
    rng = np.random.default_rng(7)
    full = {f"t{i}": rng.uniform(-1, 1, (V[i], D)).astype(np.float32) for i in range(3)}
    layer.set_embedding_tables(full)
    got_tables = layer.get_embedding_tables()
    for k in full:
        np.testing.assert_array_equal(got_tables[k].numpy(), full[k])  # shard / unshard round trip

    rng_r = np.random.default_rng(100 + rank)  # every rank has its own batch
    ids = {f"f{i}": rng_r.integers(0, V[tix[i]], (B, hots[i])).astype(np.int32) for i in range(4)}
    w = {f"f{i}": rng_r.uniform(0.1, 1, (B, hots[i])).astype(np.float32) if use_w
         else np.ones((B, hots[i]), np.float32) for i in range(4)}
    if ragged:
        # ragged bags (some empty): the reference's form is pad-to-dense with weight 0 (base:31-92), which is
        # what the unsharded oracle below computes from the masked weights
        from keras_rs_amd.layers.embed_reduce import Ragged

        keep = {k: np.arange(v.shape[1])[None, :] < rng_r.integers(0, v.shape[1] + 1, (B, 1)) for k, v in ids.items()}
        r_ids = {k: Ragged.from_rows([row[m] for row, m in zip(ids[k], keep[k])]) for k in ids}
        r_w = {k: Ragged.from_rows([row[m] for row, m in zip(w[k], keep[k])], dtype=np.float32) for k in ids}
        out = layer(r_ids, r_w if use_w else None)
        w = {k: w[k] * keep[k] for k in ids}
    else:
        keep = {k: np.ones(v.shape, bool) for k, v in ids.items()}
        if kw.get("capacity") == (4, 4):
            with torch.no_grad():
                for _ in range(3):      # steps 0-2 overflow; step 0's statistics are read at step 2
                    layer(ids, w if use_w else None)
                    assert layer.kernels.flags & ko.FLAG_CAPACITY_OVERFLOW
            assert layer.overflow_steps >= 1 and min(next(iter(layer._caps.values()))) >= 8
        if mode == "static_shrink":
            cap0 = list(next(iter(layer._caps.values()))) if layer._caps else None
            with torch.no_grad():
                for _ in range(7):      # statistics are read two steps late; three fitting steps in a row shrink
                    layer(ids, w if use_w else None)
            cap1 = list(next(iter(layer._caps.values())))
            assert layer.capacity_shrinks >= 1 and layer.overflow_steps == 0, (layer.capacity_shrinks, cap1)
            need = layer.last_exchange["need"]
            assert need[0] <= cap1[0] < 0.8 * 4.0 * (B * sum(hots) / world) and need[1] <= cap1[1], (cap0, cap1, need)
            # a HEAVIER batch behind the shrink (every id on owner 0): its lookups beyond the shrunken blocks are dropped
            # and counted for the two steps the statistics take to react, then the capacity has grown to the need on every
            # rank and nothing is dropped any more (why shrinking is opt-in: ADVICE r4)
            skew = {k: (v // world) * world for k, v in ids.items()}
            with torch.no_grad():
                for _ in range(4):
                    layer(skew, w if use_w else None)
                layer.flush_exchange_stats()
            cap2 = list(next(iter(layer._caps.values())))
            assert layer.overflow_steps >= 1 and cap2[0] > cap1[0], (cap1, cap2, layer.overflow_steps)
            assert layer.last_exchange["need"][0] <= cap2[0] and layer.last_exchange["need"][1] <= cap2[1]
            layer.kernels.flags = 0
            with torch.no_grad():
                layer(skew, w if use_w else None)
            assert not layer.kernels.flags & ko.FLAG_CAPACITY_OVERFLOW
            layer.capacity_settle_steps = 0      # (the checked call below runs on the grown blocks)
        if mode == "static_prefetch":
            # the id side of the call (route -> id all-to-all -> unpack) issued ahead of it: same result, one hit
            pre = layer.preprocess(ids, w if use_w else None)
            layer.prefetch(pre)
            out = layer(pre)
            assert layer.prefetch_hits == 1
        else:
            out = layer(ids, w if use_w else None)
        if exchange == "static":
            assert layer.last_exchange["mode"] == "static" and not layer.kernels.flags & ko.FLAG_CAPACITY_OVERFLOW
    g = {k: torch.from_numpy(rng_r.uniform(0, 1, (B, D)).astype(np.float32)) for k in out}
    if mode == "static_prefetch":
        # the gradient arrives as ONE matrix for the whole slab (layers.concat_features hands the slab itself on):
        # ranks with a lead of whole feature slots gather the segment gradients straight out of it
        import keras_rs_amd.layers as kl2

        head = torch.zeros((B, lead), requires_grad=True)
        cat = kl2.concat_features(([head] if lead else []) + [out[f"f{i}"] for i in range(4)])
        gm = torch.cat(([torch.zeros(B, lead)] if lead else []) + [g[f"f{i}"] for i in range(4)], dim=1)
        (cat * gm).sum().backward()
        assert layer.slab_grad_gathers == (1 if lead else 0)
    else:
        sum((o * g[k]).sum() for k, o in out.items()).backward()

    # unsharded oracle: forward per rank, table update from the contributions of ALL ranks
    gathered = [None] * world
    dist.all_gather_object(gathered, (ids, w, {k: v.numpy() for k, v in g.items()}, keep))
    for i in range(4):
        comb = tcs[tix[i]].combiner
        exp = ko.embed_reduce(full[f"t{tix[i]}"], ids[f"f{i}"], w[f"f{i}"], comb)
        np.testing.assert_allclose(out[f"f{i}"].detach().numpy(), exp, rtol=1e-6, atol=1e-6)
    dense = {k: np.zeros_like(v) for k, v in full.items()}
    touched = {k: np.zeros(v.shape[0], np.uint8) for k, v in full.items()}
    for r_ids, r_w, r_g, r_keep in gathered:
        for i in range(4):
            comb = tcs[tix[i]].combiner
            tabs = ko.make_tables([dense[f"t{tix[i]}"]])
            f = ko.make_features([0], [comb], [0], hots=[hots[i]], batch=B)
            scale = np.zeros(B, np.float32)
            tmp = np.zeros((B, D), np.float32)
            ko.embed_bag_fwd_raw(ko.make_tables([full[f"t{tix[i]}"]]), ko.F32, f, r_ids[f"f{i}"].reshape(-1), None,
                                 r_w[f"f{i}"].reshape(-1), B, D, tmp, scale)
            ko.embed_bag_bwd_dense(tabs, f, r_ids[f"f{i}"].reshape(-1), None, r_w[f"f{i}"].reshape(-1), scale,
                                   r_g[f"f{i}"], B, D)
            touched[f"t{tix[i]}"][r_ids[f"f{i}"][r_keep[f"f{i}"]]] = 1
    after = layer.get_embedding_tables()
    for k in full:
        exp = full[k].copy()
        acc = np.full_like(exp, 0.1)
        hyper = None
        if kind in ("adam", "ftrl"):
            acc = np.zeros((2,) + exp.shape, np.float32)
            if kind == "ftrl":
                acc[0] = 0.1
            hyper = (0.9, 0.999, 1e-7, float(np.sqrt(1 - 0.999) / (1 - 0.9))) if kind == "adam" else \
                (-0.5, 0.01, 0.02, 0.3)
        ko.apply_optimizer(exp, acc, dense[k], touched[k], 0.1, kind, hyper)
        np.testing.assert_allclose(after[k].numpy(), exp, rtol=1e-5, atol=1e-6)
    if rank == 0:
        print("SHARDED_OK", kind)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
