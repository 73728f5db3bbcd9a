"""Worker of tests/test_sharded_gpu.py: world_size-2 run of the sharded embedding with the REAL HIP
kernels, both ranks on the one GPU of the test box, collectives over gloo (staged through the host).
Checks outputs and updated tables against the single-GPU DistributedEmbedding fed the same data."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import keras_rs_amd.layers as kl
    from keras_rs_amd.sharded import ShardedDistributedEmbedding

    kind = sys.argv[1]
    opt = {"sgd": kl.SGD(0.1), "adagrad": kl.Adagrad(0.1, 0.1), "adam": kl.Adam(0.1)}[kind]
    V, D, B = [501, 37, 1000], 32, 96
    hots = [1, 7, 3, 12]
    tix = [0, 1, 2, 0]
    combs = ["sum", "mean", "sqrtn"]

    def configs():
        tcs = [kl.TableConfig(f"t{i}", V[i], D, optimizer=opt, combiner=combs[i], placement="sparsecore") for i in range(3)]
        return {f"f{i}": kl.FeatureConfig(f"f{i}", tcs[tix[i]], (B, hots[i]), (B, D)) for i in range(4)}

    rng = np.random.default_rng(3)
    full = {f"t{i}": rng.uniform(-1, 1, (V[i], D)).astype(np.float32) for i in range(3)}
    # every rank's batch, known to all (the reference layer below consumes the concatenation)
    all_ids = [{f"f{i}": np.random.default_rng(50 + r * 10 + i).integers(0, V[tix[i]], (B, hots[i])).astype(np.int32)
                for i in range(4)} for r in range(world)]
    all_w = [{f"f{i}": np.random.default_rng(90 + r * 10 + i).uniform(0.1, 1, (B, hots[i])).astype(np.float32)
              for i in range(4)} for r in range(world)]
    all_g = [{f"f{i}": np.random.default_rng(130 + r * 10 + i).uniform(0, 1, (B, D)).astype(np.float32)
              for i in range(4)} for r in range(world)]

    layer = ShardedDistributedEmbedding(configs(), slab_lead_cols=8)
    layer.build(None)
    layer.set_embedding_tables(full)
    out = layer(all_ids[rank], all_w[rank])
    sum((o * torch.from_numpy(all_g[rank][k]).cuda()).sum() for k, o in out.items()).backward()
    torch.cuda.synchronize()
    got_tables = {k: v.cpu().numpy() for k, v in layer.get_embedding_tables().items()}

    # single-GPU layer on the concatenated batch
    ref = kl.DistributedEmbedding({k: kl.FeatureConfig(k, fc.table, (world * B, fc.input_shape[1]), (world * B, D))
                                   for k, fc in configs().items()})
    ref.build(None)
    ref.set_embedding_tables(full)
    cat = lambda xs, k: np.concatenate([x[k] for x in xs], 0)  # noqa: E731
    rout = ref({k: cat(all_ids, k) for k in out}, {k: cat(all_w, k) for k in out})
    sum((o * torch.from_numpy(cat(all_g, k)).cuda()).sum() for k, o in rout.items()).backward()
    for k in out:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), rout[k].detach().cpu().numpy()[rank * B:(rank + 1) * B],
                                   rtol=1e-5, atol=1e-5)
    for k, v in ref.get_embedding_tables().items():
        np.testing.assert_allclose(got_tables[k], v.cpu().numpy(), rtol=2e-5, atol=2e-6)
    dist.barrier()
    if rank == 0:
        print("SHARDED_HIP_OK", kind)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
