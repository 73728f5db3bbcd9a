"""C4-shaped parity at size on ONE GPU (BASELINE.json configs[3]).

(a) `ShardedDistributedEmbedding` at world 1 on C3 itself -- 26 tables x 1,000,000 rows x 128 bf16, batch 65,536, the
    ml_perf bag lengths, partials in fp32 -- must equal the unsharded `DistributedEmbedding` BIT FOR BIT: the pooled
    outputs, and every table row and Adagrad accumulator after one fused update.  That pins route -> exchange ->
    owner-side pool -> combine, and gradient gather -> exchange -> fused K2 on the shard, at full size, in both
    exchange forms (exact sizes through the host / static capacity without a host wait).
(b) a world-2 run (two processes on this GPU, collectives over gloo) at the C3' vocabularies with B_local = 4096 against
    the ORACLE fed the concatenated batch (tests/_sharded_c3p_worker.py)."""

import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
HOTS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]


def _configs(kl, base, T, V, D, B, hots):
    opt = kl.Adagrad(learning_rate=0.0034, initial_accumulator_value=0.1)
    feats = {}
    for t in range(T):
        tc = kl.TableConfig(name=f"cat_{t}", vocabulary_size=V, embedding_dim=D,
                            initializer=base.RandomUniform(-0.05, 0.05, seed=1337 + t, device_rng=True),
                            optimizer=opt, combiner="sum", placement="sparsecore")
        feats[f"cat_{t:02d}_id"] = kl.FeatureConfig(f"cat_{t}", tc, (B, hots[t]), (B, D))
    return feats


@pytest.mark.parametrize("exchange", ["exact", "static"])
def test_world1_sharded_equals_unsharded_bit_for_bit_at_c3(exchange):
    import keras_rs_amd.layers as kl
    from keras_rs_amd.layers import base
    from keras_rs_amd.sharded import ShardedDistributedEmbedding

    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free HBM")
    T, V, D, B = 26, 1_000_000, 128, 65536
    ref = kl.DistributedEmbedding(_configs(kl, base, T, V, D, B, HOTS), dtype="bfloat16", slab_lead_cols=D)
    ref.build(None)
    sh = ShardedDistributedEmbedding(_configs(kl, base, T, V, D, B, HOTS), dtype="bfloat16", slab_lead_cols=D,
                                     partial_dtype="float32", exchange=exchange)
    sh.build(None)
    g = sh._sgroups[0]
    tables = ref.get_embedding_tables()        # views of the live parameters
    before0 = tables["cat_0"].clone()
    with torch.no_grad():
        for t, tc in enumerate(g.table_configs):          # world 1: the stacked shard is the tables one after the other
            sh.shard.data[g.row_off[t]: g.row_off[t + 1]].copy_(tables[tc.name])
    gen = torch.Generator(device=DEV).manual_seed(1338)
    ids = {f"cat_{t:02d}_id": torch.randint(0, V, (B, HOTS[t]), device=DEV, generator=gen, dtype=torch.int32)
           for t in range(T)}
    grad = (torch.rand(B, (T + 1) * D, device=DEV, generator=gen) - 0.5).to(torch.bfloat16)

    def run(layer):
        out = layer(layer.preprocess(ids))
        first = next(iter(out.values()))
        slab = first._krs_slab[0] if hasattr(first, "_krs_slab") else None
        views = [out[k] for k in out]
        torch.autograd.backward(views, [grad[:, (i + 1) * D:(i + 2) * D] for i in range(len(views))])
        return [v.detach() for v in views], slab

    out_ref, _ = run(ref)
    out_sh, _ = run(sh)
    torch.cuda.synchronize()
    if exchange == "static":
        assert sh.last_exchange["mode"] == "static"
    for a, b in zip(out_ref, out_sh):
        assert torch.equal(a, b)
    after = ref.get_embedding_tables()
    sh.check_ids(wait=True)
    slot_sh = sh._slot(g)
    assert not torch.equal(after["cat_0"], before0)                       # the update ran
    for t, tc in enumerate(g.table_configs):
        r0, r1 = g.row_off[t], g.row_off[t + 1]
        assert torch.equal(after[tc.name], sh.shard.data[r0:r1]), tc.name
        slot_ref = ref._table_slots[id(ref._groups["sparsecore"][0].table_configs[t])]
        assert torch.equal(slot_ref, slot_sh[r0:r1]), tc.name


def test_world2_at_c3prime_vocabularies_matches_the_oracle():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_sharded_c3p_worker.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0 and "SHARDED_C3P_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
