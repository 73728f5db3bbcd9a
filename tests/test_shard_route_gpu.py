"""K6 (krs_shard_route / krs_shard_unpack / krs_shard_combine, csrc/shard_route.hip) against the oracle's
sequential restatement (oracle/krs_oracle.c: krs_oracle_shard_*): integer outputs bit for bit, the fp32
per-lookup weights bit for bit (same summation order), the combine to one rounding."""

import numpy as np
import pytest
import torch

from oracle import krs_oracle as ko

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _desc(n_shards, batch, hots, vocabs, combs, local_off):
    d = np.zeros(len(vocabs), dtype=ko.SHARD_FEATURE_DT)
    pos = 0
    for i, v in enumerate(vocabs):
        hot = 0 if hots is None else hots[i]
        d[i] = (pos, local_off[i] * n_shards, hot, {"sum": 0, "mean": 1, "sqrtn": 2}[combs[i]], v, 0)
        pos += batch * hot
    return d


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
@pytest.mark.parametrize("csr", [False, True])
@pytest.mark.parametrize("weighted,id64,bad", [(False, False, False), (True, True, False), (True, False, True)])
def test_route_unpack_match_oracle(n_shards, csr, weighted, id64, bad):
    from keras_rs_amd.sharded import HipShardKernels

    rng = np.random.default_rng(n_shards * 7 + csr * 3 + weighted)
    batch = 301
    vocabs = [3, 1000, 57, 40_000, 9]
    combs = ["sum", "mean", "sqrtn", "mean", "sum"] if weighted else ["sum"] * 5
    local_off, off = [], 0
    for v in vocabs:
        local_off.append(off)
        off += -(-v // n_shards)
    if csr:
        lens = rng.integers(0, 7, size=(len(vocabs), batch))
        lens[3] = rng.integers(0, 60, size=batch)                      # long bags: segments span blocks
        offsets = np.concatenate([[0], np.cumsum(lens.reshape(-1))]).astype(np.int64 if id64 else np.int32)
        ids = np.concatenate([rng.integers(0, vocabs[f], int(lens[f].sum())) for f in range(len(vocabs))])
        hots = None
    else:
        hots = [2, 1, 5, 33, 3]
        ids = np.concatenate([rng.integers(0, vocabs[f], batch * hots[f]) for f in range(len(vocabs))])
        offsets = None
    ids = ids.astype(np.int64 if id64 else np.int32)
    if bad:
        ids[[5, len(ids) // 2, len(ids) - 1]] = [-1, 10 ** 6, 40_000]
    w = rng.uniform(-1, 1, len(ids)).astype(np.float32) if weighted else None
    if weighted:
        w[::17] = 0.0
    emit_w = weighted or any(c != "sum" for c in combs)
    desc = _desc(n_shards, batch, hots, vocabs, combs, local_off)
    exp = ko.shard_route(desc, ids, offsets, w, batch, n_shards, emit_w)
    assert bool(exp["flags"] & 1) == bad

    k = HipShardKernels()
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = k.route(desc, torch.from_numpy(ids).to(DEV),
                  None if offsets is None else torch.from_numpy(offsets).to(DEV),
                  None if w is None else torch.from_numpy(w).to(DEV), batch, n_shards, emit_w, err)
    counts = got["counts"].cpu().numpy()
    np.testing.assert_array_equal(counts, exp["counts"])
    assert bool(int(err.item()) & 1) == bad
    words, n_seg = int(counts[2].sum()), int(counts[1].sum())
    np.testing.assert_array_equal(got["packed"][:words].cpu().numpy(), exp["packed"][:words])
    np.testing.assert_array_equal(got["seg_grow"][:n_seg].cpu().numpy(), exp["seg_grow"][:n_seg])
    np.testing.assert_array_equal(got["bag_seg"].cpu().numpy(), exp["bag_seg"])

    # owner side: this rank receives, from "sources", the blocks it packed (a world-of-one exchange per owner)
    rows, wv, off_ = k.unpack(got["packed"][:words], counts[0].tolist(), counts[1].tolist(), emit_w)
    e_rows, e_w, e_off = ko.shard_unpack(exp["packed"][:words], counts[0], counts[1], emit_w)
    np.testing.assert_array_equal(rows.cpu().numpy(), e_rows)
    np.testing.assert_array_equal(off_.cpu().numpy(), e_off)
    if emit_w:
        np.testing.assert_array_equal(wv.cpu().numpy().view(np.uint32), e_w.view(np.uint32))
    assert int(off_[-1]) == int(counts[0].sum())


@pytest.mark.parametrize("n_shards", [1, 2, 5, 8])
@pytest.mark.parametrize("csr,weighted", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("fit", ["roomy", "tight_lookups", "tight_segments"])
def test_static_route_unpack_match_oracle(n_shards, csr, weighted, fit):
    """Static-capacity form (krs_shard_route_static / krs_shard_unpack_static) bit for bit against the oracle, with
    room to spare (nothing dropped: the kept lookups ARE the exact form's) and with capacities that drop lookups /
    segments (the reference's allow_id_dropping contract: what is dropped and the overflow flag must agree)."""
    from keras_rs_amd.sharded import HipShardKernels

    rng = np.random.default_rng(n_shards * 11 + csr * 5 + weighted)
    batch = 257
    vocabs = [5, 2000, 91, 30_000]
    combs = ["sum", "mean", "sqrtn", "mean"] if weighted else ["sum"] * 4
    local_off, off = [], 0
    for v in vocabs:
        local_off.append(off)
        off += -(-v // n_shards)
    if csr:
        lens = rng.integers(0, 9, size=(len(vocabs), batch))
        offsets = np.concatenate([[0], np.cumsum(lens.reshape(-1))]).astype(np.int32)
        ids = np.concatenate([rng.integers(0, vocabs[f], int(lens[f].sum())) for f in range(len(vocabs))])
        hots = None
    else:
        hots = [2, 1, 6, 21]
        ids = np.concatenate([rng.integers(0, vocabs[f], batch * hots[f]) for f in range(len(vocabs))])
        offsets = None
    ids = ids.astype(np.int32)
    w = rng.uniform(-1, 1, len(ids)).astype(np.float32) if weighted else None
    emit_w = weighted
    desc = _desc(n_shards, batch, hots, vocabs, combs, local_off)
    exact = ko.shard_route(desc, ids, offsets, w, batch, n_shards, emit_w)
    need_l, need_s = int(exact["counts"][0].max()), int(exact["counts"][1].max())
    up4 = lambda v: -(-int(v) // 4) * 4  # noqa: E731
    cap_l, cap_s = {"roomy": (up4(need_l * 1.2 + 4), up4(need_s * 1.2 + 4)),
                    "tight_lookups": (max(4, up4(need_l * 0.7) - 4), up4(need_s + 4)),
                    "tight_segments": (up4(need_l + 4), max(4, up4(need_s * 0.6) - 4))}[fit]
    exp = ko.shard_route_static(desc, ids, offsets, w, batch, n_shards, emit_w, cap_l, cap_s)
    assert bool(exp["flags"] & ko.FLAG_CAPACITY_OVERFLOW) == (fit != "roomy")

    k = HipShardKernels()
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = k.route_static(desc, torch.from_numpy(ids).to(DEV), None if offsets is None else torch.from_numpy(offsets).to(DEV),
                         None if w is None else torch.from_numpy(w).to(DEV), batch, n_shards, emit_w, cap_l, cap_s, err)
    assert int(err.item()) == exp["flags"]
    np.testing.assert_array_equal(got["counts"].cpu().numpy(), exp["counts"])
    np.testing.assert_array_equal(got["packed"].cpu().numpy(), exp["packed"])
    np.testing.assert_array_equal(got["seg_grow"].cpu().numpy(), exp["seg_grow"])
    np.testing.assert_array_equal(got["bag_seg"].cpu().numpy(), exp["bag_seg"])

    rows, wv, off_, stats = k.unpack_static(got["packed"], cap_l, cap_s, emit_w)
    e_rows, e_w, e_off, e_stats = ko.shard_unpack_static(exp["packed"], cap_l, cap_s, emit_w)
    np.testing.assert_array_equal(rows.cpu().numpy(), e_rows)
    np.testing.assert_array_equal(off_.cpu().numpy(), e_off)
    np.testing.assert_array_equal(stats.cpu().numpy(), e_stats)
    if emit_w:
        np.testing.assert_array_equal(wv.cpu().numpy().view(np.uint32), e_w.view(np.uint32))
    if fit == "roomy":
        # nothing dropped: the compact rows / weights / segment lengths are the exact form's
        counts = exact["counts"]
        words = int(counts[2].sum())
        x_rows, x_w, x_off = ko.shard_unpack(exact["packed"][:words], counts[0], counts[1], emit_w)
        n = len(x_rows)
        np.testing.assert_array_equal(e_rows[:n], x_rows)
        assert (e_rows[n:] == -1).all()
        lens = np.diff(e_off)
        np.testing.assert_array_equal(lens[lens > 0], np.diff(x_off)[np.diff(x_off) > 0])


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("dim", [128, 20])
def test_combine_matches_oracle(dtype, dim):
    from keras_rs_amd.sharded import HipShardKernels
    from tests.helpers import to_np

    rng = np.random.default_rng(3)
    batch, n_feats, n_shards, n_seg = 97, 4, 8, 1500
    bag_seg = np.full((batch * n_feats, n_shards), -1, np.int32)
    picks = rng.permutation(batch * n_feats * n_shards)[:n_seg]
    bag_seg.reshape(-1)[picks] = rng.permutation(n_seg).astype(np.int32)
    part = torch.from_numpy(rng.uniform(-1, 1, (n_seg, dim)).astype(np.float32)).to(DEV)
    part = part.to(torch.bfloat16) if dtype == "bfloat16" else part
    lead = 8
    slab = torch.zeros((batch, lead + n_feats * dim), dtype=part.dtype, device=DEV)
    HipShardKernels().combine(part, torch.from_numpy(bag_seg).to(DEV), batch, n_feats, dim, slab[:, lead:])
    exp = ko.shard_combine(to_np(part), bag_seg, batch, n_feats, dim)
    np.testing.assert_array_equal(to_np(slab[:, lead:]), exp)
    assert torch.count_nonzero(slab[:, :lead]) == 0
