"""C5 scale (BASELINE.json configs[4]: Criteo-1TB vocabularies, largest 40 M rows, power-law ids, bf16
tables with fp32 Adagrad accumulators): tables whose byte size passes 4 GiB, so every row offset in the
gather (K1) and in the fused row update (K2) has to be 64-bit, next to a 3-row table whose rows collect
tens of thousands of lookups each (the chunked hot-row path).  The oracle cannot hold these sizes; the
checks are exact gathers, plain-torch compositions on compact row sets and whole-table checksums."""

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D, B = 128, 65536
VOCABS = [40_000_000, 3, 40_000_000]          # 10.24 GB per big table, 20.5 GB per fp32 accumulator
HOTS = [8, 2, 1]
LR, ACC0 = 0.0034, 0.1
MULT = 7_368_787                               # odd, not a multiple of 5: coprime to 40 M = 2^9 * 5^7


def power_law_ids(n, vocab, gen):
    """id = floor(V * u^4) (1 % of the rows draw 32 % of the lookups), spread over the whole row range
    by a fixed affine permutation -- SURVEY.md section 8d, C5."""
    u = torch.rand(n, device=DEV, generator=gen, dtype=torch.float64)
    r = (u.pow(4) * vocab).long().clamp_(max=vocab - 1)
    return ((r * MULT + 12345) % vocab).to(torch.int32) if vocab > 1000 else r.to(torch.int32)


def checksum(t):
    return t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32).sum(dtype=torch.int64)


@pytest.fixture(scope="module")
def c5():
    from keras_rs_amd.embedding_ops import FusedBags

    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip("needs 90 GB of free HBM")
    gen = torch.Generator(device=DEV).manual_seed(1337)
    tables = [torch.empty(v, D, device=DEV, dtype=torch.bfloat16).uniform_(-0.05, 0.05, generator=gen) for v in VOCABS]
    slots = [torch.full((v, D), ACC0, device=DEV, dtype=torch.float32) for v in VOCABS]
    fb = FusedBags(tables, [(t, "sum", t * D) for t in range(3)], slots=slots, lrs=[LR] * 3)
    yield tables, slots, fb, gen
    del fb, tables, slots
    torch.cuda.empty_cache()


def test_rows_past_4_gib_gather_exactly(c5):
    tables, _, fb, gen = c5
    ones = [1, 1, 1]
    ids = [power_law_ids(B, v, gen) for v in VOCABS]
    ids[0][:4096] = torch.randint(VOCABS[0] - 4096, VOCABS[0], (4096,), device=DEV, generator=gen, dtype=torch.int32)
    ids[2][-1] = VOCABS[2] - 1                                      # the very last row of the last table
    assert int(ids[0].max()) * D * 2 > 2 ** 33                       # byte offsets well past 32 bits
    out, _ = fb.forward(torch.cat(ids), B, hots=ones)
    for t in range(3):
        assert torch.equal(out[:, t * D:(t + 1) * D], tables[t][ids[t].long()])


def test_power_law_bags_match_torch_on_a_slice(c5):
    tables, _, fb, gen = c5
    ids = torch.cat([power_law_ids(B * h, v, gen) for h, v in zip(HOTS, VOCABS)])
    w = torch.rand(ids.numel(), device=DEV, generator=gen)
    out, _ = fb.forward(ids, B, hots=HOTS, weights=w, out_dtype=torch.float32)
    base = 0
    for t, h in enumerate(HOTS):
        sl = ids[base: base + 1024 * h].reshape(1024, h).long()
        ws = w[base: base + 1024 * h].reshape(1024, h)
        ref = (tables[t][sl].float() * ws[..., None]).sum(1)
        torch.testing.assert_close(out[:1024, t * D:(t + 1) * D], ref, rtol=1e-5, atol=1e-5)
        base += B * h


def test_fused_adagrad_touches_exactly_the_looked_up_rows(c5):
    tables, slots, fb, gen = c5
    ids = [power_law_ids(B * h, v, gen) for h, v in zip(HOTS, VOCABS)]
    ids[0][:8] = VOCABS[0] - 1                                       # last row: table offset 10.24 GB, slot 20.5 GB
    flat = torch.cat(ids)
    grad = (torch.rand(B, 3 * D, device=DEV, generator=gen) - 0.5).to(torch.bfloat16)
    sums_before = [(checksum(t), checksum(s)) for t, s in zip(tables, slots)]
    uniq, rows_before, expect = [], [], []
    for t, h in enumerate(HOTS):
        u, inv = torch.unique(ids[t].long(), return_inverse=True)
        g = grad[:, t * D:(t + 1) * D].float().repeat_interleave(h, 0)        # bag-major lookups of this feature
        gsum = torch.zeros(u.numel(), D, device=DEV, dtype=torch.float64).index_add_(0, inv, g.double()).float()
        acc = ACC0 + gsum * gsum                                               # jax/test_utils.py:474-497
        w0 = tables[t][u]
        expect.append(((w0.float() - LR * gsum / acc.sqrt()).to(torch.bfloat16), acc))
        uniq.append(u)
        rows_before.append((w0.clone(), slots[t][u].clone()))
    ws = fb.plan_backward(flat, B, hots=HOTS, global_order=False)
    fb.backward_fused("adagrad", ws, grad, B, flat.numel(), hots=HOTS)
    for t in range(3):
        w1, a1 = tables[t][uniq[t]], slots[t][uniq[t]]
        # acc = acc0 + (sum g)^2 with the sum taken in fp32 over up to ~44 k lookups per row (kernel: chunks of
        # 2048 in sorted order; here: float64), so the relative error doubles where the sum is small
        torch.testing.assert_close(a1, expect[t][1], rtol=2e-4, atol=1e-5)
        # fp32 sums in a different order, then one bf16 rounding: at most one ulp (2^-12 below 0.0625) apart
        torch.testing.assert_close(w1.float(), expect[t][0].float(), rtol=0, atol=2 ** -11)
        assert (w1 == expect[t][0]).float().mean() > 0.99
        # whole-table checksums moved by exactly what the touched rows moved: no other row was written
        dt = checksum(tables[t]) - sums_before[t][0]
        ds = checksum(slots[t]) - sums_before[t][1]
        assert int(dt) == int(checksum(w1) - checksum(rows_before[t][0]))
        assert int(ds) == int(checksum(a1) - checksum(rows_before[t][1]))
    assert float((tables[1][:3].float() - rows_before[1][0].float()).abs().max()) > 0         # the hot rows did move
