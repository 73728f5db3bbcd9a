"""LDS bank-conflict estimator for the GEMM tile layouts (rules: MI355X_MICROARCH.md, LDS section)."""
import itertools

B128_READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
                    [x + 32 for x in list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))],
                    [x + 32 for x in list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]]


def cost(addrs, nbytes, groups, nbanks):
    """addrs[lane] = byte address; returns total LDS cycles (1 per group when conflict-free)."""
    tot = 0
    for g in groups:
        per_bank = {}
        for l in g:
            for d in range(nbytes // 4):
                a = addrs[l] + 4 * d
                per_bank.setdefault((a // 4) % nbanks, set()).add(a // 4)
        tot += max(len(v) for v in per_bank.values())
    return tot


def contiguous_groups(n):
    return [list(range(i, i + n)) for i in range(0, 64, n)]


def layout(row, chunk, stride, swz):
    return row * stride + ((chunk ^ swz(row)) * 16)


for stride in (128, 144, 160, 176):
    for name, swz in (("none", lambda r: 0), ("r>>3", lambda r: (r >> 3) & 7), ("r&7", lambda r: r & 7),
                      ("r>>1", lambda r: (r >> 1) & 7), ("r>>2", lambda r: (r >> 2) & 7)):
        # fragment read: lanes 0-31 rows 0..31 chunk c0, lanes 32-63 chunk c0+1
        rd = max(cost([layout(l & 31, c0 + (l >> 5), stride, swz) for l in range(64)], 16, B128_READ_GROUPS, 64)
                 for c0 in (0, 2, 4, 6))
        # kcontig store b128: lane t (of a wave w=0): c = t&7, row = t>>3 (+32 i)
        st = cost([layout(t >> 3, t & 7, stride, swz) for t in range(64)], 16, contiguous_groups(8), 32)
        # bf16 transposed store b64: wave covers rb = 0..15, kb = 4w..4w+3 ; row = rb*8+j ; chunk = kb>>1 ; +8*(kb&1)
        tb = max(cost([layout((t % 16) * 8 + j, (t // 16) >> 1, stride, swz) + 8 * ((t // 16) & 1) for t in range(64)],
                      8, contiguous_groups(16), 32) for j in range(8))
        # f32 transposed store b128: wave covers rb = 0..31, kb = 2w, 2w+1 ; row = rb*4+j ; chunk = kb
        tf = max(cost([layout((t % 32) * 4 + j, t // 32, stride, swz) for t in range(64)], 16, contiguous_groups(8), 32)
                 for j in range(4))
        print(f"stride {stride:4d} swz {name:5s}: frag_read {rd:3d} (min 4)  kcontig_st {st:3d} (min 8)  "
              f"bf16_T_st {tb:3d} (min 4)  f32_T_st {tf:3d} (min 8)")
