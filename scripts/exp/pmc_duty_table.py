"""gemm_pmc.txt (scripts/exp/pmc_gemm_r6.sh: per-kernel counter averages of separate rocprofv3 --pmc passes over
scripts/exp/gemm_bench) -> a markdown table of MFMA duty cycle, effective clock and the wave-time buckets per kernel."""
import collections
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    parts = line.split()
    if len(parts) < 4 or not parts[0].startswith("krs::"):
        continue
    value, count, counter = float(parts[-1]), int(parts[-2]), parts[-3]
    name = " ".join(parts[:-3])
    rows[name].setdefault(counter, value)          # (GRBM_GUI_ACTIVE is in two passes: the first one wins)
times = {}
if len(sys.argv) > 2:      # kernel-trace table of the same binary: average duration per kernel
    for line in open(sys.argv[2]):
        c = [x.strip() for x in line.split("|")]
        if len(c) > 5 and c[1].startswith("krs::"):
            times[c[1]] = float(c[4])
ROLE = {"gemm_pp64_kernel<0>": "h = x U, dh = dz K^T (64-k ring)", "gemm_pp64_kernel<1>": "y = cross(h K) (64-k ring)",
        "gemm_pp64_kernel<2>": "dx = dh U^T + g (64-k ring)", "gemm_pp64_kernel<4>": "krs_gemm_cross_bwd (64-k ring)",
        "gemm_pp256_kernel<true, 4, 0>": "weight gradients dK, dU (32-k ring, K-strided)",
        "gemm_pp256_kernel<false, 4, 0>": "h, dh on the 32-k ring (pipeline 5)", "gemm_pp256_kernel<false, 4, 1>": "cross on the 32-k ring (pipeline 5)",
        "gemm_pp256_kernel<false, 4, 2>": "dx on the 32-k ring (pipeline 5)", "gemm_pp256_kernel<false, 4, 4>": "fused backward on the 32-k ring (pipeline 5)"}
print("| kernel | product | avg us (trace) | GRBM_GUI_ACTIVE / 8 XCDs (cycles) | effective clock GHz | MFMA busy cycles per SIMD | "
      "MFMA busy % of active | wave cycles: MFMA-issue-stall / waitcnt+barrier / active (%) | LDS bank-conflict cycles / LDS instr |")
print("|---|---|---|---|---|---|---|---|---|")
for name in sorted(rows):
    r = rows[name]
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in r or "GRBM_GUI_ACTIVE" not in r or r["SQ_VALU_MFMA_BUSY_CYCLES"] == 0:
        continue
    short = name.replace("krs::", "")
    act = r["GRBM_GUI_ACTIVE"] / 8.0
    busy = r["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0          # 256 CUs x 4 SIMDs
    us = times.get(name)
    clock = "%.2f" % (act / us / 1e3) if us else "-"
    wc = r.get("SQ_WAVE_CYCLES", 0)
    buckets = "-"
    if wc and "SQ_WAIT_INST_ANY" in r:
        buckets = "%.0f / %.0f / %.0f" % (100 * r["SQ_WAIT_INST_ANY"] / wc, 100 * r.get("SQ_WAIT_ANY", 0) / wc, 100 * r.get("SQ_ACTIVE_INST_ANY", 0) / wc)
    lds = "-"
    if r.get("SQ_INSTS_LDS"):
        lds = "%.3f" % (r.get("SQ_LDS_BANK_CONFLICT", 0) / r["SQ_INSTS_LDS"])
    print("| `%s` | %s | %s | %.0f | %s | %.0f | **%.1f** | %s | %s |" % (short, ROLE.get(short, ""), "%.1f" % us if us else "-", act, clock, busy, 100 * busy / act, buckets, lds))
