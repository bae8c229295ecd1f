"""Overlap statistics of a rocprofv3 kernel-trace CSV (dev helper): for every kernel class, the share of its running time during
which a kernel of ANOTHER queue / stream was running too, and the pairs that overlap most.

    python scripts/trace_overlap.py <kernel_trace.csv> [last_fraction=0.5]

Only the last `last_fraction` of the trace (by time) is analysed: the warm-up runs on one stream."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(path)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey] if qkey else "0", r["Kernel_Name"]) for r in rows]
ev.sort()
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * (1.0 - frac)
ev = [e for e in ev if e[0] >= t_lo]
queues = sorted({e[2] for e in ev})
print(f"{len(ev)} launches in the analysed window, queues ({qkey}): {queues}")


def short(name):
    for key in ("gemm_tn_f16", "Cijk", "attention_fwd", "bias_residual_layernorm", "embed_layernorm", "pool_project", "small_dense",
                "gather_rows", "mips_filter_i8", "mips_filter_f16", "topk_merge", "bootstrap", "finalize", "prep_queries"):
        if key in name:
            return "library GEMM (Cijk...)" if key == "Cijk" else key
    return name[:40]


total = defaultdict(float)
shared = defaultdict(float)
pair = defaultdict(float)
# sweep: O(n * active) with a small active set
active = []
for s, e, q, name in ev:
    active = [a for a in active if a[1] > s]
    for s2, e2, q2, n2 in active:
        if q2 != q:
            ov = min(e, e2) - s
            if ov > 0:
                shared[short(name)] += ov
                shared[short(n2)] += ov
                pair[tuple(sorted((short(name), short(n2))))] += ov
    total[short(name)] += e - s
    active.append((s, e, q, name))
span = ev[-1][1] - ev[0][0]
busy = sum(total.values())
print(f"window {span / 1e6:.3f} ms, kernel time summed over queues {busy / 1e6:.3f} ms ({busy / span:.3f} x the window)")
for k_, v in sorted(total.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k_:32s} {v / 1e6:9.3f} ms   beside a kernel of another queue {100 * shared[k_] / v:5.1f} % of its time")
print("most overlapping pairs:")
for k_, v in sorted(pair.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {k_[0]:28s} + {k_[1]:28s} {v / 1e6:8.3f} ms")
