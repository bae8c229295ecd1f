#!/bin/bash
# PMC rows of mips_filter_i8 (and mips_filter_f16 beside it) at 2032 queries x 18M rows: separate SQ and GRBM passes
# (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Run on the GPU box from the repo root: bash scripts/dev_nominate_pmc.sh [tag]
set -u
export TMPDIR=/tmp
export PYTHONPATH=$PWD
OUT=$PWD/gpurun_out/${1:-nom_pmc}
mkdir -p $OUT
S="$PWD/scripts/dev_nominate_ab.py 18e6 2032 80 normal"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_WAVES"
cd /tmp
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/sq -o p -- python3 $S > $OUT/sq.log 2>&1
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/sq2 -o p -- python3 $S > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o p -- python3 $S > $OUT/grbm.log 2>&1
cd - > /dev/null
python3 - "$OUT" <<'PY'
import collections, csv, os, sys
root = sys.argv[1]
def load(sub, key):
    per = collections.defaultdict(float); seen = set(); dur = 0.0
    p = os.path.join(root, sub, "p_counter_collection.csv")
    if not os.path.exists(p): return None
    for r in csv.DictReader(open(p)):
        if key not in r["Kernel_Name"]: continue
        per[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per["launches"] = len(seen); per["duration_ns"] = dur
    return per
for key, cyc in (("mips_filter_i8<2>", 32), ("mips_filter_f16<2", 32)):
    sq, sq2, gr = load("sq", key), load("sq2", key), load("grbm", key)
    if not sq or not gr or not gr["launches"]: continue
    scale = gr["launches"] / max(sq["launches"], 1)
    print(f"== {key}: {int(gr['launches'])} launches, {gr['duration_ns'] / gr['launches'] / 1e3:.1f} us average, effective clock {gr['GRBM_GUI_ACTIVE'] / 8 / gr['duration_ns']:.3f} GHz")
    print(f"   matrix pipe busy of GPU-active cycles {sq['SQ_VALU_MFMA_BUSY_CYCLES'] * scale / (1024 * gr['GRBM_GUI_ACTIVE'] / 8):.3f}; of SQ-busy {sq['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * sq['SQ_BUSY_CYCLES'] / 32):.3f}")
    print(f"   wave cycles: waiting (s_waitcnt / barrier) {sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}, issue-stalled {sq['SQ_WAIT_INST_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}, issuing {sq['SQ_ACTIVE_INST_ANY'] / sq['SQ_WAVE_CYCLES']:.3f}; LDS bank conflicts {sq['SQ_LDS_BANK_CONFLICT'] / max(sq['SQ_LDS_IDX_ACTIVE'], 1):.4f} of LDS cycles")
    if sq2:
        m = max(sq2["SQ_INSTS_MFMA"], 1)
        print("   per MFMA: VALU %.2f  SALU %.2f  LDS %.2f  VMEM-write %.3f  branch %.2f; LDS-issue stall %.3f of wave cycles (this pass)" % (
            (sq2["SQ_INSTS_VALU"] - sq2["SQ_INSTS_MFMA"]) / m, sq2["SQ_INSTS_SALU"] / m, sq2["SQ_INSTS_LDS"] / m, sq2["SQ_INSTS_VMEM_WR"] / m, sq2["SQ_INSTS_BRANCH"] / m, 0.0))
PY
