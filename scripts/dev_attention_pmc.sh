#!/bin/bash
# PMC sums of attention_fwd at one encode shape (gpurun box, repo root): one counter group per pass, --kernel-trace only.
# usage: scripts/dev_attention_pmc.sh [batch] [seq_len]
export TMPDIR=/tmp PYTHONPATH=$PWD
B=${1:-64}; S=${2:-512}
OUT=gpurun_out/attention_pmc_${B}x${S}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o a -- python3 scripts/dev_encode_shape.py $B $S 3 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq2 -o a -- python3 scripts/dev_encode_shape.py $B $S 3 > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o a -- python3 scripts/dev_encode_shape.py $B $S 3 > $OUT/grbm.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
sums, launches, dur = collections.defaultdict(float), collections.defaultdict(set), collections.defaultdict(float)
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_fwd" not in r["Kernel_Name"]:
            continue
        sums[r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Counter_Name"]].add(r["Dispatch_Id"])
with open(root + "/summary.txt", "w") as out:
    for k in sorted(sums):
        line = f"{k:32s} {sums[k] / max(len(launches[k]), 1):16.1f} per launch ({len(launches[k])} launches)"
        print(line); out.write(line + "\n")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sums and "GRBM_GUI_ACTIVE" in sums:
        busy = sums["SQ_VALU_MFMA_BUSY_CYCLES"] / len(launches["SQ_VALU_MFMA_BUSY_CYCLES"])
        act = sums["GRBM_GUI_ACTIVE"] / len(launches["GRBM_GUI_ACTIVE"])
        line = f"matrix pipe busy of GPU-active cycles: {busy / 1024 / (act / 8):.3f}; GPU-active cycles per launch {act / 8:.0f}"
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/sq $OUT/sq2 $OUT/grbm
