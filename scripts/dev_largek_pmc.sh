#!/bin/bash
# PMC sums of the one scan of the top-10000 search (6980 queries over 8.8M rows; gpurun box, repo root): one counter group
# per pass, --kernel-trace only (never combined with other trace domains).
export TMPDIR=/tmp
OUT=gpurun_out/largek_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o lk -- python3 scripts/dev_largek_timing.py 6980 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq2 -o lk -- python3 scripts/dev_largek_timing.py 6980 > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o lk -- python3 scripts/dev_largek_timing.py 6980 > $OUT/grbm.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
sums, launches = collections.defaultdict(float), collections.defaultdict(set)
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mips_filter_f16<2, 8, false, false, true" not in r["Kernel_Name"]:
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
        line = f"matrix pipe busy of GPU-active cycles: {busy / 1024 / (act / 8):.3f}  (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs))"
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/sq $OUT/sq2 $OUT/grbm
