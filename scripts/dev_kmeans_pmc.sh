#!/bin/bash
# PMC sums per launch of the k-means assignment kernels (gpurun box, repo root): one counter group per pass, --kernel-trace
# only.  usage: scripts/dev_kmeans_pmc.sh [PROQA_KMEANS_TWO_PASS value]
export TMPDIR=/tmp
export PROQA_KMEANS_TWO_PASS=${1:-1}
OUT=gpurun_out/kmeans_pmc_$PROQA_KMEANS_TWO_PASS
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o km -- python3 scripts/dev_kmeans_assign_timing.py 4e6 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/sq2 -o km -- python3 scripts/dev_kmeans_assign_timing.py 4e6 > $OUT/sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o km -- python3 scripts/dev_kmeans_assign_timing.py 4e6 > $OUT/grbm.log 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
root = sys.argv[1]
sums, launches = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "kmeans_assign" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"][r["Kernel_Name"].index("kmeans_assign"):][:28]
        sums[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[name][r["Counter_Name"]].add(r["Dispatch_Id"])
with open(root + "/summary.txt", "w") as out:
    for name in sorted(sums):
        for k in sorted(sums[name]):
            line = f"{name:30s} {k:32s} {sums[name][k] / max(len(launches[name][k]), 1):16.1f} per launch ({len(launches[name][k])} launches)"
            print(line); out.write(line + "\n")
        s, l = sums[name], launches[name]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s:
            busy = s["SQ_VALU_MFMA_BUSY_CYCLES"] / len(l["SQ_VALU_MFMA_BUSY_CYCLES"]); act = s["GRBM_GUI_ACTIVE"] / len(l["GRBM_GUI_ACTIVE"])
            line = f"{name:30s} matrix pipe busy of GPU-active cycles: {busy / 1024 / (act / 8):.3f}"
            print(line); out.write(line + "\n")
PY
rm -rf $OUT/sq $OUT/sq2 $OUT/grbm
