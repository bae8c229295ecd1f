cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; rm -rf $O; mkdir -p $O
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/sq -o p -- python3 scripts/dev_gemm_prof.py > $O/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/grbm -o p -- python3 scripts/dev_gemm_prof.py > $O/grbm.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python3 scripts/dev_gemm_prof.py > $O/fetch.log 2>&1
python3 - <<'PY' > $O/summary.txt
import csv, collections, glob
for sub in ("sq","grbm","fetch"):
    f = glob.glob(f"gpurun_out/r02s/{sub}/**/p_counter_collection.csv", recursive=True)
    if not f: print(sub, "missing"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen=set()
    for r in csv.DictReader(open(f[0])):
        n = "own" if "gemm_tn_f16" in r["Kernel_Name"] else ("lib" if "Cijk" in r["Kernel_Name"] else None)
        if not n: continue
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[n]["launches"] += 1
    for n in per:
        d = per[n]; print(sub, n, "avg_us", dur[n]/d["launches"]/1e3, dict(d))
        if sub == "sq":
            print("   mfma_busy_of_sq_busy", d["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*d["SQ_BUSY_CYCLES"]/32), "wait", d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"], "stall", d["SQ_WAIT_INST_ANY"]/d["SQ_WAVE_CYCLES"], "active", d["SQ_ACTIVE_INST_ANY"]/d["SQ_WAVE_CYCLES"], "lds_conflict", d["SQ_LDS_BANK_CONFLICT"]/max(d["SQ_LDS_IDX_ACTIVE"],1), "lds_active_frac", d["SQ_LDS_IDX_ACTIVE"]/(256*d["SQ_BUSY_CYCLES"]/32))
        if sub == "grbm":
            print("   clock GHz", d["GRBM_GUI_ACTIVE"]/8/dur[n])
        if sub == "fetch":
            print("   fetch MB per launch (x2 corrected)", 2*d["FETCH_SIZE"]*1024/d["launches"]/1e6)
PY
rm -rf $O/sq $O/grbm $O/fetch
