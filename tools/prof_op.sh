#!/bin/bash
# usage: bash tools/prof_op.sh <tag> <op> [POINTOPS_DEBUG]  -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc.json (FETCH/WRITE/SQ per kernel and launch)
TAG=$1; OP=$2
R=${GRAFT_REPO_ROOT:-$PWD}
export POINTOPS_DEBUG="$3"
cd /tmp && export TMPDIR=/tmp
python $R/tools/run_op.py $OP 20
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG.trace -- python $R/tools/run_op.py $OP 10 > /dev/null 2>&1
cp $(find $R/gpurun_out/$TAG.trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/$TAG.trace
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/$TAG.pmc_$n -- python $R/tools/run_op.py $OP 18 > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections,json
out=collections.defaultdict(dict)
NCALLS=20  # run_op: 2 warm-up + 18 timed calls per counter pass
sums=collections.defaultdict(float)
for d in glob.glob("$R/gpurun_out/$TAG.pmc_*"):
    f=glob.glob(d+"/*/*counter_collection.csv")[0]
    tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void pointops::","")[:60]
        if "at::" in k: continue
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        if r["Counter_Name"] in ("FETCH_SIZE","WRITE_SIZE","SQ_INSTS_VALU"): sums[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,c in tot.items():
        for a,v in c.items(): out[k][a]=v/len(n[k])
for k in out:
    if "FETCH_SIZE" in out[k]: out[k]["fetch_MB_x2"]=out[k]["FETCH_SIZE"]*2/1024
    if "WRITE_SIZE" in out[k]: out[k]["write_MB"]=out[k]["WRITE_SIZE"]/1024
# per CALL of the op (every kernel runs once per call here): HBM-side bytes, FETCH_SIZE x 2 (MI355X_MICROARCH.md) + WRITE_SIZE
tot={"fetch_MB_x2":sums["FETCH_SIZE"]*2/1024/NCALLS,"write_MB":sums["WRITE_SIZE"]/1024/NCALLS,"valu_insts":sums["SQ_INSTS_VALU"]/NCALLS}
tot["traffic_MB"]=tot["fetch_MB_x2"]+tot["write_MB"]
out["_per_call_all_kernels_avg_over_%d_calls" % NCALLS]=tot
json.dump(out,open("$R/gpurun_out/${TAG}_pmc.json","w"),indent=1,sort_keys=True)
for k,c in out.items(): print(k, {a: round(v,1) for a,v in c.items()})
PY
rm -rf $R/gpurun_out/$TAG.pmc_*
python - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/${TAG}_kernel_stats.csv")))[:8]:
    print("  %-70s %5s %9.1f us" % (r["Name"].replace("void pointops::","")[:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
