#!/bin/bash
# rocprofv3 PMC passes for the pairing kernels (separate passes; never combined with --sys-trace etc.).
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <tag> [batch]
set -e
TAG=${1:-pmc}; B=${2:-262144}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for PASS in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  N=$(echo $PASS | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $PASS --output-format csv -d $R/gpurun_out/${TAG}_$N -- python3 $R/bench.py --batch $B --steps 1 --warmup 0 --no-secondary --no-cpu > $R/gpurun_out/${TAG}_$N.log 2>&1 || echo "pass $N failed"
done
python3 - <<PY
import csv, glob, collections, os
R="$R"; TAG="$TAG"
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{R}/gpurun_out/{TAG}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0]; c=row["Counter_Name"]; agg[k][c]+=float(row["Counter_Value"]); cnt[k][c]+=1
with open(f"{R}/gpurun_out/{TAG}_summary.txt","w") as out:
    for k in sorted(agg):
        if not k.startswith("k_"): continue
        out.write(k+"\n")
        for c in sorted(agg[k]): out.write("   %-28s per-dispatch mean %.6g  (dispatches %d)\n"%(c, agg[k][c]/cnt[k][c], cnt[k][c]))
import json
traffic={}
for k in agg:
    kk=k.strip('"')
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        n=cnt[k]["FETCH_SIZE"]
        traffic[kk]={"fetch_kb": agg[k]["FETCH_SIZE"]/n, "write_kb": agg[k]["WRITE_SIZE"]/cnt[k]["WRITE_SIZE"], "batch": $B if n==1 else min($B,262144), "launches_per_batch": 1}
json.dump(traffic, open(f"{R}/gpurun_out/{TAG}_traffic.json","w"), indent=1)
# per-kernel instruction counts, independent of the box's clock: copy to profiles/pmc_counters.json for bench.py
counters={"source": f"profiles/{TAG}_counters.json (tools/pmc_run.sh, batch $B per launch)", "kernels": {}}
for k in agg:
    kk=k.strip('"')
    if not kk.startswith("k_") or "SQ_WAVES" not in agg[k]: continue
    w=agg[k]["SQ_WAVES"]/cnt[k]["SQ_WAVES"]
    g=lambda c: (agg[k][c]/cnt[k][c]) if c in agg[k] else None
    counters["kernels"][kk]={"waves": w, "valu_instr_per_wave": g("SQ_INSTS_VALU")/w if g("SQ_INSTS_VALU") else None,
        "wait_any_share_of_wave_cycles": g("SQ_WAIT_ANY")/g("SQ_WAVE_CYCLES") if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") else None,
        "vmem_instr_per_wave": (g("SQ_INSTS_VMEM") or 0)/w, "lds_instr_per_wave": (g("SQ_INSTS_LDS") or 0)/w, "salu_instr_per_wave": (g("SQ_INSTS_SALU") or 0)/w}
json.dump(counters, open(f"{R}/gpurun_out/{TAG}_counters.json","w"), indent=1)
print(open(f"{R}/gpurun_out/{TAG}_summary.txt").read())
PY
