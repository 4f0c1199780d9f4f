set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for L in intree wnaf4; do
  if [ $L = wnaf4 ]; then export GPBC_LIB_PATH=$R/variants/libgpbc_wnaf4.so; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmcab_$L -- python3 $R/bench.py --batch 262144 --steps 1 --warmup 0 --no-secondary --no-cpu > $R/gpurun_out/pmcab_$L.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmcab2_$L -- python3 $R/bench.py --batch 262144 --steps 1 --warmup 0 --no-secondary --no-cpu > $R/gpurun_out/pmcab2_$L.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for L in ("intree","wnaf4"):
    agg=collections.defaultdict(float); cnt=collections.defaultdict(int)
    for f in glob.glob("$R/gpurun_out/pmcab*_%s/**/*counter_collection.csv"%L, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith("k_final_exp("):
                agg[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
    print(L, {c: agg[c]/cnt[c] for c in sorted(agg)})
PY
