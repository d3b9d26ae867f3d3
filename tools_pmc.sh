#!/bin/bash
# Collect PMC counters for the propagation kernels (separate passes, kernel-trace only), 1e6-packet bench.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/bench.py --packets ${PMC_PACKETS:-1000000} --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "pass $tag rc=$?"
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc"
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root+"/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::","")
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
with open(root+"/summary.txt","w") as out:
    for k,v in sorted(agg.items()):
        if not any(x in k for x in ("k_thermal","k_rpkt","k_slow","k_macroatom")): continue
        out.write(k+"\n")
        for c,val in sorted(v.items()): out.write(f"   {c:40s} {val:.6g}\n")
print(open(root+"/summary.txt").read())
PY
