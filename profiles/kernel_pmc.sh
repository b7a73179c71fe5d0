#!/bin/bash
# SQ counters of ONE kernel of the benchmark step (run through gpurun from the repo root):
#   bash profiles/kernel_pmc.sh <kernel-name-substring> <tag>        e.g.  bash profiles/kernel_pmc.sh k_splat_tile r04_splat
# Separate --pmc passes, never combined with tracing; totals over the launches of one `bench.py --steps 1 --warmup 0`.
set -e
name=$1; tag=${2:-pmc}
repo=$PWD
out=$repo/gpurun_out/${tag}_sq
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/p$i.log 2>&1 || echo "pass $i failed"
done
cd $repo
python - "$out" "$tag" "$name" <<'PY'
import csv, glob, json, sys, os
root, tag, name = sys.argv[1], sys.argv[2], sys.argv[3]
acc = {}
for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if name in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(acc, open(os.path.join(root, f"{tag}_sq_pmc.json"), "w"), indent=1, sort_keys=True)
w = acc.get("SQ_WAVES", 1.0)
print(json.dumps({k: round(v / w, 1) for k, v in sorted(acc.items())}))
PY
