#!/bin/bash
# SQ counters of the kNN kernel (run through gpurun from the repo root): bash profiles/knn_pmc.sh <tag>
set -e
tag=${1:-r01_vX}
repo=$PWD
out=$repo/gpurun_out/${tag}_sq
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/p$i.log 2>&1 || echo "pass $i failed"
done
cd $repo
python - "$out" "$tag" <<'PY'
import csv, glob, hashlib, json, sys, os
root, tag = sys.argv[1], sys.argv[2]
acc = {}
for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_knn_" in r["Kernel_Name"] and "k_knn_heap" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            acc["kernel"] = r["Kernel_Name"][:80]
# what bench.py checks before it quotes these numbers: the kernel source they were measured with, and the launch they describe
acc["knn_hip_sha16"] = hashlib.sha256(open("treee_amd/csrc/knn.hip", "rb").read()).hexdigest()[:16]
acc["points_per_launch"] = 15000000
acc["waves_per_simd"] = int(os.environ.get("KNN_WAVES_PER_SIMD", "3"))
json.dump(acc, open(os.path.join(root, f"{tag}_knn_sq_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(acc, indent=1, sort_keys=True))
PY
