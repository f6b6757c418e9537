#!/bin/bash
# Instruction-cache counters of the 4 096-env step kernel for builds given as arguments (paths of libtaco_env.so variants):
#   bash tools/icache_ab.sh ab/lib_a.so ab/lib_b.so      (run through gpurun; prints per-launch means of the SQC i-cache counters)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp TACO_ENV_LIB_SKIP_ABI=1
for lib in "$@"; do
  v=$(basename "$lib" .so)
  export TACO_ENV_LIB=$R/$lib
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$O/ic_$v" -- python3 "$R/tools/prof_step.py" --envs 4096 --steps 300 > "$O/ic_$v.log" 2>&1
  python3 - "$O/ic_$v" "$v" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "taco_step_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(x[100:]) / max(1, len(x[100:]))) for k, x in sorted(acc.items())})
PY
done
