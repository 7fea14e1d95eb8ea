#!/bin/bash
# usage (GPU box): tools/pmc_vcycle.sh <tag> "<counters>" [kernel regex]  -> gpurun_out/pmc_<tag>.txt (per-kernel mean of each counter)
set -e
tag=$1; ctrs=$2; pat=${3:-pair}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcdir_$tag
rm -rf $out
PMG_VC_ONLY=${VC_ONLY:-513} rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/vcyclebench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
python3 - "$out" "$pat" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, re, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if re.search(sys.argv[2], name):
            agg[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), d in sorted(agg.items()):
    print(name, grid, {k: round(sum(v) / len(v)) for k, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
rm -rf $out
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
