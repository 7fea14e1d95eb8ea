#!/bin/bash
# GPU box: mean of one or more PMC counters per kernel for a python tool.  usage: tools/pmc_one.sh <tag> "<counters>" <regex> <tool.py> [args]
set -e
tag=$1; ctrs=$2; pat=$3; shift 3
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc1_$tag
rm -rf $out
rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python3 $root/$@ > $root/gpurun_out/pmc1_$tag.log 2>&1
python3 - "$out" "$pat" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if re.search(sys.argv[2], name):
            agg[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), d in sorted(agg.items()):
    print(name[:70], grid, {k: round(sum(v) / len(v)) for k, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
rm -rf $out
