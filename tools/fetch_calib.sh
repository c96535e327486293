#!/bin/bash
# On the GPU box: the byte counters of tools/ubench/fetch_calib (built beforehand) -> gpurun_out/fetch_calib.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/fetch_calib
mkdir -p $O
timeout -k 10 120 ./tools/ubench/fetch_calib > $O/plain.txt || exit 1
i=0
for PMC in "FETCH_SIZE TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc$i -o pmc -- ./tools/ubench/fetch_calib > $O/pmc$i.txt 2> $O/pmc$i.err || { echo "pass $i failed"; tail -3 $O/pmc$i.err; exit 1; }
done
python3 - <<'PY' > gpurun_out/fetch_calib.txt
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/fetch_calib/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print(open("gpurun_out/fetch_calib/plain.txt").read())
for k, d in vals.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
cat gpurun_out/fetch_calib.txt
