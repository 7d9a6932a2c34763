#!/bin/bash
# LDS bank-conflict share per kernel (rocprofv3 PMC pass): tools/lds_conflicts.sh python3 /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_lds -- "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_lds/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    if "hsefr" not in k:
        continue
    c = {n: sum(x) / len(x) for n, x in v.items()}
    act = c.get("SQ_LDS_IDX_ACTIVE", 0)
    name = k.split("::")[-1][:70]
    print("%-72s launches %4d  LDS active %10.0f  conflicts %10.0f  (%4.1f %%)" % (name, len(v["SQ_LDS_IDX_ACTIVE"]), act, c.get("SQ_LDS_BANK_CONFLICT", 0),
                                                                                  100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / act if act else 0))
PY
