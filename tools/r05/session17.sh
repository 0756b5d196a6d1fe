R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for abl in 0 128; do
  rm -rf $O/r05y_pmc_$abl
  FQ_ABL_LIST=$abl timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/r05y_pmc_$abl -- python $R/tools/fq_ablate.py > /dev/null 2>&1
  f=$(find $O/r05y_pmc_$abl -name "*counter_collection.csv" | head -1)
  python - "$f" $abl <<'PY'
import csv,sys
from collections import defaultdict
tot=defaultdict(float); cnt=defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "fine_vs_kernel" in r["Kernel_Name"]:
        tot[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("ablate", sys.argv[2], {k: round(tot[k]/cnt[k]/1e6,3) for k in tot}, "M per launch;", dict(cnt))
PY
done
