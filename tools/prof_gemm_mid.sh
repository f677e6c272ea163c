set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
O=gpurun_out/gemm_mid; mkdir -p $O; rm -f $O/summary.txt
for shape in "4096 512 2048 0 silu" "4096 2048 512 1 none" "4096 512 512 1 none" "4096 1024 512 1 none" "2048 512 2048 0 silu" "2048 2048 512 1 none"; do
  set -- $shape
  for path in own lib; do
    timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/t -o r --output-format csv -- python3 tools/micro/gemm_one.py $1 $2 $3 $path $4 $5 > /dev/null 2>&1
    python3 - "$shape" $path >> $O/summary.txt <<PY
import csv, sys
rows = list(csv.DictReader(open("$O/t/r_kernel_stats.csv")))
rows = [r for r in rows if int(r["Calls"]) >= 190]
print(sys.argv[1], sys.argv[2], " | ".join(f"{r['Name'][:60]} x{r['Calls']} {float(r['AverageNs'])/1e3:.1f}us" for r in rows[:3]))
PY
    rm -rf $O/t
  done
done
cat $O/summary.txt
