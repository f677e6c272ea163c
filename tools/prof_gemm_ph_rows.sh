# kernel durations of the layer GEMM shapes at window-batch / decode-batch row counts for every tile choice (GPU box):
#   tools/prof_gemm_ph_rows.sh <tag> [rows,rows,...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
O=gpurun_out/gemm_ph_rows; mkdir -p $O; rm -f $O/$1.txt
ROWS=${2:-8996,17992,24000,32000}
for v in auto small 256 192 128 64; do
  if [ $v = small ]; then export PAFC_PH_MIN_FILL=100000; else unset PAFC_PH_MIN_FILL; fi
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t -o r --output-format csv -- python3 tools/micro/gemm_sweep_ph.py $ROWS $v > /dev/null 2>&1
  python3 - $ROWS $v >> $O/$1.txt <<PY
import csv, sys, glob, collections
rows = list(csv.DictReader(open(glob.glob("$O/t/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
SH = ["w_1", "w_2", "pw2", "out", "rkv6"]
groups, cur = [], None
for r in rows:
    if "FillFunctor" in r["Kernel_Name"]:
        cur = []; groups.append(cur); continue
    if cur is not None:
        cur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
groups = [g for g in groups if len(g) >= 50]
Ms = [int(v) for v in sys.argv[1].split(",")]
assert len(groups) == len(Ms) * len(SH), len(groups)
gi = 0
for M in Ms:
    for sh in SH:
        g = groups[gi]; gi += 1
        print(f"{sys.argv[2]:5s} M={M:6d} {sh:5s} {sum(g) / 50.0 / 1e3:7.1f}")
PY
  rm -rf $O/t
done
python3 - <<PY
import collections
t = collections.defaultdict(dict)
for l in open("$O/$1.txt"):
    v, m, sh, us = l.split()
    t[(int(m[2:]) if m.startswith("M=") and len(m) > 2 else int(sh), sh)][v] = float(us)
PY
cat $O/$1.txt | awk '{k=$2" "$3" "$4; a[k]=a[k]" "$1"="$NF} END {for (k in a) print k, a[k]}' | sort -k2,2n -k3,3 > $O/$1_table.txt
cat $O/$1_table.txt
