# kernel durations of the layer's GEMM shapes at mid row counts, hand-written dispatcher vs library (GPU box):
#   tools/prof_gemm_mid2.sh <tag> [rows,rows,...]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
O=gpurun_out/gemm_mid2; mkdir -p $O
ROWS=${2:-1024,2048,3992,8192}
for path in own lib; do
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t -o r --output-format csv -- python3 tools/micro/gemm_sweep.py $ROWS $path > /dev/null 2>&1
  python3 - $ROWS $path >> $O/$1.txt <<PY
import csv, sys, glob, collections
rows = list(csv.DictReader(open(glob.glob("$O/t/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
SH = ["w_1 512x2048 silu", "w_2 2048x512 res", "pw2 512x512 res", "out 1024x512 res", "rkv 6x512x512", "lora down 2x512x128 tanh"]
groups, cur = [], None
for r in rows:
    if "fill" in r["Kernel_Name"].lower() or "FillFunctor" in r["Kernel_Name"]:
        cur = collections.defaultdict(list); groups.append(cur); continue
    if cur is not None:
        cur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
Ms = [int(v) for v in sys.argv[1].split(",")]
groups = [g for g in groups if sum(len(v) for v in g.values()) >= 100]     # (library helpers may emit fills of their own)
assert len(groups) == len(Ms) * len(SH), (len(groups), len(Ms) * len(SH))
gi = 0
for M in Ms:
    for sh in SH:
        g = groups[gi]; gi += 1
        per_call = sum(sum(v) for v in g.values()) / 100.0 / 1e3
        names = " + ".join(f"{k.split('(')[0][-48:]} x{len(v)}" for k, v in g.items() if len(v) >= 50)
        print(f"{sys.argv[2]:3s} M={M:5d} {sh:26s} {per_call:7.1f} us   {names}")
PY
  rm -rf $O/t
done
sort -k2,2 -k3,3 -s $O/$1.txt | head -80
