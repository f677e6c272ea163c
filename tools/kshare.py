"""Share of the kernel time per kernel from a rocprofv3 --kernel-trace --stats output directory:  python tools/kshare.py <dir> [n]"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms in {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f} %  {r["Calls"]:>7s} x {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:100]}')
