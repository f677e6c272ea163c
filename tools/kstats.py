"""Print the top rows of a rocprofv3 --kernel-trace --stats output directory (kernel name, calls, average ns)."""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in list(csv.DictReader(open(files[0])))[:n]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:10.1f} us')
