import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def grp(name):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if name in r["Kernel_Name"]]
    n = 23
    return [sum(d[i * n + 3:(i + 1) * n]) / 20 / 1e3 for i in range(len(d) // n)]
a, b = grp("gemm_tn_kernel"), grp("gemm_tn_reduce")
shapes = [(2048, 512), (512, 2048), (512, 512), (1024, 512), (512, 1024), (5000, 512), (512, 128), (128, 512), (512, 64), (64, 512)]
for s, x, y in zip(shapes, a, b):
    print("%5d x %4d: main %6.1f us  reduce %5.1f us  (%.0f TFLOP/s in the main kernel)" % (s[0], s[1], x, y, 2.0 * 15392 * s[0] * s[1] / x / 1e6))
