import glob, os, sqlite3, sys
base = os.path.join("gpurun_out", "pmc_" + sys.argv[1])
for p in sorted(glob.glob(os.path.join(base, "p*", "pmc_results.db"))):
    d = sqlite3.connect(p)
    for name, kern, n, tot, mx in d.execute("select counter_name, kernel_name, count(*), sum(value), max(value) from counters_collection group by counter_name, kernel_name"):
        if "k_render" in kern and mx > 0:
            print(f"{name:<40} launches={n:<3} max_launch={mx:.6g}")
