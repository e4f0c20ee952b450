import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv)>2 else 8]:
    print(r["Name"].split("(")[0][:44].ljust(44), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"])/1e3), "%8.2f ms total" % (float(r["TotalDurationNs"])/1e6), "min %.1f max %.1f" % (float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
