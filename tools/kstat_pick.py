"""tools/kstat_pick.py DIR substr... -- average duration (us) of the kernels whose name contains a substring, from rocprofv3 --stats CSVs under DIR"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r["Name"] for s in sys.argv[2:]):
            print("  %-50s calls %5s avg %7.1f us" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3))
