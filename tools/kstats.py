import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms %.2f"%(tot/1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 20]:
    n=r['Name'].split('(')[0].replace('void ','').replace('frcnn::','')[:70]
    print("%-72s %6s %9.2f ms %8.1f us %5.1f%%"%(n,r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,float(r['Percentage'])))
