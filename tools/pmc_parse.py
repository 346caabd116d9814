import csv,sys,collections,glob
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
rows=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    rows[(int(r['Dispatch_Id']),r['Kernel_Name'][:60])][r['Counter_Name']]=float(r['Counter_Value'])
seen=set()
for (d,k),v in sorted(rows.items(),reverse=True):
    if k in seen: continue
    seen.add(k); print(d,k,{a:int(b) for a,b in v.items()})
