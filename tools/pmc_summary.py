import csv, collections, glob, re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "halo"
for d in ('pmc1','pmc2'):
    for f in glob.glob(f'gpurun_out/{d}/runc/*_counter_collection.csv'):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k=re.sub(r'\(.*','',r['Kernel_Name']).replace('void vk::','')
            if pat not in k: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,c in agg.items():
            print(k)
            for n,v in sorted(c.items()):
                print(f"    {n:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
