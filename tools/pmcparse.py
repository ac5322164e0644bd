import csv, collections, sys, glob
base=sys.argv[1]
for k in sys.argv[2:]:
    m={}
    for f in glob.glob(f'{base}/{k}/*_counter_collection.csv'):
        rows=list(csv.DictReader(open(f)))
        d=collections.defaultdict(list)
        for r in rows:
            if 'mdx' in r['Kernel_Name'] and 'splitk' not in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
        for c,v in d.items(): m[c]=sum(v)/len(v)
    print(k, {c: round(v,1) for c,v in sorted(m.items())})
