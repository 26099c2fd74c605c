import csv,sys,collections
for f in sys.argv[1:]:
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'tsvpp::' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(f.split('/')[-1], k, 'n=%d'%len(v), 'mean=%.4g'%(sum(v)/len(v)))
