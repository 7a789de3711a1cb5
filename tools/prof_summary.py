import csv, glob, collections, sys, os
d = sys.argv[1]
for f in glob.glob(os.path.join(d, 'kt', '*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %s avg %.3f ms  %s%%" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6, r['Percentage']))
tot = collections.defaultdict(float); n = collections.Counter()
for f in sorted(glob.glob(os.path.join(d, 'pmc_*', '*counter_collection.csv'))):
    seen = set()
    for r in csv.DictReader(open(f)):
        if 'mesh_dp' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            seen.add(r['Dispatch_Id'])
    for c in set(r2 for r2 in tot): pass
    n[f] = len(seen)
nl = max(n.values()) if n else 1
w = tot.get('SQ_WAVES', 1) / nl
print("per launch (avg of %d launches), per wave:" % nl)
for k in sorted(tot):
    print("  %-26s %12.4g   per wave %10.4g" % (k, tot[k]/nl, tot[k]/nl/w))
