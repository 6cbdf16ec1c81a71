import sys, os, csv, glob, collections
O=sys.argv[1]
# kernel stats
for f in glob.glob(O+'/trace/**/*kernel_stats.csv', recursive=True)+glob.glob(O+'/stats/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats', os.path.basename(f))
    for r in csv.DictReader(open(f)):
        print('  %-90s calls %4s avg_ns %12s total_ns %14s pct %s'%(r['Name'][:90], r['Calls'], r.get('AverageNs',r.get('Average')), r.get('TotalDurationNs',''), r.get('Percentage','')))
# counters: per kernel average
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'synth' in k: continue
    print('==',k)
    for c,vals in sorted(v.items()):
        print('   %-34s avg %.6g  (n=%d)'%(c,sum(vals)/len(vals),len(vals)))
