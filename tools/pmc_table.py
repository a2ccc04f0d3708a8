"""Average per-dispatch counter values per kernel from rocprofv3 --pmc csv output(s): python tools/pmc_table.py counter_collection.csv [...] [--filter substr]"""
import csv, sys, collections
files = [a for a in sys.argv[1:] if not a.startswith('--')]
flt = [a[9:] for a in sys.argv[1:] if a.startswith('--filter=')]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if flt and not any(x in n for x in flt):
            continue
        key = (n.split('(')[0][-44:], r['Grid_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if (f, r['Dispatch_Id']) not in seen:
            seen.add((f, r['Dispatch_Id']))
            dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    print(k, 'avg dur %.1f us over %d dispatches' % (sum(dur[k]) / len(dur[k]) / 1e3, len(dur[k])))
    for c in sorted(v):
        print('   %-30s %.4g' % (c, sum(v[c]) / len(v[c])))
