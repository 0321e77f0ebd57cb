"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name (sums over dispatches)."""
import csv, collections, sys
d = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']; d[k][r['Counter_Name']] += float(r['Counter_Value'])
names = sorted({c for k in d for c in d[k]})
print('kernel'.ljust(48), *[n[-18:].rjust(18) for n in names])
for k in sorted(d, key=lambda k: -d[k].get('SQ_WAVE_CYCLES', d[k].get(names[0], 0)))[:14]:
    print(k.replace('void ts2d::', '').replace('(ts2d::ConvArgs)', '')[:48].ljust(48), *[f'{d[k][n]:18.4g}' for n in names])
