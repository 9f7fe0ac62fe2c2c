import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms over %d kernels (both repetitions of the probe: two calibrating forwards, four plain ones)" % (tot/1e6, len(rows)))
for r in rows[:22]:
    print("%-72s calls %5s total %7.2f ms avg %7.1f us %5s%%" % (r['Name'][:72], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
