"""Summarise rocprofv3 --pmc CSVs under gpurun_out/pmc_*/ for the path kernel (sum over dispatches)."""
import collections, csv, glob, json, sys
agg = collections.defaultdict(float); disp = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pt_path' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); disp.add((f, r['Dispatch_Id']))
out = dict(agg); out["dispatches_seen"] = len(disp)
print(json.dumps(out, indent=1))
