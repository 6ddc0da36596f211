import csv, glob, collections, sys
base = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
def load(d):
    fs = glob.glob(f'{base}/{d}/*/*_counter_collection.csv')
    if not fs: return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg
print("rocprofv3 --pmc passes (tools_pmc.sh), bench.py c30 102x102x19 nt=30, mean per launch; FETCH/WRITE_SIZE in KB")
print("(gfx950: FETCH_SIZE counts 64 B per 128-B request -> up to 2x the bytes for wide coalesced reads, MI355X_MICROARCH.md)")
for d in ['sq1','sq2','fetch','write','tcc']:
    agg = load(d)
    for k in sorted(agg):
        if k.startswith('k_'):
            print(f"{d:6s} {k:20s} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c,v in sorted(agg[k].items())))
