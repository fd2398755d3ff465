"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch and kernel.

usage: python tools/pmc_traffic.py <dir-of-FETCH_SIZE-pass> <dir-of-WRITE_SIZE-pass> [out.json]

Counter values are summed over the dispatches of a kernel and divided by their number.  rocprofv3
reports both counters in KiB; FETCH_SIZE tallies every 128-B request as 64 B on gfx950
(MI355X_MICROARCH.md, HBM section), so the fetch bytes are doubled."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, counter):
    tot, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                k = row['Kernel_Name'].split('(')[0].replace('void ', '')
                tot[k] += float(row['Counter_Value'])
                n[k].add(row.get('Dispatch_Id', len(n[k])))
    return {k: (tot[k], len(n[k])) for k in tot}


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    F, W = collect(fd, 'FETCH_SIZE'), collect(wd, 'WRITE_SIZE')
    out = []
    for k in sorted(F):
        f, nf = F[k]
        w, nw = W.get(k, (0.0, 1))
        fetch = f * 1024 / max(nf, 1)           # KiB -> bytes, per launch
        write = w * 1024 / max(nw, 1)
        out.append({'kernel': k, 'launches': nf, 'fetch_size_bytes_raw': fetch, 'fetch_bytes_corrected_x2': 2 * fetch,
                    'write_bytes': write, 'hbm_bytes_per_launch': 2 * fetch + write})
    js = json.dumps({'kernels': out}, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(js)
    for o in out:
        print('%-40s n=%3d fetch*2 %8.3f GB  write %8.3f GB  total %8.3f GB' % (
            o['kernel'][:40], o['launches'], o['fetch_bytes_corrected_x2'] / 1e9, o['write_bytes'] / 1e9, o['hbm_bytes_per_launch'] / 1e9))


if __name__ == '__main__':
    main()
