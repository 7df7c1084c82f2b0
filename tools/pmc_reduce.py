#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of the
same command to HBM bytes per launch of the kernels whose name contains a pattern (or one of several, comma-separated).

    python tools/pmc_reduce.py --fetch <dir-or-csv> --write <dir-or-csv> --kernel k_zgemm --out profiles/x.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE counts half of the bytes of wide
coalesced reads and is doubled here (same guide)."""
import argparse, csv, glob, json, os


def load(path, counter, pattern):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True)
    tot, n = 0.0, 0
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter or not any(pt in r.get('Kernel_Name', '') for pt in pattern.split(',')):
                continue
            key = r.get('Dispatch_Id')
            per[key] = per.get(key, 0.0) + float(r['Counter_Value'])
    for v in per.values():
        tot += v; n += 1
    return tot, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fetch', required=True); ap.add_argument('--write', required=True)
    ap.add_argument('--kernel', required=True); ap.add_argument('--out', required=True)
    ap.add_argument('--note', default='')
    ap.add_argument('--grid', type=int, default=1024); ap.add_argument('--batch', type=int, default=256)
    a = ap.parse_args()
    f_kib, nf = load(a.fetch, 'FETCH_SIZE', a.kernel)
    w_kib, nw = load(a.write, 'WRITE_SIZE', a.kernel)
    rd = 2.0 * f_kib * 1024.0
    wr = w_kib * 1024.0
    out = dict(kernel_pattern=a.kernel, launches_fetch_pass=nf, launches_write_pass=nw,
               fetch_KiB_raw=f_kib, write_KiB=w_kib, hbm_read_bytes_corrected=rd, hbm_write_bytes=wr,
               traffic_bytes_per_launch=(rd / max(nf, 1) + wr / max(nw, 1)), grid=[a.grid, a.grid], batch=a.batch,
               units='FETCH_SIZE/WRITE_SIZE in KiB per dispatch; FETCH_SIZE doubled (gfx950)', note=a.note)
    json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
