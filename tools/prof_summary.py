"""Condense a rocprofv3 --kernel-trace --stats output directory into a small summary
(top kernels by total time + every ursa kernel), so it can be committed under profiles/.
    python tools/prof_summary.py <rocprof_out_dir> <summary.csv> [--keep-trace]
Deletes the (large) per-dispatch trace unless --keep-trace."""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    stats = sorted(glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True))
    if not stats:
        raise SystemExit(f'no kernel_stats.csv under {src}')
    rows = list(csv.DictReader(open(stats[0])))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    keep = [r for i, r in enumerate(rows) if i < 25 or 'k_' in r['Name'] and 'anonymous namespace' in r['Name']]
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        for r in keep:
            name = r['Name'] if len(r['Name']) < 160 else r['Name'][:157] + '...'
            w.writerow([name, r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs'],
                        r['StdDev']])
        w.writerow(['# total kernel time ns', '', int(total), '', '', '', '', ''])
        w.writerow(['# kernels (distinct)', len(rows), '', '', '', '', '', ''])
        w.writerow(['# dispatches', sum(int(r['Calls']) for r in rows), '', '', '', '', '', ''])
    if '--keep-trace' not in sys.argv:
        for f in glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True):
            os.remove(f)
    print(open(dst).read())


if __name__ == '__main__':
    main()
