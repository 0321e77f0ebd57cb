"""Build profiles/pmc_traffic.json from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, each collected on its own with
--kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) of `python3 bench.py --steps S --warmup W --no-cpu-baseline`.

    python scripts/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps+warmup of split leg> > profiles/pmc_traffic.json

Corrections: FETCH_SIZE is reported in KiB and, on gfx950, counts wide coalesced reads at half their size -> x 1024 x 2;
WRITE_SIZE in KiB -> x 1024.  The stride-1 3x3 family of the split mode = every kernel whose name starts with
conv3x3_f16x3_one< or conv3x3_f16x3< and ends in the fp32-storage / 3-product instantiation."""
import csv, collections, json, sys


def load(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        tot[k] += float(r['Counter_Value']); n[k] += 1
    return tot, n


def is_s1_split(k):
    k = k.replace('void ', '').replace('ts2d::', '')
    if k.startswith('conv3x3_f16x3_one<'):
        return True
    return k.startswith('conv3x3_f16x3<') and 'float, 3>' in k


def main():
    fetch, nf = load(sys.argv[1], 'FETCH_SIZE')
    write, nw = load(sys.argv[2], 'WRITE_SIZE')
    passes = int(sys.argv[3])            # forwards of the split leg in the profiled run (steps + warmup + the untimed first one)
    per = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) * 2048 + write.get(k, 0) * 1024)):
        per[k[:80]] = {'dispatches': int(nf.get(k, nw.get(k, 0))), 'fetch_B_corrected_total': int(fetch.get(k, 0) * 2048),
                       'write_B_total': int(write.get(k, 0) * 1024)}
    s1 = [k for k in set(fetch) | set(write) if is_s1_split(k)]
    s1_bytes = sum(fetch.get(k, 0) * 2048 + write.get(k, 0) * 1024 for k in s1)
    s1_disp = sum(nf.get(k, 0) for k in s1)
    out = {
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --no-cpu-baseline, B=64',
        'correction': 'FETCH_SIZE [KiB] x 1024 x 2 (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section); WRITE_SIZE [KiB] x 1024',
        'split_forwards_in_run': passes,
        'conv3x3_s1_split_kernels': sorted(k[:80] for k in s1),
        'conv3x3_s1_split_launches_per_step': s1_disp // passes if passes else None,
        'conv3x3_s1_split_hbm_bytes_per_step': int(s1_bytes / passes) if passes else None,
        'conv3x3_s1_split_hbm_bytes_per_launch_avg': int(s1_bytes / s1_disp) if s1_disp else None,
        'conv3x3_s1_split_algorithmic_act_bytes_per_step': 29420000000,
        'per_kernel_totals_over_the_run': per,
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
