"""Build profiles/pmc_traffic.json from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, each collected on its own with
--kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) of
`python3 bench.py --steps S --warmup W --no-cpu-baseline --no-other-modes --no-profile`.

    python scripts/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <forwards in the run = S + W> > profiles/pmc_traffic.json

Corrections: FETCH_SIZE is reported in KiB and, on gfx950, counts wide coalesced reads at half their size -> x 1024 x 2;
WRITE_SIZE in KiB -> x 1024.  The file is stamped with the hash of the kernel sources it was measured on (bench.py refuses a
stale figure)."""
import csv, collections, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        tot[k] += float(r['Counter_Value']); n[k] += 1
    return tot, n


def short(k):
    """rocprof kernel name -> the label ts2d_engine_op_kernel reports (variants that share a label are summed)."""
    k = k.replace('void ', '').replace('ts2d::', '')
    k = re.sub(r'\(.*$', '', k)
    m = re.match(r'([A-Za-z0-9_]+)(?:<(\d+))?', k)
    base, first = m.group(1), m.group(2)
    if base in ('conv3x3_f16x3_p', 'conv3x3_f16x3_one', 'conv3x3_h32', 'conv3x3_upc', 'conv3x3s2_v2') and first:
        return f'{base}<{first}>'
    return {'head_mfma32': 'head', 'head_1x1': 'head', 'finalize_stats_t': 'finalize_stats', 'convT2x2_f16x3_one': 'convT2x2_f16x3'}.get(base, base)


def main():
    from bench import csrc_hash
    fetch, nf = load(sys.argv[1], 'FETCH_SIZE')
    write, nw = load(sys.argv[2], 'WRITE_SIZE')
    passes = int(sys.argv[3])
    agg = {}
    for k in set(fetch) | set(write):
        g = agg.setdefault(short(k), {'disp': 0, 'fb': 0.0, 'wb': 0.0})
        g['disp'] += int(nf.get(k, nw.get(k, 0))); g['fb'] += fetch.get(k, 0) * 2048; g['wb'] += write.get(k, 0) * 1024
    per, avg = {}, {}
    for k, g in sorted(agg.items(), key=lambda kv: -(kv[1]['fb'] + kv[1]['wb'])):
        per[k] = {'dispatches': g['disp'], 'launches_per_step': g['disp'] / passes, 'fetch_GB_per_step_corrected': round(g['fb'] / passes / 1e9, 3),
                  'write_GB_per_step': round(g['wb'] / passes / 1e9, 3)}
        if g['disp']:
            avg[k] = int((g['fb'] + g['wb']) / g['disp'])
    out = {
        'csrc_hash': csrc_hash(),
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --no-cpu-baseline --no-other-modes --no-profile, B=64, split mode',
        'correction': 'FETCH_SIZE [KiB] x 1024 x 2 (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section); WRITE_SIZE [KiB] x 1024',
        'forwards_in_run': passes,
        'hbm_bytes_per_launch_avg': avg,
        'per_kernel': per,
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
