"""Build profiles/pmc_traffic.json from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, each collected on its own with
--kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) of
`python3 bench.py --steps S --warmup W --no-cpu-baseline --no-other-modes --no-profile`.

    python scripts/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <forwards in the run = S + W> [kernel_stats.csv] > profiles/pmc_traffic.json

Corrections: FETCH_SIZE is reported in KiB and, on gfx950, counts whole-line reads at half their size (x 1024 x 2, MI355X_MICROARCH.md) -
but not every access shape: profiles/r04_fetch_calibration.json holds the factor measured per shape with scripts/probes/fetch_calib_probe
(1 GiB read once per shape: 2.00 for coalesced / whole-record / LDS-DMA reads, 1.66 for the four-lanes-per-pixel quarter slices of
conv3x3_f16x3_qp / conv3x3s2_v2, 1.43 for the two-lanes-per-pixel 32-byte pieces of the other conv kernels) and which shape each kernel
family's reads have; the family's factor is applied.  WRITE_SIZE in KiB -> x 1024.  With a kernel_stats.csv (rocprofv3 --stats of the
same command) every family gets `implied_TBps` = counted bytes / average launch duration and `exceeds_sustained` when that is above the
4.8 TB/s this pool sustains for a mixed stream (such a figure is a miscalibrated counter, not traffic).  The file is
stamped with the hash of the kernel sources it was measured on (bench.py refuses a stale figure)."""
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
    targs = [t.strip() for t in k.split('<', 1)[-1].rsplit('>', 1)[0].split(',')] if '<' in k else []
    if base == 'conv3x3s2_v2' and first:
        return f'{base}<{first},k32>' if len(targs) >= 6 and targs[5] == 'true' else f'{base}<{first}>'      # (K32: 32-channel chunks, 16-bit mode)
    if base == 'conv3x3_first_split':
        return 'conv3x3_first_stats' if targs and targs[-1] == 'false' else 'conv3x3_first_split'          # STORE = false: the statistics-only pass
    if base in ('conv3x3_f16x3_p', 'conv3x3_f16x3_one', 'conv3x3_h32', 'conv3x3_upc') and first:
        return f'{base}<{first}>'
    if base == 'conv3x3_res32' and 'true' in k.split('<', 1)[-1].split('>')[0].split(',')[-1]:
        return 'conv3x3_res32f'                                   # the FUSE instantiation (first block recomputed inside)
    if base == 'conv3x3_first' and k.split('<', 1)[-1].split('>')[0].split(',')[-1].strip() == 'false' and k.count(',') >= 4:
        return 'conv3x3_first_stats'                              # STORE = false
    return {'head_mfma32': 'head', 'head_1x1': 'head', 'finalize_stats_t': 'finalize_stats', 'convT2x2_f16x3_one': 'convT2x2_f16x3'}.get(base, base)


SUSTAINED_TBPS = 4.8          # scripts/probes/hbm_probe.hip on this pool: copy / mixed streams 4.5-5.0 TB/s


def main():
    from bench import csrc_hash
    fetch, nf = load(sys.argv[1], 'FETCH_SIZE')
    write, nw = load(sys.argv[2], 'WRITE_SIZE')
    passes = int(sys.argv[3])
    cal = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r04_fetch_calibration.json')))
    factor_of = lambda label: cal['shapes'][cal['kernel_shape'].get(label, cal['default_shape'])]['factor']
    dur = {}                                                    # label -> [total ns, calls] from the --stats pass of the same command
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        for r in csv.DictReader(open(sys.argv[4])):
            d = dur.setdefault(short(r['Name']), [0.0, 0]); d[0] += float(r['TotalDurationNs']); d[1] += int(r['Calls'])
    agg = {}
    for k in set(fetch) | set(write):
        lab = short(k)
        g = agg.setdefault(lab, {'disp': 0, 'fb': 0.0, 'wb': 0.0})
        g['disp'] += int(nf.get(k, nw.get(k, 0))); g['fb'] += fetch.get(k, 0) * 1024 * factor_of(lab); g['wb'] += write.get(k, 0) * 1024
    per, avg = {}, {}
    for k, g in sorted(agg.items(), key=lambda kv: -(kv[1]['fb'] + kv[1]['wb'])):
        per[k] = {'dispatches': g['disp'], 'launches_per_step': g['disp'] / passes, 'fetch_GB_per_step_corrected': round(g['fb'] / passes / 1e9, 3),
                  'write_GB_per_step': round(g['wb'] / passes / 1e9, 3), 'fetch_factor': factor_of(k),
                  'read_shape': cal['kernel_shape'].get(k, cal['default_shape'])}
        if g['disp']:
            avg[k] = int((g['fb'] + g['wb']) / g['disp'])
            if k in dur and dur[k][1]:
                tbps = avg[k] / (dur[k][0] / dur[k][1]) / 1e3          # bytes / ns = GB/s -> / 1e3 = TB/s
                per[k]['avg_launch_us'] = round(dur[k][0] / dur[k][1] / 1e3, 1)
                per[k]['implied_TBps'] = round(tbps, 2)
                per[k]['exceeds_sustained'] = bool(tbps > SUSTAINED_TBPS)
    out = {
        'csrc_hash': csrc_hash(),
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --no-cpu-baseline --no-other-modes --no-profile, B=64, split mode',
        'correction': 'FETCH_SIZE [KiB] x 1024 x the factor of the family\'s read shape (profiles/r04_fetch_calibration.json: 2.00 whole lines, 1.66 / 1.43 '
                      'quarter / 32-byte slices); WRITE_SIZE [KiB] x 1024',
        'forwards_in_run': passes,
        'hbm_bytes_per_launch_avg': avg,
        'per_kernel': per,
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
