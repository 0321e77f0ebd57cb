"""Build the committed profiles/ summaries of a round from the output directory of scripts/collect_profiles.sh.

    python scripts/build_profiles.py gpurun_out/prof r02

Writes profiles/<tag>_bench.json, <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_traffic.json (+ the
pmc_traffic.json bench.py reads; FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections), <tag>_sq_counters.txt and
<tag>_phase_stamps.txt (in-kernel stamps, TS2D_DBG=256), <tag>_resources.txt (VGPR / AGPR / SGPR / spills / scratch per kernel instantiation)."""
import collections, csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from pmc_traffic import short  # noqa: E402

FORWARDS = 13          # scripts/collect_profiles.sh: --steps 10 --warmup 3
NAMES = ['SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_ANY',
         'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY']


def sq_table(path, tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name']); agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_BUSY_CYCLES':
            disp[k] += 1
    out = ['# rocprofv3 --kernel-trace --pmc ' + ' '.join(NAMES) + ' -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-profile',
           f'# {tag}, final kernel sources (B=64 2x512x512, split mode); sums over the 13 forwards of the run; kernel labels as ts2d_engine_op_kernel',
           '# derived: mfma/busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (summed over the SIMDs of a shader engine: 32 = every matrix pipe busy all the time);',
           '#          lds_conflict% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;  lds/busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES (8 = every CU of the SE indexing LDS all the time)',
           '%-28s %5s %12s %12s %12s %12s %12s %9s %13s %8s' % ('kernel', 'disp', 'busy', 'wave_cyc', 'mfma_busy', 'lds_active', 'lds_conflict', 'mfma/busy', 'lds_conflict%', 'lds/busy')]
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['SQ_BUSY_CYCLES']):
        if v['SQ_BUSY_CYCLES'] < 1e6:
            continue
        out.append('%-28s %5d %12.4g %12.4g %12.4g %12.4g %12.4g %9.2f %12.1f%% %8.2f' % (
            k[:28], disp[k], v['SQ_BUSY_CYCLES'], v['SQ_WAVE_CYCLES'], v['SQ_VALU_MFMA_BUSY_CYCLES'], v['SQ_LDS_IDX_ACTIVE'], v['SQ_LDS_BANK_CONFLICT'],
            v['SQ_VALU_MFMA_BUSY_CYCLES'] / v['SQ_BUSY_CYCLES'], 100 * v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1), v['SQ_LDS_IDX_ACTIVE'] / v['SQ_BUSY_CYCLES']))
    return '\n'.join(out) + '\n'


def main():
    src, tag = sys.argv[1], sys.argv[2]
    P = os.path.join(ROOT, 'profiles')
    shutil.copy(os.path.join(src, 'bench.json'), os.path.join(P, f'{tag}_bench.json'))
    shutil.copy(os.path.join(src, 'prof_kt', 'runc_kernel_stats.csv'), os.path.join(P, f'{tag}_kernel_stats.csv'))
    f16 = os.path.join(src, 'prof_kt_f16', 'runc_kernel_stats.csv')
    if os.path.exists(f16):
        with open(os.path.join(P, f'{tag}_kernel_stats_f16.csv'), 'w') as f:
            f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --precision f16 --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-profile (13 forwards)\n')
            f.write(open(f16).read())
    hp = os.path.join(src, 'hbm_probe.txt')
    if os.path.exists(hp):
        shutil.copy(hp, os.path.join(P, f'{tag}_hbm_probe.txt'))
    pj = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'pmc_traffic.py'), os.path.join(src, 'pmc_fetch', 'runc_counter_collection.csv'),
                         os.path.join(src, 'pmc_write', 'runc_counter_collection.csv'), str(FORWARDS), os.path.join(src, 'prof_kt', 'runc_kernel_stats.csv')],
                        check=True, capture_output=True, text=True).stdout
    for name in ('pmc_traffic.json', f'{tag}_pmc_traffic.json'):
        open(os.path.join(P, name), 'w').write(pj)
    open(os.path.join(P, f'{tag}_sq_counters.txt'), 'w').write(sq_table(os.path.join(src, 'pmc_sq', 'runc_counter_collection.csv'), tag))
    # register / scratch / LDS table of every kernel instantiation (hipcc cross-compiles here: amdhsa metadata)
    subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'kernel_resources.py'), os.path.join(P, f'{tag}_resources.txt')], check=True, capture_output=True)
    st = os.path.join(src, 'phase_stamps.txt')
    if os.path.exists(st):
        lines = [l for l in open(st) if l.startswith('[phases]') or l.startswith('[split')]
        seen, keep = set(), []
        for l in reversed(lines):                      # the last forward's line per op
            k = l.split()[1] if l.startswith('[phases]') else l[:20]
            if k not in seen:
                seen.add(k); keep.append(l)
        hdr = ('# TS2D_DBG=256 python scripts/gpu_ops_only.py split: shader-clock cycles of wave 0 per workgroup between in-kernel stamps (kernels.h TS2D_STAMP_AT)\n'
               '# conv3x3_upc (decN.c0, N<=4): [0]/[1] phase-1 staging / MFMAs (+ wait at the next barrier), [2]/[3] phase 2, [4] bias + stores, [5]/[6] barriers\n'
               '# conv3x3s2_v2 (round 4): [3] accumulate + merge, [1] top barrier, [2] wait for the raw patch, [6] conversion, [0] DMA wait + barrier, [5] the nine taps (+ one patch load each), [4] tile epilogue\n'
               '# conv3x3_f16x3_q: [0] wait at the end-of-chunk barrier, [1] chunk body, [3]-[5] epilogue\n')
        open(os.path.join(P, f'{tag}_phase_stamps.txt'), 'w').write(hdr + ''.join(reversed(keep)))
    j = json.load(open(os.path.join(P, f'{tag}_bench.json')))
    print(tag, j['value'], 'slices/s;', j['roofline']['kernel'], 'frac', j['roofline']['frac'])


if __name__ == '__main__':
    main()
