"""Register / scratch / LDS table of every kernel instantiation in libts2d_engine.so's device code ("report VGPR / AGPR /
scratch per instantiation").  Runs in the build container (hipcc cross-compiles without a GPU):

    python scripts/kernel_resources.py [profiles/r06_resources.txt]

Compiles csrc/engine.hip to gfx950 assembly with the Makefile's device flags and reads the amdhsa metadata hipcc writes per kernel
(the same numbers `llvm-readelf --notes` prints for the code object): arch VGPRs, AGPRs, SGPRs, spilled registers, scratch bytes per
lane, static LDS, max workgroup size -> waves per SIMD the register file allows (512 registers per SIMD lane: 2 waves at <= 256)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'totalsegmentator2d_amd', 'csrc', 'engine.hip')
HIPCC = '/opt/rocm/bin/hipcc'
FILT = 'c++filt'


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    devflags = subprocess.check_output(['make', '-s', '-C', os.path.dirname(SRC), 'flags'], text=True).split()
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, 'engine.s')
        subprocess.check_call([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', *devflags, '-o', s, SRC],
                              stderr=subprocess.DEVNULL)
        asm = open(s).read()
    meta = asm[asm.index('amdhsa.kernels:'):]
    rows = []
    for blk in re.split(r'\n  - \.', meta)[1:]:
        def get(key, default='0'):
            m = re.search(r'\.' + key + r':\s*(\S+)', '.' + blk)
            return m.group(1) if m else default
        name = get('name', '?')
        rows.append(dict(name=name, vgpr=int(get('vgpr_count')), agpr=int(get('agpr_count')), sgpr=int(get('sgpr_count')),
                         vspill=int(get('vgpr_spill_count')), sspill=int(get('sgpr_spill_count')),
                         scratch=int(get('private_segment_fixed_size')), lds=int(get('group_segment_fixed_size')),
                         wg=int(get('max_flat_workgroup_size'))))
    names = subprocess.check_output([FILT], input='\n'.join(r['name'].replace('DF16_', 'Dh') for r in rows), text=True).split('\n')
    for r, n in zip(rows, names):
        n = re.sub(r'^void ts2d::', '', n)
        n = re.sub(r'\(.*\)$', '', n)
        r['pretty'] = n.replace('(anonymous namespace)::', '').replace('half', '_Float16')      # (binutils' c++filt does not know DF16_: fed as Dh)
    rows.sort(key=lambda r: r['pretty'])
    lines = ['# scripts/kernel_resources.py: amdhsa metadata of every kernel in csrc/engine.hip, gfx950, Makefile device flags',
             '# vgpr = architectural VGPRs (incl. the AGPR half when used), waves/SIMD = floor(512 / max(vgpr, 1)) capped at 8;',
             '# vspill / sspill = spilled vector / scalar registers, scratch = bytes per lane, lds = static bytes (dynamic LDS is set at launch)',
             f'{"kernel":<64} {"vgpr":>5} {"agpr":>5} {"sgpr":>5} {"vspill":>6} {"sspill":>6} {"scratch":>8} {"lds":>7} {"wg":>5} {"waves/SIMD":>10}']
    for r in rows:
        waves = min(8, 512 // max(r['vgpr'], 1))
        lines.append(f'{r["pretty"][:64]:<64} {r["vgpr"]:>5} {r["agpr"]:>5} {r["sgpr"]:>5} {r["vspill"]:>6} {r["sspill"]:>6} {r["scratch"]:>8} {r["lds"]:>7} {r["wg"]:>5} {waves:>10}')
    text = '\n'.join(lines) + '\n'
    if out_path:
        open(out_path, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main()
