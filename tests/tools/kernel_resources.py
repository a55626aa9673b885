"""Register / LDS / scratch footprint of every kernel in libsavp_hip.so, read from the gfx950 code objects' metadata (no GPU needed).
usage: python tests/tools/kernel_resources.py [--filter REGEX] [--lib PATH] > profiles/<tag>_kernel_resources.txt
Columns: VGPRs (arch), AGPRs, SGPRs, spilled SGPRs / VGPRs, LDS bytes (static), scratch bytes per lane, max workgroup size, the waves
per SIMD the VGPR count allows (512 unified registers per lane on gfx950: floor(512 / (vgpr + agpr rounded up to 8)), capped at 8)."""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def code_objects(lib):
    """The gfx950 ELF images of every clang offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
        data = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out = []
    for m in re.finditer(magic, data):
        base = m.start()
        n, = struct.unpack_from('<Q', data, base + len(magic))
        pos = base + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if 'gfx950' in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(elf_bytes)
        f.flush()
        text = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], stdout=subprocess.PIPE, text=True).stdout
    demangled = {}
    mangled = subprocess.run(['c++filt'], input='\n'.join(re.findall(r'\.name:\s+(\S+)', text)), stdout=subprocess.PIPE, text=True).stdout.split('\n')
    for raw, nice in zip(re.findall(r'\.name:\s+(\S+)', text), mangled):
        demangled[raw] = nice
    recs = []
    for block in re.split(r'\n\s+- \.agpr_count:', text)[1:]:
        block = '.agpr_count:' + block

        def field(key, default='0'):
            mm = re.search(r'\.%s:\s+(\S+)' % key, block)
            return mm.group(1) if mm else default
        name = field('name', '?')
        if not name.startswith('_Z') and name == '?':
            continue
        recs.append(dict(name=demangled.get(name, name), vgpr=int(field('vgpr_count')), agpr=int(field('agpr_count')), sgpr=int(field('sgpr_count')),
                         sspill=int(field('sgpr_spill_count')), vspill=int(field('vgpr_spill_count')), lds=int(field('group_segment_fixed_size')),
                         scratch=int(field('private_segment_fixed_size')), wg=int(field('max_flat_workgroup_size'))))
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=os.path.join(ROOT, 'video_prediction_amd', 'libsavp_hip.so'))
    ap.add_argument('--filter', default='.')
    args = ap.parse_args()
    recs = []
    for co in code_objects(args.lib):
        recs += kernels_of(co)
    recs = [r for r in recs if re.search(args.filter, r['name'])]
    recs.sort(key=lambda r: r['name'])
    print('# %d kernels in %s' % (len(recs), os.path.basename(args.lib)))
    print('%5s %5s %5s %7s %7s %8s %8s %5s %6s  %s' % ('vgpr', 'agpr', 'sgpr', 'sspill', 'vspill', 'lds_B', 'scratch', 'wg', 'waves', 'kernel'))
    for r in recs:
        regs = -(-(r['vgpr'] + r['agpr']) // 8) * 8
        waves = min(8, 512 // max(regs, 1))
        short = re.sub(r'\(.*$', '', r['name'].replace('void ', ''))
        print('%5d %5d %5d %7d %7d %8d %8d %5d %6d  %s' % (r['vgpr'], r['agpr'], r['sgpr'], r['sspill'], r['vspill'], r['lds'], r['scratch'], r['wg'], waves, short))


if __name__ == '__main__':
    main()
