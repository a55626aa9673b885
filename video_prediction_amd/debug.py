"""Developer / soak-test switches: poison everything a correct launch sequence must never read, and record where a run happened.

None of this is on by default and nothing here computes a result.  It exists because the round-5 parity suite once failed on ONE box with the
binary that was green on four others: the questions "does a kernel read memory / LDS it has not written?" and "which box was that?" need an
answer that does not depend on luck.

* ``POISON['alloc']``    tests/conftest.py (SAVP_POISON=1) wraps torch.empty / empty_like / empty_strided / new_empty so that every device
                         allocation handed to the engine holds 0xFF bytes (NaN as fp32 / bf16 / fp64, -1 as an integer)
* ``POISON['scratch']``  kernels.Scratch refills the caller-owned scratch (split-K slices, weight-gradient partials, colsum / dense
                         workspaces) with NaN before every hand-out
* ``POISON['lds']``      every stream-ordered entry of the C ABI is preceded by savp_debug_poison_lds on the same stream (lib.get())
* ``POISON['arena']``    kernels.ZeroArena.take verifies (eager mode only) that the slice it hands out really is all-zero
"""
import ctypes
import json
import os
import subprocess
import time

POISON = {'alloc': False, 'scratch': False, 'lds': False, 'arena': False}
NAN_WORD = 0xFFFFFFFF
COUNTS = {'alloc': 0, 'scratch': 0, 'lds': 0, 'arena': 0}      # how often each mode fired (printed by the pytest session summary)


def configure_from_env():
    """SAVP_POISON=1 (everything) or a comma list of alloc,scratch,lds,arena."""
    v = os.environ.get('SAVP_POISON', '')
    if not v or v == '0':
        return False
    names = POISON.keys() if v in ('1', 'all', 'nan') else [s.strip() for s in v.split(',')]
    for n in names:
        if n not in POISON:
            raise ValueError('SAVP_POISON: unknown item %r (alloc, scratch, lds, arena)' % n)
        POISON[n] = True
    return True


def poison_tensor(t):
    """Fill a device tensor's bytes with 0xFF through the library's own fill kernel (stream-ordered, safe inside a capture)."""
    from . import lib
    if t is None or not t.is_cuda or t.numel() == 0:
        return t
    nbytes = t.numel() * t.element_size()
    if not t.is_contiguous() or (t.data_ptr() & 3) or (nbytes & 3):
        import torch
        if t.dtype.is_floating_point:
            t.fill_(float('nan'))
        else:
            t.fill_(-1 if t.dtype != torch.uint8 and t.dtype != torch.bool else 1)
        return t
    lib.check(lib.get_raw().savp_debug_fill_u32(lib.stream(), ctypes.c_void_p(t.data_ptr()), nbytes // 4, NAN_WORD), 'savp_debug_fill_u32')
    return t


def poison_lds():
    from . import lib
    raw = lib.get_raw()
    lib.check(raw.savp_debug_poison_lds(lib.stream(), NAN_WORD, None), 'savp_debug_poison_lds')


def poison_free_blocks(big_gb=8, small_mb=256):
    """Fill the caching allocator's FREE blocks with NaN: allocate, fill, release (the blocks stay cached, poisoned)."""
    import torch
    blocks = []
    try:
        blocks.append(torch.empty(big_gb * (1 << 28), device='cuda'))
    except RuntimeError:
        pass
    for mb in (64, 16, 4, 2, 1):                                   # the large pool's mid-sized free blocks (best fit: a 4 MB request re-uses a freed 4 MB block)
        try:
            blocks += [torch.empty(mb << 18, device='cuda') for _ in range(8)]
        except RuntimeError:
            break
    blocks += [torch.empty(1 << 17, device='cuda') for _ in range(small_mb * 2)]
    blocks += [torch.empty(1 << 8, device='cuda') for _ in range(2048)]
    for b in blocks:
        poison_tensor(b)
    torch.cuda.synchronize()
    del blocks


# ---- where did this run happen? ---------------------------------------------------------------------------------------------------------

def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _run(cmd, timeout=20):
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
        return r.stdout
    except (OSError, subprocess.SubprocessError):
        return None


def box_fingerprint(with_tools=True):
    """Everything cheap that identifies the GPU box: device name / CU / XCD count, driver + firmware versions, compute / memory partition
    mode, the GPU's unique id, RAS / ECC error counters, clocks.  Read from sysfs first (works without root), rocminfo / rocm-smi / amd-smi
    when present.  Every evidence script and the pytest session header print it."""
    fp = {'time': time.strftime('%Y-%m-%dT%H:%M:%S'), 'host': _read('/etc/hostname'), 'kernel': _read('/proc/sys/kernel/osrelease'),
          'amdgpu_version': _read('/sys/module/amdgpu/version'), 'rocm': _read('/opt/rocm/.info/version'), 'cards': []}
    drm = '/sys/class/drm'
    try:
        cards = sorted(c for c in os.listdir(drm) if c.startswith('card') and c[4:].isdigit())
    except OSError:
        cards = []
    for c in cards:
        d = os.path.join(drm, c, 'device')
        if _read(os.path.join(d, 'vendor')) != '0x1002':
            continue
        card = {'card': c}
        try:
            card['pci'] = os.path.basename(os.path.realpath(d))
        except OSError:
            pass
        for key in ('device', 'revision', 'unique_id', 'vbios_version', 'current_compute_partition', 'current_memory_partition',
                    'mem_info_vram_total', 'mem_info_vram_used', 'pcie_replay_count', 'current_link_speed', 'current_link_width',
                    'power_dpm_force_performance_level', 'gpu_busy_percent', 'serial_number', 'product_name'):
            v = _read(os.path.join(d, key))
            if v is not None:
                card[key] = v
        fw = os.path.join(d, 'fw_version')
        if os.path.isdir(fw):
            card['fw'] = {f: _read(os.path.join(fw, f)) for f in sorted(os.listdir(fw))}
        ras = os.path.join(d, 'ras')
        if os.path.isdir(ras):
            card['ras'] = {f: _read(os.path.join(ras, f)) for f in sorted(os.listdir(ras)) if f.endswith('err_count') or f == 'features'}
        fp['cards'].append(card)
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(0)
            fp['torch'] = {'name': p.name, 'gcn_arch': getattr(p, 'gcnArchName', None), 'cus': p.multi_processor_count,
                           'total_memory': p.total_memory, 'torch': torch.__version__, 'hip': torch.version.hip,
                           'pci': '%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), getattr(p, 'pci_bus_id', 0), getattr(p, 'pci_device_id', 0)),
                           'uuid': str(getattr(p, 'uuid', None))}
            # the host may expose all eight cards in sysfs while this process sees one: the visible one is the card at that PCI address
            for c in fp['cards']:
                if c.get('pci', '').lower().startswith(fp['torch']['pci']):
                    fp['visible_card'] = c
    except Exception as e:                                        # the fingerprint must never take a run down
        fp['torch'] = {'error': repr(e)}
    if with_tools:
        out = _run(['/opt/rocm/bin/rocminfo'])
        if out:
            keep = [ln.strip() for ln in out.splitlines()
                    if any(k in ln for k in ('Marketing Name', 'Name:', 'Compute Unit', 'Shader Engines', 'Num XCC', 'Max Clock', 'Uuid', 'Chip ID',
                                             'ASIC Revision', 'Internal Node ID', 'Features:', 'Coherent Host Access'))]
            fp['rocminfo'] = keep[:80]
        out = _run(['/opt/rocm/bin/rocm-smi', '--showuniqueid', '--showdriverversion', '--showvbios', '--showcomputepartition', '--showmemorypartition',
                    '--showrasinfo', 'all', '--showperflevel', '--showclocks', '--json'])
        if out:
            try:
                fp['rocm_smi'] = json.loads(out[out.index('{'):])
            except ValueError:
                fp['rocm_smi_raw'] = out[-4000:]
        out = _run(['/opt/rocm/bin/amd-smi', 'static', '--json'], timeout=30)
        if out:
            try:
                fp['amd_smi_static'] = json.loads(out[out.index('['):] if '[' in out[:50] else out[out.index('{'):])
            except ValueError:
                fp['amd_smi_raw'] = out[-4000:]
    return fp


def fingerprint_id(fp):
    """Short stable id of the box (unique id of card 0 when the driver exposes it, else a hash of the static parts)."""
    import hashlib
    for c in ([fp['visible_card']] if fp.get('visible_card') else []) + fp.get('cards', []):
        if c.get('unique_id'):
            return c['unique_id']
    static = json.dumps({'host': fp.get('host'), 'cards': [{k: v for k, v in c.items() if k not in ('mem_info_vram_used', 'gpu_busy_percent', 'ras')}
                                                           for c in fp.get('cards', [])]}, sort_keys=True)
    return 'h' + hashlib.sha256(static.encode()).hexdigest()[:15]


def write_fingerprint(path, **extra):
    fp = box_fingerprint()
    fp['id'] = fingerprint_id(fp)
    fp.update(extra)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(fp, f, indent=1, sort_keys=True, default=str)
    return fp


configure_from_env()      # SAVP_POISON in the environment switches the modes on in every process that imports the package


if __name__ == '__main__':
    print(json.dumps(box_fingerprint(), indent=1, sort_keys=True, default=str))
