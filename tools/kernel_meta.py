"""Per-kernel resource metadata of the SHIPPED library: VGPRs, SGPRs, LDS, scratch ("private segment") and spill counts
of every gfx950 kernel inside libqd_hip.so, read from the code objects the .so embeds (no rebuild, no assembly files).

    python tools/kernel_meta.py [path/to/lib.so]        # table, kernels with scratch or spills first

tests/test_abi.py uses kernels() to require that NO kernel of the library uses scratch memory: a spilled per-lane array
in an HBM-bound kernel is extra, uncounted memory traffic.

How: the `.hip_fatbin` section of the .so is a concatenation of clang offload bundles (one per translation unit,
magic `__CLANG_OFFLOAD_BUNDLE__`); each bundle lists (offset, size, triple) entries; the `hipv4-amdgcn-amd-amdhsa--gfx950`
entry is an ELF whose NT_AMDGPU_METADATA note carries the kernels' metadata, which `llvm-readelf --notes` prints.
"""
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
LLVM_BIN = '/opt/rocm/lib/llvm/bin'


def _tool(name):
    exe = shutil.which(name) or os.path.join(LLVM_BIN, name)
    if not os.path.exists(exe):
        raise RuntimeError('%s not found (looked on PATH and in %s)' % (name, LLVM_BIN))
    return exe


def code_objects(so_path):
    """The gfx950 ELF images embedded in so_path, as bytes objects."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.check_call([_tool('llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', so_path, fat])
        blob = open(fat, 'rb').read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        p = pos + len(MAGIC)
        (nent,) = struct.unpack_from('<Q', blob, p)
        p += 8
        for _ in range(nent):
            off, size, idlen = struct.unpack_from('<QQQ', blob, p)
            p += 24
            ident = blob[p:p + idlen].decode()
            p += idlen
            if 'amdgcn' in ident and 'gfx950' in ident and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    if not out:
        raise RuntimeError('no gfx950 code object found in %s' % so_path)
    return out


_FIELDS = ('.name', '.vgpr_count', '.agpr_count', '.sgpr_count', '.private_segment_fixed_size', '.group_segment_fixed_size',
           '.vgpr_spill_count', '.sgpr_spill_count', '.max_flat_workgroup_size', '.uses_dynamic_stack')


def kernels(so_path):
    """[{name, vgpr_count, sgpr_count, private_segment_fixed_size (scratch bytes per lane), group_segment_fixed_size
    (static LDS bytes), vgpr_spill_count, sgpr_spill_count, ...}] for every kernel of the library."""
    res = []
    for img in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(img)
            f.flush()
            txt = subprocess.check_output([_tool('llvm-readelf'), '--notes', f.name]).decode()
        # the metadata is YAML: "amdhsa.kernels:" followed by list items starting with "  - .xxx:"
        m = re.search(r'amdhsa\.kernels:\s*\n(.*?)(?:\namdhsa\.|\Z)', txt, re.S)
        if not m:
            continue
        for item in re.split(r'\n\s*-\s+(?=\.[a-z_]+:)', '\n' + m.group(1)):
            d = {}
            for key in _FIELDS:
                mm = re.search(r'^\s*' + re.escape(key) + r':\s*(.+?)\s*$', item, re.M)
                if mm:
                    v = mm.group(1).strip("'\"")
                    d[key[1:]] = int(v) if re.fullmatch(r'-?\d+', v) else v
            if 'name' in d and 'vgpr_count' in d:
                res.append(d)
    return res


def wide_stores(so_path):
    """{kernel symbol: (16-byte global stores, how many of them carry the `nt` hint)} from the disassembly of the shipped code
    objects (llvm-objdump).  The write-once outputs of the streaming kernels are stored non-temporally; a hint lost on the
    way through the optimiser (merged loop copies drop !nontemporal) shows up here as a plain global_store_dwordx4."""
    res = {}
    for img in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(img)
            f.flush()
            txt = subprocess.check_output([_tool('llvm-objdump'), '-d', '--mcpu=gfx950', f.name]).decode()
        name = None
        for line in txt.splitlines():
            m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
            if m:
                name = m.group(1)
                continue
            if name and 'global_store_dwordx4' in line:
                tot, nt = res.get(name, (0, 0))
                res[name] = (tot + 1, nt + (1 if re.search(r'\bnt\b', line) else 0))
    return res


def demangle(names):
    try:
        out = subprocess.check_output([shutil.which('c++filt') or _tool('llvm-cxxfilt')] + list(names)).decode().splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, '..', 'quantized_distillation_amd', 'libqd_hip.so')
    ks = kernels(so)
    dm = demangle([k['name'] for k in ks])
    ks.sort(key=lambda k: (-(k.get('private_segment_fixed_size', 0)), -k['vgpr_count']))
    print('%d kernels in %s' % (len(ks), os.path.relpath(so)))
    print('%7s %5s %5s %7s %6s  %s' % ('scratch', 'vgpr', 'sgpr', 'lds', 'spills', 'kernel'))
    for k in ks:
        print('%7d %5d %5d %7d %6d  %s' % (k.get('private_segment_fixed_size', 0), k['vgpr_count'], k.get('sgpr_count', 0),
                                          k.get('group_segment_fixed_size', 0),
                                          k.get('vgpr_spill_count', 0) + k.get('sgpr_spill_count', 0),
                                          dm[k['name']].replace('(anonymous namespace)::', '')))
    bad = [k for k in ks if k.get('private_segment_fixed_size', 0) or k.get('vgpr_spill_count', 0)]
    print('%d kernels use scratch memory' % len(bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
