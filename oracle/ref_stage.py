"""Stages the REFERENCE's own quantization package as compiled bytecode -- TEST INFRASTRUCTURE ONLY.

`north_star` asks for "the reference's own CPU quantize timed on the GPU box's host cores in the
same run", but /root/reference does not exist on the GPU box.  The recipe below compiles the three
files of the path from the sources where they lie,

    /root/reference/quantization/__init__.py
    /root/reference/quantization/quant_functions.py        (uniformQuantization: :155-194)
    /root/reference/quantization/help_functions.py

with py_compile into oracle/_ref/quantization/*.pyc (bytecode only -- no reference source is
copied; oracle/_ref/ is git-ignored, so nothing of it enters the history, and it is NOT
gpurun-ignored, so it travels to the GPU box like the repo's own built .so files).  Python
imports a package made of .pyc files alone ("sourceless" import) when the interpreter version
matches, which it does: the GPU box runs this same image.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call load(); the
product (quantized_distillation_amd/, quantization/, harness/) never does.
"""
import hashlib
import importlib
import json
import os
import py_compile
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = '/root/reference'
STAGE_DIR = os.path.join(_HERE, '_ref')
PKG_DIR = os.path.join(STAGE_DIR, 'quantization')
FILES = ('__init__.py', 'quant_functions.py', 'help_functions.py')
_cached = None


def stage(ref_root=REF_ROOT, force=False):
    """Compile the reference's quantization package into oracle/_ref/ (no-op when the reference
    is absent, e.g. on the GPU box, where the staged files arrive with the snapshot).  Returns
    the staged package directory or None."""
    src_dir = os.path.join(ref_root, 'quantization')
    if not all(os.path.exists(os.path.join(src_dir, f)) for f in FILES):
        return PKG_DIR if is_staged() else None
    os.makedirs(PKG_DIR, exist_ok=True)
    manifest = {'python': '%d.%d.%d' % sys.version_info[:3], 'source': src_dir, 'files': {}}
    for f in FILES:
        src = os.path.join(src_dir, f)
        with open(src, 'rb') as fh:
            manifest['files'][f] = hashlib.sha256(fh.read()).hexdigest()
        out = os.path.join(PKG_DIR, f + 'c')                    # legacy location: importable without the source
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            py_compile.compile(src, cfile=out, dfile='reference/quantization/' + f, doraise=True)
    with open(os.path.join(STAGE_DIR, 'STAGED.json'), 'w') as fh:
        json.dump(manifest, fh, indent=1)
    return PKG_DIR


def is_staged():
    return all(os.path.exists(os.path.join(PKG_DIR, f + 'c')) for f in FILES)


def manifest():
    try:
        with open(os.path.join(STAGE_DIR, 'STAGED.json')) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def load():
    """The reference's `quantization` package (a module object), imported from the staged bytecode
    WITHOUT disturbing the product's package of the same name: the reference files import each
    other absolutely (`import quantization`, `import quantization.help_functions as qhf`,
    quant_functions.py:4-5), so for the duration of the import the name `quantization` in
    sys.modules is pointed at the staged package, then the previous entries are put back.  The
    reference modules keep direct references to each other in their globals, so they go on
    working afterwards.  Returns None when nothing is staged and the reference is absent."""
    global _cached
    if _cached is not None:
        return _cached
    if not is_staged() and stage() is None:
        return None
    names = ('quantization', 'quantization.quant_functions', 'quantization.help_functions')
    saved = {k: sys.modules.pop(k) for k in names if k in sys.modules}
    sys.path.insert(0, STAGE_DIR)
    importlib.invalidate_caches()
    try:
        mod = importlib.import_module('quantization')
        if not os.path.abspath(getattr(mod, '__file__', '') or '').startswith(STAGE_DIR):
            raise ImportError('imported %r instead of the staged reference package' % (mod,))
        _cached = mod
    finally:
        sys.path.remove(STAGE_DIR)
        for k in names:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
        importlib.invalidate_caches()
    return _cached
