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

The same recipe stages the reference's TRAINING LOOP for the drop-in test
(tests/test_hip_dropin_reference_loop.py): cnn_models/{__init__,conv_forward_model,help_fun}.py and
helpers/functions.py under oracle/_ref/loop/.  Two statements cannot run on a current torch and are replaced in
the source text IN MEMORY before compiling: `loss.data[0]` on a 0-dim tensor (cnn_models/help_fun.py:156,158 --
SURVEY.md section 4) becomes `loss.item()`, and `optimizer.zero_grad()` in optimize_quantization_points
(conv_forward_model.py:519), after which the loop writes `points.grad.data`, becomes
`optimizer.zero_grad(set_to_none=False)` (torch >= 2.0 drops the gradients there by default).  Nothing else is
touched and no source is written anywhere in the repository.  load_loop(pkg) imports that loop with `import quantization`
resolving to `pkg` -- this repository's package or the staged reference package.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call load() / load_loop(); the
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

LOOP_DIR = os.path.join(STAGE_DIR, 'loop')
# (path relative to the reference root, [(old, new), ...] applied to the source text before compiling)
_ITEM_FIX = [('return loss.data[0], count_asked_teacher, count_total', 'return loss.item(), count_asked_teacher, count_total'),
             ('        return loss.data[0]\n', '        return loss.item()\n')]
# optimize_quantization_points writes `points.grad.data = ...` after `optimizer.zero_grad()` (conv_forward_model.py:519,545):
# torch >= 2.0 sets gradients to None there by default
_ZERO_GRAD_FIX = [('            optimizer.zero_grad()\n', '            optimizer.zero_grad(set_to_none=False)\n')]
LOOP_FILES = (('cnn_models/__init__.py', ()), ('cnn_models/conv_forward_model.py', _ZERO_GRAD_FIX),
              ('cnn_models/help_fun.py', _ITEM_FIX), ('helpers/functions.py', ()))


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


def stage_loop(ref_root=REF_ROOT, force=False):
    """Compile the reference's CNN training loop (train_model and what it imports) into oracle/_ref/loop/."""
    import tempfile
    if not all(os.path.exists(os.path.join(ref_root, rel)) for rel, _ in LOOP_FILES):
        return LOOP_DIR if loop_is_staged() else None
    for rel, fixes in LOOP_FILES:
        src = os.path.join(ref_root, rel)
        out = os.path.join(LOOP_DIR, rel + 'c')
        os.makedirs(os.path.dirname(out), exist_ok=True)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
            continue
        if not fixes:
            py_compile.compile(src, cfile=out, dfile='reference/' + rel, doraise=True)
            continue
        with open(src) as fh:
            text = fh.read()
        for old, new in fixes:
            if old not in text:
                raise RuntimeError('%s: the statement to fix is not there: %r' % (rel, old))
            text = text.replace(old, new)
        with tempfile.TemporaryDirectory() as tmp:              # the patched text never lands in the repository
            tmp_src = os.path.join(tmp, os.path.basename(rel))
            with open(tmp_src, 'w') as fh:
                fh.write(text)
            py_compile.compile(tmp_src, cfile=out, dfile='reference/' + rel + ' (+ torch-2 fix)', doraise=True)
    return LOOP_DIR


def loop_is_staged():
    return all(os.path.exists(os.path.join(LOOP_DIR, rel + 'c')) for rel, _ in LOOP_FILES)


def load_loop(quantization_pkg):
    """The reference's cnn_models.conv_forward_model (train_model, ConvolForwardNet, ...) from the staged bytecode,
    imported so that its `import quantization` / `import quantization.help_functions` resolve to `quantization_pkg`.
    Every call returns a FRESH set of module objects (two loops bound to two quantizers can live side by side);
    sys.modules is put back as it was.  None when nothing is staged and the reference is absent."""
    if not loop_is_staged() and stage_loop() is None:
        return None
    names = ('cnn_models', 'cnn_models.conv_forward_model', 'cnn_models.help_fun', 'helpers', 'helpers.functions',
             'quantization', 'quantization.quant_functions', 'quantization.help_functions')
    saved = {k: sys.modules.pop(k) for k in names if k in sys.modules}
    sys.modules['quantization'] = quantization_pkg
    sys.modules['quantization.help_functions'] = quantization_pkg.help_functions
    sys.modules['quantization.quant_functions'] = quantization_pkg.quant_functions
    sys.path.insert(0, LOOP_DIR)
    importlib.invalidate_caches()
    # conv_forward_model.py imports scikit-learn at the top (:19-23) for functions train_model never calls; the GPU box
    # has no scikit-learn, so empty stand-ins satisfy the import statements there
    stubs = []
    try:
        import sklearn  # noqa: F401
    except ImportError:
        import types
        for name in ('sklearn', 'sklearn.tree', 'sklearn.ensemble', 'sklearn.naive_bayes', 'sklearn.linear_model'):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                stubs.append(name)
        for name in stubs[1:]:
            setattr(sys.modules['sklearn'], name.split('.')[1], sys.modules[name])
    try:
        mod = importlib.import_module('cnn_models.conv_forward_model')
        if not os.path.abspath(getattr(mod, '__file__', '') or '').startswith(LOOP_DIR):
            raise ImportError('imported %r instead of the staged reference loop' % (mod,))
    finally:
        sys.path.remove(LOOP_DIR)
        for k in names + tuple(stubs):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
        importlib.invalidate_caches()
    return mod


def manifest():
    try:
        with open(os.path.join(STAGE_DIR, 'STAGED.json')) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


_cached_by_dir = {}


def _load_from(stage_dir):
    """Import the `quantization` package staged under stage_dir WITHOUT disturbing the product's package of the same
    name: the reference files import each other absolutely (`import quantization`, `import quantization.help_functions
    as qhf`, quant_functions.py:4-5), so for the duration of the import the name `quantization` in sys.modules is pointed
    at the staged package, then the previous entries are put back.  The reference modules keep direct references to
    each other in their globals, so they go on working afterwards."""
    if stage_dir in _cached_by_dir:
        return _cached_by_dir[stage_dir]
    names = ('quantization', 'quantization.quant_functions', 'quantization.help_functions')
    saved = {k: sys.modules.pop(k) for k in names if k in sys.modules}
    sys.path.insert(0, stage_dir)
    importlib.invalidate_caches()
    try:
        mod = importlib.import_module('quantization')
        if not os.path.abspath(getattr(mod, '__file__', '') or '').startswith(stage_dir):
            raise ImportError('imported %r instead of the staged reference package' % (mod,))
        _cached_by_dir[stage_dir] = mod
    finally:
        sys.path.remove(stage_dir)
        for k in names:
            sys.modules.pop(k, None)
        sys.modules.update(saved)
        importlib.invalidate_caches()
    return _cached_by_dir[stage_dir]


def load():
    """The reference's `quantization` package (a module object), imported from the staged bytecode.  Returns None when
    nothing is staged and the reference is absent."""
    if not is_staged() and stage() is None:
        return None
    return _load_from(STAGE_DIR)


# ---- the reference package with the two shape fixes of SURVEY.md section 8c ------------------------------------------
# uniformQuantization_variable.backward ('complicated' STE, quant_functions.py:319-406) raises as shipped for more than
# one bucket (:369-370: a (nb,) index tensor is added to a (nb, 1) one and expanded; :398-400: an (N, 1) product is
# handed to a (N,) view).  The fixes are applied to the source text IN MEMORY before compiling -- the same two string
# replacements tests/golden/gen_golden.py makes -- and only bytecode lands under oracle/_ref/patched/.
PATCHED_DIR = os.path.join(STAGE_DIR, 'patched')
PATCHED_PKG = os.path.join(PATCHED_DIR, 'quantization')
_STE_A = "adder_for_buckets = torch.arange(0, self.bucket_size * total_num_buckets, self.bucket_size).long()"
_STE_B = "(grad_output*(quantized_tensor_unscaled-(tensor-beta)/alpha).view(-1)).view(-1,1))"
_STE_FIXES = [(_STE_A, _STE_A + ".view(-1, 1)"), (_STE_B, _STE_B + ".view(-1)")]


def patched_is_staged():
    return all(os.path.exists(os.path.join(PATCHED_PKG, f + 'c')) for f in FILES)


def stage_patched(ref_root=REF_ROOT, force=False):
    import tempfile
    src_dir = os.path.join(ref_root, 'quantization')
    if not all(os.path.exists(os.path.join(src_dir, f)) for f in FILES):
        return PATCHED_PKG if patched_is_staged() else None
    os.makedirs(PATCHED_PKG, exist_ok=True)
    for f in FILES:
        src = os.path.join(src_dir, f)
        out = os.path.join(PATCHED_PKG, f + 'c')
        if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
            continue
        if f != 'quant_functions.py':
            py_compile.compile(src, cfile=out, dfile='reference/quantization/' + f, doraise=True)
            continue
        with open(src) as fh:
            text = fh.read()
        for old, new in _STE_FIXES:
            if text.count(old) != 1:
                raise RuntimeError('reference %s: expected exactly one occurrence of %r' % (f, old))
            text = text.replace(old, new)
        with tempfile.TemporaryDirectory() as tmp:              # the patched text never lands in the repository
            tmp_src = os.path.join(tmp, f)
            with open(tmp_src, 'w') as fh:
                fh.write(text)
            py_compile.compile(tmp_src, cfile=out, dfile='reference/quantization/' + f + ' (+ the two 8c shape fixes)', doraise=True)
    return PATCHED_PKG


def load_patched():
    """The reference's `quantization` package whose 'complicated' backward runs for any number of buckets."""
    if not patched_is_staged() and stage_patched() is None:
        return None
    return _load_from(PATCHED_DIR)
