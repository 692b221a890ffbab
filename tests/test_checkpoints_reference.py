"""runs written by harness/checkpoints.py are opened with the REFERENCE's own
ModelManager (model_manager.py) -- load(), load_metadata(name, run), load_model_state_dict(name),
get_num_training_runs -- and the metadata is compared with what the reference's own save_metadata
writes for the same dictionaries.  Runs in a subprocess with /root/reference on the path (the
reference's `helpers.functions` imports the reference's `quantization`, which must not mix with
this repository's package of the same name).  Skipped where /root/reference is absent (GPU box)."""
import os
import pickle
import subprocess
import sys

import pytest
import torch

from harness import checkpoints, models

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'model_manager.py')),
                                reason='the reference checkout is not present')

READER = r'''
import pickle, sys, torch
sys.path.insert(0, %(ref)r)
import model_manager
mm = model_manager.ModelManager(%(manager)r, verbose=False)            # create_new_model_manager=False -> load()
out = {'name': mm.name, 'models': mm.list_models(), 'runs': mm.get_num_training_runs('student'),
       'base': mm.get_model_base_path('student'),
       'meta_last': mm.load_metadata('student'), 'meta_0': mm.load_metadata('student', 0),
       'meta_1': mm.load_metadata('student', 1),
       'sd_keys': sorted(mm.load_model_state_dict('student').keys()),
       'sd_sum': float(sum(v.double().sum() for v in mm.load_model_state_dict('student', 1).values()))}
# the reference's own writer on the same dictionaries: the files must be byte-for-byte what ours wrote
args, info = pickle.load(open(%(raw)r, 'rb'))
args = dict(args, loss_function=torch.nn.functional.cross_entropy)
mm.save_metadata((args, info), %(refmeta)r)
out['ref_meta'] = pickle.load(open(%(refmeta)r, 'rb'))
# and the reference continues the history on top of ours
mm.add_new_model('other', %(other)r, {'k': 1.5})
out['models_after'] = sorted(mm.list_models())
pickle.dump(out, open(%(out)r, 'wb'))
'''


def test_reference_model_manager_reads_our_runs(tmp_path):
    torch.manual_seed(0)
    net = models.student()
    store = checkpoints.RunStore(str(tmp_path / 'manager'), 'cifar10', verbose=False, create=True)
    store.add_new_model('student', str(tmp_path / 'student'), {'useBatchNorm': True, 'spec': {'conv': [75, 50]}})
    raw_args = {'numBits': 4, 'bucket_size': 256, 'quantizeWeights': True, 'initial_learning_rate': 1e-3,
                'backprop_quantization_style': None, 'learning_rate_style': 'generic', 'epochs': (1, 2),
                'odd': {1: 2}}
    info1 = {'numEpochsTrained': 1, 'lossSaved': [2.0], 'predictionAccuracy': [0.31]}
    info2 = {'numEpochsTrained': 3, 'lossSaved': [2.0, 1.5, 1.2], 'predictionAccuracy': [0.31, 0.4, 0.47]}
    args_with_fn = dict(raw_args, loss_function=torch.nn.functional.cross_entropy)
    store.append_run('student', net.state_dict(), args_with_fn, info1)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.25)
    store.append_run('student', net.state_dict(), args_with_fn, info2)
    with open(tmp_path / 'raw.pkl', 'wb') as f:
        pickle.dump((raw_args, info2), f)
    script = READER % dict(ref=REF, manager=str(tmp_path / 'manager'), raw=str(tmp_path / 'raw.pkl'),
                           refmeta=str(tmp_path / 'ref_meta'), other=str(tmp_path / 'other'), out=str(tmp_path / 'out.pkl'))
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    p = subprocess.run([sys.executable, '-c', script], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    with open(tmp_path / 'out.pkl', 'rb') as f:
        out = pickle.load(f)
    assert out['name'] == 'cifar10' and out['models'] == ['student'] and out['runs'] == 2
    assert out['base'] == str(tmp_path / 'student')
    import re
    strip = lambda m: [{k: re.sub(r' at 0x[0-9a-f]+', '', v) if isinstance(v, str) else v for k, v in d.items()} for d in m]
    # identical up to the function's address inside its repr (two different processes)
    assert strip(out['meta_last']) == strip(out['ref_meta']), 'our metadata file differs from what the reference writes'
    assert out['meta_last'][1] == info2 and out['meta_1'][1] == info1
    assert out['meta_last'][0]['odd'] == repr({1: 2}) and out['meta_last'][0]['epochs'] == (1, 2)
    assert out['meta_last'][0]['loss_function'].startswith('Name: cross_entropy. Repr: ')
    assert out['meta_0'][0] == {'useBatchNorm': True, 'spec': repr({'conv': [75, 50]})} and out['meta_0'][1] == {}
    assert out['sd_keys'] == sorted(net.state_dict().keys())
    # run 1 holds the weights before the +0.25 shift, the last run after it
    sd_last = checkpoints.RunStore(str(tmp_path / 'manager')).load_model_state_dict('student')
    want_last = float(sum(v.double().sum() for v in net.state_dict().values()))
    assert abs(float(sum(v.double().sum() for v in sd_last.values())) - want_last) < 1e-6
    assert abs(out['sd_sum'] - want_last) > 1.0
    # the reference appended to OUR manager file; our reader opens what it wrote
    again = checkpoints.RunStore(str(tmp_path / 'manager'))
    assert sorted(again.saved_models) == ['other', 'student'] == out['models_after']
    assert again.load_metadata('other', 0)[0] == {'k': 1.5}
