"""The library establishes its own launch configuration: HIP_FORCE_DEV_KERNARG=1 is set when
`mmnas` / `mmnas_amd` is imported before HIP initialises and the user has not set it, an explicit setting is respected,
an import after HIP is up warns once, and `ops.runtime_config()` reports what the process runs with.  Every case runs in a
fresh child process (the variable is read by the HIP runtime once per process)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(code, env_kernarg=None):
    env = dict(os.environ)
    env.pop('HIP_FORCE_DEV_KERNARG', None)
    if env_kernarg is not None:
        env['HIP_FORCE_DEV_KERNARG'] = env_kernarg
    p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1]), p.stderr


IMPORT_LINES = ('import json, os\n'
                'from mmnas.model.hygr_vqa import Net_Search\n'      # search_vqa.py:21: the reference's own import line
                'from mmnas_amd import ops\n'
                'c = ops.runtime_config()\n'
                'print(json.dumps({"env": os.environ.get("HIP_FORCE_DEV_KERNARG"), "value": c["hip_force_dev_kernarg"],\n'
                '                  "source": c["hip_force_dev_kernarg_source"], "abi": c["abi_version"], "hip": c["hip_initialised"]}))\n')


def test_import_sets_the_variable_when_the_user_has_not():
    d, _ = _child(IMPORT_LINES)
    assert d == {'env': '1', 'value': '1', 'source': 'set_by_library', 'abi': 1, 'hip': False}


def test_an_explicit_setting_is_respected():
    d, _ = _child(IMPORT_LINES, env_kernarg='0')
    assert d['env'] == '0' and d['value'] == '0' and d['source'] == 'inherited'


@pytest.mark.gpu
def test_dropin_statements_in_a_child_without_the_variable_run_with_it_set():
    """The reference's statement order: import the model, build it, move it to the GPU, run a step -- the variable is in the
    process environment before the process's first HIP call, so the runtime that initialises at `.cuda()` reads 1."""
    code = ('import json, os, numpy as np, torch\n'
            'assert "HIP_FORCE_DEV_KERNARG" not in os.environ\n'
            'from mmnas.model.full_vqa import Net_Full\n'
            'at_first_hip_call = os.environ.get("HIP_FORCE_DEV_KERNARG")\n'
            'assert not torch.cuda.is_initialized()\n'
            'from tests.golden import cases\n'
            'c = cases.net_case("vqa", "mmnas_vqa", 20261002, HSIZE=128, B=3, Sx=6, Sy=9)\n'
            'init = {"token_size": c["token_size"], "ans_size": c["ans_size"], "pretrained_emb": np.zeros((c["token_size"], c["cfg"].WORD_EMBED_SIZE), np.float32)}\n'
            'net = Net_Full(c["cfg"], init).cuda().train()\n'
            'pred = net(tuple(torch.from_numpy(a).cuda() for a in c["inputs"]))\n'
            'torch.nn.functional.binary_cross_entropy_with_logits(pred, torch.from_numpy(c["target"]).cuda(), reduction="sum").backward()\n'
            'torch.cuda.synchronize()\n'
            'from mmnas_amd import ops\n'
            'r = ops.runtime_config()\n'
            'print(json.dumps({"first": at_first_hip_call, "value": r["hip_force_dev_kernarg"], "source": r["hip_force_dev_kernarg_source"],\n'
            '                  "loaded": r["lib_loaded"], "finite": bool(torch.isfinite(pred).all())}))\n')
    d, _ = _child(code)
    assert d == {'first': '1', 'value': '1', 'source': 'set_by_library', 'loaded': True, 'finite': True}


@pytest.mark.gpu
def test_import_after_hip_initialised_warns_once():
    code = ('import json, logging, os, torch\n'
            'logging.basicConfig(level=logging.WARNING)\n'
            'torch.zeros(1, device="cuda")\n'
            'from mmnas_amd import ops\n'
            'r = ops.runtime_config()\n'
            'print(json.dumps({"value": r["hip_force_dev_kernarg"], "source": r["hip_force_dev_kernarg_source"]}))\n')
    d, err = _child(code)
    assert d == {'value': None, 'source': 'unset_after_hip_init'}
    assert err.count('HIP_FORCE_DEV_KERNARG is not set') == 1, err[-2000:]
