"""Network initialisation against the imported reference (tests/golden/init_reference.npz, tools/gen_golden_init.py): the same
seed, the same construction order and the reference's orthogonal-initialisation loop must give the same tensors -- which pins
(i) that every layer the reference re-initialises (all nn.Linear: fc1, fc2, fc, out; not the Conv1d branches) is reached here,
(ii) that the shared feature net is visited once, (iii) the order in which the torch RNG is consumed.  Runs on the CPU (parameter
containers only; nothing is evaluated)."""
import os

import numpy as np
import torch

Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'init_reference.npz'))


def _check(prefix, sd):
    keys = [k.split('::')[1] for k in Z.files if k.startswith(prefix + '::')]
    assert keys == list(sd.keys()), (prefix, keys[:4], list(sd.keys())[:4])
    for k, v in sd.items():
        v = v.double().reshape(-1)
        got = np.concatenate([[v.sum().item(), v.abs().sum().item()], v[:4].numpy(), np.zeros(max(0, 4 - v.numel()))])
        np.testing.assert_allclose(got, Z[f'{prefix}::{k}'], rtol=1e-12, atol=1e-12, err_msg=f'{prefix} {k}')


def test_mansy_nets_initialise_like_the_reference():
    from mansy_immersivevideostreaming_amd.bitrate_selection.models import mansy as mm
    torch.manual_seed(int(Z['seed']))
    fn = mm.FeatureNet(8, 64, 5, 128, device='cpu')
    actor = mm.Actor(fn, feature_dim=1280, hidden_dim=128, action_space=15, device='cpu')
    critic = mm.Critic(fn, feature_dim=1280, hidden_dim=128, device='cpu')
    mm.orthogonal_init(actor, critic)
    ident = mm.QoEIdentifier(mm.QoEIdentifierFeatureNet(8, 64, 5, 15, 128, device='cpu'), feature_dim=1280, hidden_dim=128, device='cpu')
    mm.orthogonal_init(ident)
    _check('mansy/actor', actor.state_dict())
    _check('mansy/critic', critic.state_dict())
    _check('mansy/identifier', ident.state_dict())
    # the loop reached every Linear: their biases are exactly zero, the Conv1d branches keep their default init
    assert float(actor.state_dict()['fc.0.bias'].abs().sum()) == 0.0 and float(actor.state_dict()['feature_net.fc2.0.bias'].abs().sum()) == 0.0
    assert float(actor.state_dict()['feature_net.conv1d2.0.bias'].abs().sum()) > 0.0


def test_simple_rl_nets_initialise_like_the_reference():
    from mansy_immersivevideostreaming_amd.bitrate_selection.models import mansy as mm
    from mansy_immersivevideostreaming_amd.bitrate_selection.models import simple_rl as sr
    torch.manual_seed(int(Z['seed']))
    fn = sr.FeatureNet(8, 64, 5, device='cpu')
    actor = sr.Actor(fn, feature_dim=5 * 128, action_space=15, device='cpu')
    critic = sr.Critic(fn, feature_dim=5 * 128, device='cpu')
    mm.orthogonal_init(actor, critic)
    _check('simple/actor', actor.state_dict())
    _check('simple/critic', critic.state_dict())
