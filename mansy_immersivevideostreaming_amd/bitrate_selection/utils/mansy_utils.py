"""Host mirror of bitrate_selection/utils/mansy_utils.py: train_identifier (:9-39), calculate_indentifier_reward (:42-49),
behavior_cloning_pretraining (:52-93)."""
import io
import pickle
from random import choice

import numpy as np
import torch

from ..envs.mansy_env import OBS_LD, OBS_SLICES
from ..models.mansy import obs_to_tensor, _engine_of


# ---------------------------------------------------------------------------------------------- demonstration files
class _TsStub:
    """Stand-in for any tianshou class met while unpickling a reference-written demonstration file: keeps the pickled state."""

    def __init__(self, *a, **k):
        self.__dict__.update(k)

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'_state': state})

    def __getitem__(self, k):
        return self.__dict__[k]

    def keys(self):
        return self.__dict__.keys()


class _DemoUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split('.')[0] == 'tianshou':
            return _TsStub
        return super().find_class(module, name)


def _field(node, key):
    return node[key] if isinstance(node, dict) else getattr(node, key)


def _as_demo(obj):
    """One demonstration in this build's form {'obs' [len, 780] float32, 'act' [len] int32, 'done' [len] bool}.  Accepts that form
    as is, or what the reference's run_expert.py:35-41 pickles: a tianshou ReplayBuffer filled by add(Batch(obs=<observation
    dict>, act, rew, done, obs_next, info)) -- T2, release 0.4.8 layout: the transitions live in `_meta` (a Batch whose `obs` is a
    Batch of stacked arrays, one per observation key), `_size` of them are valid."""
    if isinstance(obj, dict) and 'obs' in obj and 'act' in obj and not isinstance(obj['obs'], (dict, _TsStub)):
        return obj
    meta = _field(obj, '_meta') if not isinstance(obj, dict) else obj
    size = int(_field(obj, '_size')) if not isinstance(obj, dict) and '_size' in obj.__dict__ else len(np.asarray(_field(meta, 'act')))
    obs = _field(meta, 'obs')
    rows = np.zeros((size, OBS_LD), np.float32)
    for k, (a, b, shape) in OBS_SLICES.items():
        try:
            v = _field(obs, k)
        except (KeyError, AttributeError):
            continue
        rows[:, a:b] = np.asarray(v, np.float32)[:size].reshape(size, -1)
    act = np.asarray(_field(meta, 'act'))[:size].astype(np.int32)
    done = np.asarray(_field(meta, 'done'))[:size].astype(bool) if 'done' in (meta.keys() if hasattr(meta, 'keys') else ()) else np.zeros(size, bool)
    return {'obs': rows, 'act': act, 'done': done}


def load_demonstrations(path):
    """{(video, user, trace, qoe_weight): demonstration} from a demonstration file written by this build's run_expert (plain
    dicts) or by the reference's (tianshou ReplayBuffers; read without tianshou through stand-in classes).  The reference
    layout is restated from the 0.4.8 release and could not be checked against a reference-written file here (tianshou is not
    installable): PARITY UNPINNED for that branch; tests/test_host_demos.py exercises it on a file with the same class paths."""
    with open(path, 'rb') as f:
        raw = _DemoUnpickler(io.BytesIO(f.read())).load()
    return {k: _as_demo(v) for k, v in raw.items()}


def train_identifier(identifier, identifier_optim, tracjetory, update_round=2, policy=None):
    """`tracjetory` is the RolloutBuffer of the collector.  Uses the policy's fused identifier step when given, else a
    stand-alone engine around `identifier` (lr / weight decay read from `identifier_optim`)."""
    if policy is None:
        raise ValueError('pass policy=PPOPolicy(...): the identifier shares its engine')
    if identifier_optim is not None:
        policy.identifier_optim = identifier_optim
    return policy.train_identifier(tracjetory, update_round=update_round)


def calculate_indentifier_reward(identifier, state, action_one_hot):
    """1 - MSE(identifier(state, action_one_hot), state['qoe_weight']) for one (un-batched) or many transitions."""
    eng = _engine_of(identifier)
    obs = obs_to_tensor(state, eng.device)
    if action_one_hot is not None and not torch.is_tensor(state):
        obs[:, 748:763] = torch.as_tensor(np.asarray(action_one_hot, np.float32).reshape(obs.shape[0], -1), device=obs.device)
    pred = eng.identifier_forward(obs)
    r = 1.0 - ((pred - obs[:, 745:748]) ** 2).mean(dim=-1)
    return r.cpu().numpy() if r.numel() > 1 else r.cpu().numpy().reshape(())


class _DemoBuffer:
    """A demonstration (run_expert's {'obs' [len,780], 'act' [len]} dict) behind the two accessors train_identifier needs."""

    def __init__(self, obs):
        self.obs, self.filled = obs.unsqueeze(1), obs.shape[0]

    def __len__(self):
        return self.filled


def _demo_tensors(demo, device):
    return (torch.as_tensor(np.ascontiguousarray(demo['obs'], dtype=np.float32), device=device),
            torch.as_tensor(np.ascontiguousarray(demo['act']).astype(np.int32), device=device))


def behavior_cloning_pretraining(args, policy, identifier, policy_optim, identifier_optim, train_demos, valid_demos, max_steps, valid_per_step,
                                 identifier_max_steps, identifier_update_round, policy_save_path, identifier_save_path):
    """utils/mansy_utils.py:52-93 on run_expert's demonstrations: per step one random demonstration, loss = CE(logits, expert
    action) - 0.1 * mean entropy, Adam step on the actor (+ shared feature net); every `valid_per_step` steps the mean
    validation cross entropy decides the best checkpoint; the identifier is trained on the same demonstration for the
    first `identifier_max_steps` steps.  (The reference also samples an action inside policy(samples); that only advances
    torch's RNG and is not reproduced.)"""
    dev = policy.engine.device
    if policy_optim is not None:
        policy.optim = policy_optim
    best_loss, best_step = float('inf'), 0
    for i in range(max_steps):
        demo = choice(train_demos)
        obs, act = _demo_tensors(demo, dev)
        loss = policy.bc_step(obs, act, ent_coef=0.1, train=True)[0]
        print(f'BC (Training): loss={loss.item()} ({i + 1}/{max_steps})')
        if i % valid_per_step == 0:
            valid_loss = 0.
            for d in valid_demos:
                vo, va = _demo_tensors(d, dev)
                valid_loss += policy.bc_step(vo, va, train=False)[1].item()
            valid_loss = valid_loss / len(valid_demos)
            if best_loss > valid_loss:
                best_loss, best_step = valid_loss, i
                torch.save(policy.state_dict(), policy_save_path)
            print(f'BC (Validation): valid loss={valid_loss} - best loss={best_loss} at step {best_step}')
        if i < identifier_max_steps:
            train_identifier(identifier, identifier_optim, _DemoBuffer(obs), identifier_update_round, policy=policy)
            torch.save(identifier.state_dict(), identifier_save_path)
    return best_loss, best_step
