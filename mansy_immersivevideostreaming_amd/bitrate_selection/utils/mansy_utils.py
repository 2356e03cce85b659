"""Host mirror of bitrate_selection/utils/mansy_utils.py: train_identifier (:9-39), calculate_indentifier_reward (:42-49),
behavior_cloning_pretraining (:52-93)."""
from random import choice

import numpy as np
import torch

from ..models.mansy import obs_to_tensor, _engine_of


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
