"""Host mirror of bitrate_selection/utils/mansy_utils.py:9-49 (train_identifier, calculate_indentifier_reward).
behavior_cloning_pretraining (:52-93) is out of scope (README: no gain)."""
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
