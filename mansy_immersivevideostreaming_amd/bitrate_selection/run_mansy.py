#!/usr/bin/env python3
"""CLI counterpart of bitrate_selection/run_mansy.py (flags :283-337, net/optimiser/policy construction :205-251, train()
:25-140, test() :143-176, checkpoint / result directory naming :58-61,193-199) on the HIP engine and the vectorised
device environment.

  python -m mansy_immersivevideostreaming_amd.bitrate_selection.run_mansy --train --test --epochs 10 --step-per-epoch 4096 \\
      --step-per-collect 4096 --lr 0.0005 --batch-size 512 --train-dataset Jin2022 --test-dataset Jin2022 --test-on-seen \\
      --qoe-test-ids 0 1 2 3 --lamb 0.5 --train-identifier --use-identifier --device cuda:0 [--train-num 256] [--config ../config.yml]

`--train-num` (environments stepped per launch) is honoured here -- the reference forces 1 (run_mansy.py:37); with 1 the
episode order equals the reference's.  `--bc` pre-trains on the demonstrations written by run_expert (:255-274)
before PPO training; `--init-from-bc` starts from the behaviour-cloning checkpoints (:73-83)."""
import argparse
import os
import pickle
import random
import sys

import numpy as np
import torch
from torch.distributions import Categorical

from .. import _lib
from .envs.mansy_env import EnvTables, MANSYVecEnv
from .models.mansy import Actor, Critic, FeatureNet, QoEIdentifier, QoEIdentifierFeatureNet, orthogonal_init
from .models.mansy_ppo import PPOPolicy, VecCollector
from .models.mansy_trainer import OnpolicyTrainer, run_episodes, write_episode_log
from .utils.common import get_config_from_yml, read_log_file
from .utils.mansy_utils import behavior_cloning_pretraining, load_demonstrations


def train(args, config, policy, qoe_weights, identifier, identifier_optimizer, models_dir, policy_bc_path=None, identifier_bc_path=None):
    train_log_path = os.path.join(models_dir, 'train_log.csv')
    valid_log_path = os.path.join(models_dir, 'valid_log.csv')
    for p in (train_log_path, valid_log_path):
        if os.path.exists(p):
            os.remove(p)
    dev = args.device
    t_train = EnvTables.from_dataset(config, args.train_dataset, args.network_dataset, 'train', qoe_weights, dev, seed=args.seed,
                                     use_identifier=args.use_identifier)
    t_valid = EnvTables.from_dataset(config, args.train_dataset, args.network_dataset, 'valid', qoe_weights, dev, seed=args.seed,
                                     use_identifier=args.use_identifier)
    args.test_num = len(qoe_weights)
    args.episode_per_test = t_valid.n_sample
    print('Training num:', args.train_num)
    print('Test num:', args.test_num)
    print('Episode per test:', args.episode_per_test)
    print('Training QoE weights:', qoe_weights)
    train_env = MANSYVecEnv(t_train, args.train_num, seed=args.seed, worker_num=args.train_num)
    valid_env = MANSYVecEnv(t_valid, args.test_num, seed=args.seed, worker_num=args.test_num)
    checkpoint_path = os.path.join(models_dir, 'checkpoint.pth')
    identifier_checkpoint_path = os.path.join(models_dir, 'identifier_checkpoint.pth')
    best_policy_path = os.path.join(models_dir, 'best_policy.pth')
    best_identifier_path = os.path.join(models_dir, 'best_identifier.pth')
    if args.resume:
        for path, mod, name in ((checkpoint_path, policy, 'agent'), (identifier_checkpoint_path, identifier, 'identifier')):
            if os.path.exists(path):
                mod.load_state_dict(torch.load(path, map_location=args.device))
                print(f'Successfully loaded {name} from:', path)
            else:
                print(f'Failed to load {name}:', path, 'no such file')
    elif args.init_from_bc:
        for path, mod, name in ((policy_bc_path, policy, 'agent'), (identifier_bc_path, identifier, 'identifier')):
            if path and os.path.exists(path):
                mod.load_state_dict(torch.load(path, map_location=args.device))
                print(f'Successfully init {name} from behavior cloning:', path)
            else:
                print(f'Failed to load {name}:', path, 'no such file')

    def save_best_fn(policy):
        torch.save(policy.state_dict(), best_policy_path)
        torch.save(identifier.state_dict(), best_identifier_path)
        print('Best policy save at ' + best_policy_path)
        print('Best identifier save at ' + best_identifier_path)

    def stop_fn(mean_rewards):
        return mean_rewards >= args.reward_threshold

    def save_checkpoint_fn(epoch, env_step, gradient_step):
        torch.save(policy.state_dict(), checkpoint_path)
        torch.save(identifier.state_dict(), identifier_checkpoint_path)
        print('Checkpoint saved at ' + checkpoint_path)
        print('Identifier checkpoint saved at ' + identifier_checkpoint_path)
        return checkpoint_path

    trainer = OnpolicyTrainer(policy, VecCollector(policy, train_env, seed=args.seed), VecCollector(policy, valid_env, seed=args.seed + 1),
                              args.epochs, args.step_per_epoch, args.repeat_per_collect, episode_per_test=args.episode_per_test,
                              batch_size=args.batch_size, step_per_collect=args.step_per_collect, stop_fn=stop_fn, save_best_fn=save_best_fn,
                              save_checkpoint_fn=save_checkpoint_fn, args=args, identifier=identifier,
                              identifier_optimizer=identifier_optimizer, test_log=(valid_log_path, t_valid, qoe_weights),
                              train_log=(train_log_path, t_train, qoe_weights))
    for epoch, epoch_stat, info in trainer:
        print(f'Epoch: {epoch}')
        print('loss:', epoch_stat.get('loss'), ' --- ', 'loss/clip:', epoch_stat.get('loss/clip'), ' --- ', 'loss/vf:', epoch_stat.get('loss/vf'),
              ' --- ', 'loss/ent:', epoch_stat.get('loss/ent'))
    return trainer


def test(args, config, policy, qoe_weights, identifier, models_dir, results_dir):
    test_log_path = os.path.join(results_dir, 'results.csv')
    if os.path.exists(test_log_path):
        os.remove(test_log_path)
    tables = EnvTables.from_dataset(config, args.test_dataset, args.network_dataset, 'test', qoe_weights, args.device, seed=args.seed)
    policy_path = args.policy_path or os.path.join(models_dir, 'best_policy.pth')
    if os.path.exists(policy_path):
        policy.load_state_dict(torch.load(policy_path, map_location=args.device))
        print('Successfully loaded agent from:', policy_path)
    else:
        raise FileExistsError(f'File not exist: {policy_path}')
    n = tables.n_sample
    n_env = min(args.test_envs, n)
    venv = MANSYVecEnv(tables, n_env, seed=0, worker_num=n_env)       # env i plays samples i, i+n_env, ... : every combination once
    first = {}
    with torch.no_grad():
        while len(first) < n:                                         # lock-step envs finish at slightly different times
            run_episodes(policy, venv, n_env, seed=args.seed + len(first), reset=not first)
            for r in venv.pop_episode_log():                          # keep the first completion of every catalogue entry
                first.setdefault(int(r[0]), r)
    ordered = [first[k] for k in sorted(first)]
    write_episode_log(test_log_path, tables, qoe_weights, ordered)
    read_log_file(test_log_path, verbose=args.verbose_table)
    print('Results saved at:', test_log_path)
    return ordered


def run(args, config):
    if getattr(args, 'precision', 'f32') not in _lib.PRECISIONS:
        raise _lib.MansyError(f'unknown --precision {args.precision!r}: one of f32, bf16, bf16x3, bf16x6')
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)
    random.seed(args.seed)
    if args.qoe_train_ids is None:
        args.qoe_train_ids = list(range(len(config.qoe_split['train'])))
    split = 'train' if args.test_on_seen else 'test'
    if args.qoe_test_ids is None:
        args.qoe_test_ids = list(range(len(config.qoe_split[split])))
    prefix = f'epochs_{args.epochs}_bs_{args.batch_size}_lr_{args.lr}_gamma_{args.gamma}_seed_{args.seed}_ent_{args.ent_coef}_useid_{args.use_identifier}' \
             f'_lambda_{args.lamb}_ilr_{args.identifier_lr}_iur_{args.identifier_update_round}_bc_{args.bc or args.init_from_bc}'
    models_dir = os.path.join(config.bs_models_dir, args.model, args.train_dataset + '_' + args.network_dataset,
                              'qoe' + '_'.join(map(str, args.qoe_train_ids)), prefix)
    seen = 'seen_qoe' if args.test_on_seen else 'unseen_qoe'
    results_dir = os.path.join(config.bs_results_dir, args.model, args.test_dataset + '_' + args.network_dataset,
                               seen + '_'.join(map(str, args.qoe_test_ids)), prefix)
    os.makedirs(models_dir, exist_ok=True)
    os.makedirs(results_dir, exist_ok=True)
    # run_mansy.py:205-251
    feature_dim = args.hidden_dim * 10
    feature_net = FeatureNet(config.past_k, config.tile_total_num, len(config.video_rates), args.hidden_dim, device=args.device)
    actor = Actor(feature_net, feature_dim=feature_dim, hidden_dim=args.hidden_dim, action_space=config.action_space, device=args.device)
    critic = Critic(feature_net, feature_dim=feature_dim, hidden_dim=args.hidden_dim, device=args.device)
    orthogonal_init(actor, critic)
    ac_params = list(actor.parameters()) + [p for n, p in critic.named_parameters() if not n.startswith('feature_net.')]
    optimizer = torch.optim.Adam(ac_params, lr=args.lr, weight_decay=args.weight_decay)
    identifier_feature_net = QoEIdentifierFeatureNet(config.past_k, config.tile_total_num, len(config.video_rates), config.action_space,
                                                     args.hidden_dim, device=args.device)
    identifier = QoEIdentifier(identifier_feature_net, feature_dim=feature_dim, hidden_dim=args.hidden_dim, device=args.device)
    orthogonal_init(identifier)
    identifier_optimizer = torch.optim.Adam(identifier.parameters(), lr=args.identifier_lr, weight_decay=args.weight_decay)
    policy = PPOPolicy(actor, critic, optimizer, lambda logits: Categorical(logits=logits), discount_factor=args.gamma,
                       max_grad_norm=args.max_grad_norm, eps_clip=args.eps_clip, vf_coef=args.vf_coef, ent_coef=args.ent_coef,
                       reward_normalization=args.rew_norm, advantage_normalization=args.norm_adv, recompute_advantage=args.recompute_adv,
                       dual_clip=args.dual_clip, value_clip=args.value_clip, gae_lambda=args.gae_lambda, action_space=config.action_space,
                       action_scaling=False, args=args, identifier=identifier, identifier_optim=identifier_optimizer).to(args.device)
    policy.engine.precision = getattr(args, 'precision', 'f32')      # carried into every engine call (no process-wide mode)
    out = {'policy': policy, 'models_dir': models_dir, 'results_dir': results_dir}      # (the reference's run() returns nothing: extra, for callers / tests)
    if args.train:
        bc_file_prefix = f'bc_ms_{args.bc_max_steps}_ims_{args.bc_identifier_max_steps}_ilr_{args.identifier_lr}_iur_{args.identifier_update_round}'
        policy_bc_path = os.path.join(models_dir, bc_file_prefix + '_policy.pth')
        identifier_bc_path = os.path.join(models_dir, bc_file_prefix + '_identifier.pth')
        if args.bc:          # run_mansy.py:260-274: demonstrations written by run_expert --train --valid
            demos_dir = os.path.join(config.bs_models_dir, 'expert', args.train_dataset + '_' + args.network_dataset,
                                     'qoe' + '_'.join(map(str, args.qoe_train_ids)))
            train_demos_path = os.path.join(demos_dir, 'train_demonstrations.pkl')
            valid_demos_path = os.path.join(demos_dir, 'valid_demonstrations.pkl')
            assert os.path.exists(train_demos_path) and os.path.exists(valid_demos_path)
            train_demos = list(load_demonstrations(train_demos_path).values())       # this build's or the reference's file format
            valid_demos = list(load_demonstrations(valid_demos_path).values())
            behavior_cloning_pretraining(args, policy, identifier, optimizer, identifier_optimizer, train_demos, valid_demos,
                                         max_steps=args.bc_max_steps, valid_per_step=args.bc_valid_per_step,
                                         identifier_max_steps=args.bc_identifier_max_steps,
                                         identifier_update_round=args.identifier_update_round, policy_save_path=policy_bc_path,
                                         identifier_save_path=identifier_bc_path)
        qoe_weights = [config.qoe_split['train'][i] for i in args.qoe_train_ids]
        out['trainer'] = train(args, config, policy, qoe_weights, identifier, identifier_optimizer, models_dir, policy_bc_path, identifier_bc_path)
    if args.test:
        qoe_weights = [config.qoe_split[split][i] for i in args.qoe_test_ids]
        out['test_records'] = test(args, config, policy, qoe_weights, identifier, models_dir, results_dir)
    return out


# the reference's command line (run_mansy.py:284-337) as data: (flag, type-or-None for store_true, default)
_T, _F = 'store_true', None
_FLAGS = [
    ('--task', str, 'mansy'), ('--reward-threshold', float, 500000.0), ('--seed', int, 5), ('--buffer-size', int, 1000000),
    ('--lr', float, 5e-4), ('--weight-decay', float, 1e-2), ('--gamma', float, 0.95), ('--epochs', int, 1000),
    ('--step-per-epoch', int, 4096), ('--step-per-collect', int, 4096), ('--episode-per-collect', int, 10),
    ('--repeat-per-collect', int, 2), ('--batch-size', int, 512), ('--train-num', int, 1), ('--test-num', int, None),
    ('--episode-per-test', int, None), ('--device', str, 'cuda:0'), ('--logdir', str, 'log_tensorboard'), ('--vf-coef', float, 0.5),
    ('--ent-coef', float, 0.02), ('--eps-clip', float, 0.2), ('--max-grad-norm', float, 1), ('--gae-lambda', float, 0.95),
    ('--rew-norm', int, 1), ('--dual-clip', float, None), ('--value-clip', int, 1), ('--norm-adv', int, 1), ('--recompute-adv', int, 0),
    ('--resume', _T, _F), ('--save-interval', int, 4), ('--model', str, 'mansy'), ('--hidden-dim', int, 128),
    ('--identifier-lr', float, 1e-4), ('--identifier-update-round', int, 2), ('--identifier-epochs', int, 1000), ('--lamb', float, 0.5),
    ('--train', _T, _F), ('--train-identifier', _T, _F), ('--use-identifier', _T, _F), ('--test', _T, _F), ('--test-on-seen', _T, _F),
    ('--train-dataset', str, 'Jin2022'), ('--test-dataset', str, 'Jin2022'), ('--network-dataset', str, '4G'),
    ('--policy-path', str, None), ('--bc', _T, _F), ('--bc-max-steps', int, 150), ('--bc-valid-per-step', int, 50),
    ('--bc-identifier-max-steps', int, 150), ('--init-from-bc', _T, _F),
    # additions of this build
    ('--config', str, None), ('--test-envs', int, 256), ('--verbose-table', _T, _F),
    # precision of the dense products: f32 (exact fp32 MFMA, the parity mode) | bf16x3 | bf16x6 (split-bf16 MFMA; BASELINE configs[4]) | bf16 (one product: perf mode)
    ('--precision', str, 'f32'),
]


def build_parser():
    parser = argparse.ArgumentParser(description='MANSY PPO bitrate selection on MI355X')
    for flag, kind, default in _FLAGS:
        if kind == _T:
            parser.add_argument(flag, action='store_true')
        else:
            parser.add_argument(flag, type=kind, default=default)
    for flag in ('--qoe-train-ids', '--qoe-test-ids'):
        parser.add_argument(flag, type=int, nargs='*')
    return parser


def main(argv=None):
    args = build_parser().parse_known_args(argv)[0]          # unknown flags are ignored, as in the reference (:340)
    print(args)
    config = get_config_from_yml(args.config)
    return run(args, config)


if __name__ == '__main__':
    main()
