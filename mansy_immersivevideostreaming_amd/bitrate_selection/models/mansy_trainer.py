"""On-policy trainer counterpart of bitrate_selection/models/mansy_trainer.py (:18-95 `__next__`, :162-177
`policy_update_fn`) on top of the vectorised collector: per epoch  collect step_per_collect -> [train identifier] ->
relabel + PPO update -> reset buffer  until step_per_epoch, then checkpoint, validation episodes, best-model save.
Order of operations and the callback signatures are the reference's (tianshou 0.4.8 BaseTrainer semantics, T2), including two
things that decide which files exist after a run:
  * iterating the trainer first calls reset() (tianshou BaseTrainer.__iter__ / reset): ONE test of the still-untrained policy
    (epoch 0) seeds best_reward / best_epoch, and save_best_fn is called once -- best_policy.pth exists before epoch 1;
  * the reference's own `__next__` (mansy_trainer.py:24-27) stops on `epoch >= max_epoch` from the second iteration on, so
    `--epochs E` runs max(1, E - 1) epochs (E = 1 runs one: the shipped example `epochs_1_...`)."""
import time

import numpy as np
import torch

from .mansy_ppo import RolloutBuffer, VecCollector


def write_episode_log(log_path, tables, qoe_weights, records):
    """mansy_env.py:271-290 rows from device episode records (sorted by catalogue position for a deterministic file)."""
    import os
    if not len(records):
        return
    if not os.path.exists(log_path):
        with open(log_path, 'w', encoding='utf-8') as f:
            f.write('video,user,trace,qoe_w1,qoe_w2,qoe_w3,qoe,qoe1,qoe2,qoe3\n')
    with open(log_path, 'a', encoding='utf-8') as f:
        for sid, _, n, sq, s1, s2, s3, qi in records:
            w = np.array(qoe_weights[int(qi)], dtype=np.float32)
            video, user, trace = tables.ids[3][int(sid)] if tables.ids is not None else (int(sid), -1, -1)
            f.write(f'{video},{user},{trace},{w[0]},{w[1]},{w[2]},{round(sq / n / sum(w), 5)},{round(s1 / n, 5)},{round(s2 / n, 5)},'
                    f'{round(s3 / n, 5)}\n')


def run_episodes(policy, venv, n_episode, seed=0, reset=True, log=None):
    """Test collector: step `venv` with sampled actions until n_episode episodes have finished; returns their returns.
    reset=False continues from the environments' current state.
    log: a list that receives the finished episodes' device records in the order the reference's CSV rows appear -- by vector step, and
    within a step by environment index (tianshou's DummyVectorEnv steps its workers in order and each MANSYEnv appends its row when its
    episode ends, mansy_env.py:230-232,271-290); the device appends records of ONE step in arbitrary order."""
    eng = policy.engine
    N = venv.n_env
    obs = venv.reset() if reset else venv.obs
    ret = torch.zeros(N, dtype=torch.float64, device=obs.device)
    done_returns = []
    step = 0
    while len(done_returns) < n_episode:
        _, _, act, _ = eng._policy_forward(obs, False, True, None, seed, step * N)
        obs, rew, done, _ = venv.step(act, auto_reset=True)
        ret += rew.double()
        d = done.bool()
        if d.any():
            done_returns += ret[d].cpu().tolist()
            ret[d] = 0
            if log is not None:
                rec = venv.pop_episode_log()
                log.extend(rec[np.argsort(rec[:, 1], kind='stable')])
        step += 1
    return np.array(done_returns[:n_episode])


class OnpolicyTrainer:
    def __init__(self, policy, train_collector, test_collector, max_epoch, step_per_epoch, repeat_per_collect, episode_per_test, batch_size,
                 step_per_collect=None, stop_fn=None, save_best_fn=None, save_checkpoint_fn=None, logger=None, args=None, identifier=None,
                 identifier_optimizer=None, test_log=None, train_log=None, verbose=True, **kwargs):
        self.policy, self.train_collector, self.test_collector = policy, train_collector, test_collector
        self.max_epoch, self.step_per_epoch, self.repeat_per_collect = max_epoch, step_per_epoch, repeat_per_collect
        self.episode_per_test, self.batch_size, self.step_per_collect = episode_per_test, batch_size, step_per_collect
        self.stop_fn, self.save_best_fn, self.save_checkpoint_fn = stop_fn, save_best_fn, save_checkpoint_fn
        self.args, self.identifier, self.identifier_optimizer = args, identifier, identifier_optimizer
        self.test_log = test_log          # (log_path, tables, qoe_weights) for the validation CSV
        self.train_log = train_log        # the same for the training CSV (rows appended after every collect)
        self.history = []                 # per collect: env_step, n/ep, len -- what tianshou's logger writes as train/episode, train/length
        self.verbose = verbose
        self.epoch, self.iter_num, self.env_step, self.gradient_step = 0, 0, 0, 0
        self.best_reward, self.best_reward_std, self.best_epoch = -np.inf, 0.0, -1
        self.stop_fn_flag = False
        N = train_collector.venv.n_env
        self.buffer = RolloutBuffer(max(1, step_per_collect // N), N, train_collector.venv.device)
        if identifier_optimizer is not None:
            policy.identifier_optim = identifier_optimizer
        # T2: tianshou's Collector.__init__ ends in reset() -> reset_env(): every environment has been reset ONCE before the trainer touches
        # it.  For the training collector that reset is the start of the first collect (VecCollector.reset_env at the first collect); for
        # the test collector it moves each environment one entry along its catalogue walk (worker_id += worker_num, mansy_env.py:100-101)
        # before test_episode resets again -- the shipped valid_log.csv starts at catalogue entries 5, 6, 7, 4, not 1, 2, 3, 0.
        tv = getattr(test_collector, 'venv', None)
        if tv is not None and hasattr(tv, 'reset') and not getattr(test_collector, '_constructed_reset', False):
            tv.reset()
            test_collector._constructed_reset = True
        self.start_time = time.time()

    def reset(self):
        """T2: tianshou 0.4.8 BaseTrainer.reset(): statistics cleared, an initial test at epoch 0 (best_epoch = 0, best_reward = its
        reward), then save_best_fn(policy) unconditionally."""
        self.epoch, self.iter_num, self.env_step = 0, 0, 0
        self.stop_fn_flag = False
        self.start_time = time.time()
        if self.test_collector is not None:
            self.best_epoch = -1
            self.test_step(save=False)
        if self.save_best_fn:
            self.save_best_fn(self.policy)

    def __iter__(self):
        self.reset()
        return self

    def test_step(self, save=True):
        # T2: tianshou test_episode = collector.reset_env() -> collect(n_episode) -- which resets every finished environment and, having
        # collected by episodes, ends in ANOTHER reset_env() (Collector.collect's closing `if n_episode: self.reset_env()`).  Each test thus moves an
        # environment 2 + episodes-played entries along its catalogue walk; the shipped valid_log.csv (second test starts at entries
        # 13, 14, 15, 12) holds exactly that.
        venv = self.test_collector.venv
        if self.test_log is not None:
            venv.pop_episode_log()
            records = []
            rets = run_episodes(self.policy, venv, self.episode_per_test, seed=self.test_collector.seed, log=records)
            write_episode_log(self.test_log[0], self.test_log[1], self.test_log[2], records[:self.episode_per_test])
            self.last_test_lengths = [int(r[2]) for r in records[:self.episode_per_test]]
        else:
            rets = run_episodes(self.policy, venv, self.episode_per_test, seed=self.test_collector.seed)
        if hasattr(venv, 'reset'):
            venv.reset()
        rew, rew_std = float(rets.mean()), float(rets.std())
        if self.best_epoch < 0 or self.best_reward < rew:
            self.best_epoch, self.best_reward, self.best_reward_std = self.epoch, rew, rew_std
            if self.save_best_fn and save:
                self.save_best_fn(self.policy)
        if self.verbose:
            print(f'Epoch #{self.epoch}: test_reward: {rew:.6f} ± {rew_std:.6f}, best_reward: {self.best_reward:.6f} ± '
                  f'{self.best_reward_std:.6f} in #{self.best_epoch}', flush=True)
        stop = bool(self.stop_fn and self.stop_fn(self.best_reward))
        return {'test_reward': rew, 'test_reward_std': rew_std, 'best_reward': self.best_reward, 'best_epoch': self.best_epoch}, stop

    def __next__(self):
        self.epoch += 1
        self.iter_num += 1
        if self.iter_num > 1 and (self.epoch >= self.max_epoch or self.stop_fn_flag):      # mansy_trainer.py:24-31
            raise StopIteration
        self.policy.train()
        epoch_stat, n_done, losses = {}, 0, {}
        while n_done < self.step_per_epoch:
            result = self.train_collector.collect(self.step_per_collect, self.buffer)
            n_done += result['n/st']
            self.env_step += result['n/st']
            if self.train_log is not None:       # the training environments' finished episodes: CSV rows + the collect's n/ep and mean length
                rec = self.train_collector.venv.pop_episode_log()
                write_episode_log(self.train_log[0], self.train_log[1], self.train_log[2], rec)
                self.history.append({'env_step': self.env_step, 'n/ep': len(rec), 'len': float(np.mean(rec[:, 2])) if len(rec) else 0.0})
            if self.args is not None and getattr(self.args, 'train_identifier', False):
                print('==================== Start Training QOE identifier ====================')
                self.policy.train_identifier(self.buffer, update_round=self.args.identifier_update_round)
                print('==================== End Training identifier ====================')
            losses = self.policy.update(0, self.buffer, batch_size=self.batch_size, repeat=self.repeat_per_collect, is_train=True)
            self.buffer.reset()
            self.gradient_step += max(1, getattr(losses, 'n_steps', 0))
        if self.save_checkpoint_fn:
            self.save_checkpoint_fn(self.epoch, self.env_step, self.gradient_step)
        if self.test_collector is not None:
            test_stat, self.stop_fn_flag = self.test_step()
            epoch_stat.update(test_stat)
        epoch_stat.update({k: float(np.mean(v)) for k, v in losses.items()})
        epoch_stat.update({'gradient_step': self.gradient_step, 'env_step': self.env_step, 'n/st': n_done})
        info = {'duration': time.time() - self.start_time, 'best_reward': self.best_reward, 'train_step': self.env_step}
        return self.epoch, epoch_stat, info


    def run(self):
        """tianshou BaseTrainer.run: iterate to the end, return the last epoch's info dict (best_reward, duration, train_step)."""
        info = {'best_reward': self.best_reward, 'train_step': self.env_step}
        for _, _, info in self:
            pass
        return info


def onpolicy_trainer(*args, **kwargs):
    """mansy_trainer.py:180-187: `OnpolicyTrainer(...).run()`."""
    return OnpolicyTrainer(*args, **kwargs).run()


onpolicy_trainer_iter = OnpolicyTrainer      # mansy_trainer.py:190
BaseTrainer = OnpolicyTrainer                # the reference's BaseTrainer override (:18-95) and its on-policy subclass are one class here

# the vectorised collector doubles as the "test collector" handle (policy + venv)
TestCollector = VecCollector
