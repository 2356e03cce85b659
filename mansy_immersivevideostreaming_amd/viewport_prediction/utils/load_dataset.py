"""Host mirror of viewport_prediction/utils/load_dataset.py (pack_data :55-69, ViewportDataset :6-52, create_dataset
:72-128) with the sliding-window gather on the device: all traces of a split live in ONE HBM table
[n_trace, trace_len, 2]; a mini-batch is one `mansy_traj_gather` launch over (trace slot, timestep) index pairs instead
of per-sample Python indexing + collate + H2D copy.

`ViewportDataset.__getitem__` keeps the reference's return tuple (numpy views) for drop-in use; `DeviceLoader` is the
fast path and reproduces torch DataLoader's batch order (RandomSampler draws the same torch RNG values)."""
import os

import numpy as np
import torch

from ..._lib import MansyError, check, lib, ptr, stream_ptr


class ViewportDataset:
    def __init__(self, total_traces, videos, users, his_window, fut_window, trim_head, trim_tail, step):
        self.total_traces = total_traces
        self.videos, self.users = videos, users
        self.history_window, self.future_window = his_window, fut_window
        self.trim_head, self.trim_tail, self.step = trim_head, trim_tail, step
        self.trace_indices = []
        for video in videos:
            for user in users:
                trace = self.total_traces[video][user]
                for timestep in range(self.trim_head, len(trace) - self.trim_tail, self.step):
                    self.trace_indices.append((video, user, timestep))
        self._device_table = None

    def __len__(self):
        return len(self.trace_indices)

    def __getitem__(self, index):
        video, user, timestep = self.trace_indices[index]
        tr = self.total_traces[video][user]
        return (tr[timestep - self.history_window:timestep], tr[timestep:timestep + 1],
                tr[timestep + 1:timestep + self.future_window + 1], video, user, timestep)

    # ---- device side ------------------------------------------------------------------------------------------
    def to_device(self, device):
        """Builds the HBM table once: [n_trace, Lmax, 2] + int32 [n_samples, 2] (slot, timestep)."""
        if self._device_table is not None and self._device_table[0].device == torch.device(device):
            return self._device_table
        # The gather kernel reads table[slot, t - S .. t + T] unchecked.  A window that leaves its trace (trim_head <
        # his_window or trim_tail < fut_window + 1) gives ragged items in the reference, which then fails in collate; fail here.
        S, T = self.history_window, self.future_window
        for v, u, t in self.trace_indices:
            if t - S < 0 or t + T + 1 > len(self.total_traces[v][u]):
                raise MansyError(f'sample (video {v}, user {u}, timestep {t}) needs samples {t - S}..{t + T} of a trace of '
                                 f'{len(self.total_traces[v][u])}: trim_head must be >= his_window ({S}) and trim_tail >= fut_window ({T})')
        pairs = [(v, u) for v in self.videos for u in self.users]
        slot = {p: i for i, p in enumerate(pairs)}
        lmax = max(len(self.total_traces[v][u]) for v, u in pairs)
        table = np.zeros((len(pairs), lmax, 2), np.float32)
        for (v, u), i in slot.items():
            tr = np.asarray(self.total_traces[v][u], np.float32)
            table[i, :len(tr)] = tr
        idx = np.array([(slot[(v, u)], t) for v, u, t in self.trace_indices], np.int32).reshape(-1, 2)
        meta = np.array([(v, u, t) for v, u, t in self.trace_indices], np.int64).reshape(-1, 3)
        self._device_table = (torch.from_numpy(table).to(device), torch.from_numpy(idx).to(device), meta)
        return self._device_table

    def gather(self, sample_ids, device):
        """sample_ids: int64 numpy / tensor of dataset indices -> (history, current, future) device tensors + host ids."""
        table, idx, meta = self.to_device(device)
        ids = torch.as_tensor(sample_ids, dtype=torch.long)
        sel = idx.index_select(0, ids.to(idx.device)).contiguous()
        B, S, T = len(ids), self.history_window, self.future_window
        hist = torch.empty(B, S, 2, dtype=torch.float32, device=table.device)
        cur = torch.empty(B, 1, 2, dtype=torch.float32, device=table.device)
        fut = torch.empty(B, T, 2, dtype=torch.float32, device=table.device)
        check(lib().mansy_traj_gather(ptr(table), table.shape[1], 2, ptr(sel), B, S, T, ptr(hist), ptr(cur), ptr(fut), stream_ptr(table.device)),
              'mansy_traj_gather')
        m = meta[ids.numpy()]
        return hist, cur, fut, m[:, 0], m[:, 1], m[:, 2]


class DeviceLoader:
    """DataLoader(dataset, batch_size, shuffle) counterpart yielding device tensors.  With shuffle=True the permutation is
    drawn exactly like torch's RandomSampler (a seed from the global generator, then randperm on a private generator)."""

    def __init__(self, dataset, batch_size, shuffle=False, device='cuda', drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.device, self.drop_last = dataset, batch_size, shuffle, device, drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        # torch's DataLoader iterator first draws a `_base_seed` (worker seeding, drawn even with num_workers=0) ...
        torch.empty((), dtype=torch.int64).random_()
        if self.shuffle:
            # ... then RandomSampler draws its own seed and permutes with a private generator
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
            order = torch.randperm(n, generator=g)
        else:
            order = torch.arange(n)
        for s in range(0, n, self.batch_size):
            ids = order[s:s + self.batch_size]
            if self.drop_last and len(ids) < self.batch_size:
                break
            h, c, f, v, u, t = self.dataset.gather(ids, self.device)
            yield h, c, f, torch.from_numpy(v), torch.from_numpy(u), torch.from_numpy(t)


def pack_data(dataset_dir, video_user_pairs, frequency):
    """{video: {user: float32 [len, 2]}} from `<dir>/video<v>/<f>Hz/simple_<f>Hz_user<u>.npy` (column 0 = timestamp, dropped)."""
    traces = {}
    for video, user in video_user_pairs:
        arr = np.load(os.path.join(dataset_dir, f'video{video}', f'{frequency}Hz', f'simple_{frequency}Hz_user{user}.npy'))
        traces.setdefault(video, {})[user] = arr[:, 1:]
    return traces


def _resolve_splits(config, dataset, include, video_split, user_split):
    """Video / user lists per requested split.  `test_seen` = test videos x the first min(|valid|, |test|) *valid* users,
    `test_unseen` = test videos x the first min(..) *test* users (reference load_dataset.py:104-111)."""
    videos = dict(config.video_split[dataset]) if video_split is None else dict(video_split)
    users = dict(config.user_split[dataset]) if user_split is None else dict(user_split)
    for name, source in (('test_seen', 'valid'), ('test_unseen', 'test')):
        if name in include:
            n = min(len(users['valid']), len(users['test']))
            videos[name], users[name] = videos['test'], users[source][:n]
    return {s: (videos[s], users[s]) for s in include}


def create_dataset(dataset, config, his_window, fut_window, trim_head=None, trim_tail=None, frequency=None, sample_step=None,
                   dataset_video_split=None, dataset_user_split=None, include=['train', 'valid', 'test', 'test_seen', 'test_unseen']):
    """Same signature and result order as the reference: one ViewportDataset per entry of `include`, all sharing one trace dict."""
    def default(value, key):
        return config[key] if value is None else value
    trim_head, trim_tail = default(trim_head, 'trim_head'), default(trim_tail, 'trim_tail')
    frequency, sample_step = default(frequency, 'frequency'), default(sample_step, 'sample_step')
    splits = _resolve_splits(config, dataset, include, dataset_video_split, dataset_user_split)
    needed = {(v, u) for vids, usrs in splits.values() for v in vids for u in usrs}
    traces = pack_data(config.viewport_datasets_dir[dataset], sorted(needed), frequency)
    return [ViewportDataset(traces, *splits[s], his_window, fut_window, trim_head, trim_tail, sample_step) for s in include]
