#!/usr/bin/env python3
"""The FULL Jin2022 x 4G environment tables of the reference's shipped PPO run, and that run's own artefacts as known answers.

Build container only (imports /root/reference with the gym / munch / prettytable stubs of tools/refstubs.py); DATA only travels:

  tests/golden/env_tables_jin2022_4g.npz
    {train,valid,test}/<EnvTables field>   what the reference's loaders hold for every episode of the split's catalogue -- read out of
                                           `Simulator` objects (simulators/simulator.py:9-47: HMDTrace pickle, NetworkTrace pickle, manifest
                                           JSON) built for the catalogue entries `MANSYEnv.__init__` enumerates (envs/mansy_env.py:44-52:
                                           generate_environment_samples / generate_environment_test_samples, utils/common.py:60-98).
                                           train = 72 episodes (18 videos x 45 users x 24 traces x 4 preferences, zipped), valid = 48,
                                           test = 1440 (videos 21/14/16 x 15 users x 8 traces x 4 preferences, exhaustive).
    {split}/ids_*, list_*                  the dataset ids behind the table rows (CSV columns video,user,trace) and config.yml's split lists.
    shipped/results_*                      results/bitrate_selection/mansy/Jin2022_4G/seen_qoe0_1_2_3/<prefix>/results.csv (1440 rows) -- what the
                                           shipped best_policy.pth produced (run_mansy.py:143-176): text + parsed columns.
    shipped/train_log_*, valid_log_*       models/.../<prefix>/train_log.csv (119 rows), valid_log.csv (96 = 2 x 48 rows): the episode order
                                           of the training / validation collectors of that run.
    shipped/tb/<tag>                       the scalars of models/.../mansy_tb_logger/events.out.tfevents.* as [k, 2] (step, value) arrays
                                           (TFRecord framing + the Event / Summary protobuf wire format decoded by hand: no tensorboard here).
                                           `save/gradient_step` = 18 at `save/env_step` = 6000 is a reference-held pin of tianshou's
                                           merge_last split (2000 transitions at batch 512 = 3 minibatches) and of the trainer's step count
                                           (models/mansy_trainer.py:162-177).
"""
import glob
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import refstubs  # noqa: E402
refstubs.install()
REF = '/root/reference/bitrate_selection'
sys.path.insert(0, REF)
os.chdir(REF)          # the reference resolves '../config.yml' relative to its own directory
from utils.common import get_config_from_yml, generate_environment_samples, generate_environment_test_samples  # noqa: E402
from simulators.simulator import Simulator  # noqa: E402

PREFIX = 'epochs_1_bs_512_lr_0.0005_gamma_0.95_seed_5_ent_0.02_useid_True_lambda_0.5_ilr_0.0001_iur_2_bc_False'
MODELS = f'/root/reference/models/bitrate_selection/mansy/Jin2022_4G/qoe0_1_2_3/{PREFIX}/'
RESULTS = f'/root/reference/results/bitrate_selection/mansy/Jin2022_4G/seen_qoe0_1_2_3/{PREFIX}/'
OUT = os.path.join(ROOT, 'tests', 'golden', 'env_tables_jin2022_4g.npz')


def split_tables(config, mode, qoe_weights, seed=5):
    videos, users, traces = config.video_split['Jin2022'][mode], config.user_split['Jin2022'][mode], config.network_split['4G'][mode]
    if mode != 'test':
        samples = generate_environment_samples(videos, users, traces, qoe_weights, seed=seed)
    else:
        samples = generate_environment_test_samples(videos, users, traces, qoe_weights)
    used_v = sorted({videos[s[0]] for s in samples})
    used_vp = sorted({(videos[s[0]], users[s[1]]) for s in samples})
    used_t = sorted({traces[s[2]] for s in samples})
    # one reference Simulator per distinct (video, user) and per distinct trace is enough to see every file it reads
    sims_vp, sims_t = {}, {}
    for a, b, c, _ in samples:
        v, u, t = videos[a], users[b], traces[c]
        if (v, u) not in sims_vp or t not in sims_t:
            sim = Simulator(config=config, dataset='Jin2022', video=v, user=u, network_dataset='4G', trace=t, startup_download=config.startup_download)
            sims_vp.setdefault((v, u), sim)
            sims_t.setdefault(t, sim)
    n_chunk = 0
    for (v, u), sim in sims_vp.items():
        n_chunk = max(n_chunk, max(int(c) for c in sim.chunk_info) + 1)
    size = np.zeros((len(used_v), n_chunk, 5, 64), np.int32)
    qual = np.zeros((len(used_v), n_chunk, 5, 64), np.float32)
    vlen = np.zeros(len(used_v), np.int32)
    seen_v = set()
    for (v, u), sim in sims_vp.items():
        if v in seen_v:
            continue
        seen_v.add(v)
        i = used_v.index(v)
        vlen[i] = sim.video_length
        for c, info in sim.chunk_info.items():
            size[i, int(c)] = np.array(info['size'], np.int32)
            qual[i, int(c)] = np.array(info['quality'], np.float32)
    nvc = max(len(sims_vp[k].hmd_trace.viewports) for k in used_vp)
    gt = np.zeros((len(used_vp), nvc, 64), np.uint8)
    pr = np.zeros((len(used_vp), nvc, 64), np.uint8)
    acc = np.zeros((len(used_vp), nvc), np.float64)
    vstart = np.zeros(len(used_vp), np.int32)
    vend = np.zeros(len(used_vp), np.int32)
    for i, k in enumerate(used_vp):
        h = sims_vp[k].hmd_trace
        vstart[i], vend[i] = h.start_chunk, h.end_chunk
        for j, p in enumerate(h.viewports):
            assert int(p[0]) == h.start_chunk + j
            gt[i, j], pr[i, j], acc[i, j] = p[1], p[2], p[3]
    trs = [np.array([x[1] for x in sims_t[t].net_trace.trace], np.float64) for t in used_t]
    tmax = max(len(t) for t in trs)
    bw = np.zeros((len(trs), tmax), np.float64)
    tl = np.zeros(len(trs), np.int32)
    for i, t in enumerate(trs):
        bw[i, :len(t)], tl[i] = t, len(t)
    smp = np.array([(used_v.index(videos[a]), used_vp.index((videos[a], users[b])), used_t.index(traces[c]), d) for a, b, c, d in samples], np.int32)
    # the end chunk and episode length the reference's Simulator derives (simulator.py:41-45,106): known answers for the episode-length test
    ep_len = np.array([min(int(vend[s[1]]), int(vlen[s[0]]) - 1) - config.startup_download for s in smp], np.int32)
    return dict(size=size, quality=qual, video_len=vlen, vp_gt=gt, vp_pred=pr, vp_acc=acc, vp_start=vstart, vp_end=vend, trace_bw=bw, trace_len=tl,
                samples=smp, ids_v=np.array(used_v, np.int32), ids_vp=np.array(used_vp, np.int32), ids_t=np.array(used_t, np.int32),
                ids_samples=np.array([(videos[a], users[b], traces[c]) for a, b, c, _ in samples], np.int32),
                qoe_w=np.array(qoe_weights, np.float32), episode_len=ep_len,
                # the split lists of config.yml in THEIR order (catalogue entries are positions in these lists)
                list_videos=np.array(videos, np.int32), list_users=np.array(users, np.int32), list_traces=np.array(traces, np.int32))


def parse_csv(path):
    text = open(path).read()
    rows = [l.split(',') for l in text.strip().splitlines()[1:]]
    ids = np.array([[int(r[0]), int(r[1]), int(r[2])] for r in rows], np.int32)
    w = np.array([[float(x) for x in r[3:6]] for r in rows], np.float32)
    q = np.array([[float(x) for x in r[6:10]] for r in rows], np.float64)
    return text, ids, w, q


# ---- tfevents: TFRecord framing (u64 length, u32 crc, payload, u32 crc) around Event protobufs -------------------------------------------
def _varint(b, i):
    x, s = 0, 0
    while True:
        x |= (b[i] & 0x7F) << s
        s += 7
        i += 1
        if not b[i - 1] & 0x80:
            return x, i


def _fields(b):
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 2:
            n, i = _varint(b, i)
            v, i = b[i:i + n], i + n
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(wt)
        yield f, wt, v


def read_tfevents(path):
    """Event{wall_time=1 (double), step=2 (int64), summary=5 {value=1 {tag=1, simple_value=2 (float)}}}."""
    data = open(path, 'rb').read()
    out, i = {}, 0
    while i < len(data):
        n = struct.unpack('<Q', data[i:i + 8])[0]
        ev = data[i + 12:i + 12 + n]
        i += 12 + n + 4
        step, summ = 0, None
        for f, wt, v in _fields(ev):
            if f == 2 and wt == 0:
                step = v
            elif f == 5 and wt == 2:
                summ = v
        if summ is None:
            continue
        for f, wt, v in _fields(summ):
            if f == 1 and wt == 2:
                tag, val = None, None
                for g, wt2, u in _fields(v):
                    if g == 1 and wt2 == 2:
                        tag = u.decode()
                    elif g == 2 and wt2 == 5:
                        val = struct.unpack('<f', u)[0]
                if tag is not None and val is not None:
                    out.setdefault(tag, []).append((step, val))
    return {k: np.array(v, np.float64) for k, v in out.items()}


def main():
    config = get_config_from_yml()
    rec = {}
    for mode in ('train', 'valid', 'test'):
        # the shipped run: --qoe-train-ids default (all four train preferences), --test-on-seen --qoe-test-ids 0 1 2 3 (run_mansy.py:186-189,279)
        qw = config.qoe_split['train']
        tb = split_tables(config, mode, qw)
        for k, v in tb.items():
            rec[f'{mode}/{k}'] = v
        print(mode, {k: v.shape for k, v in tb.items() if k in ('size', 'vp_gt', 'trace_bw', 'samples')}, 'episode lengths', np.unique(tb['episode_len']))
    rec['const/video_rates'] = np.array(config.video_rates, np.int32)
    rec['const/misc'] = np.array([config.startup_download, config.chunk_length, config.max_size, config.max_throughput], np.float64)
    for name, path in (('results', RESULTS + 'results.csv'), ('train_log', MODELS + 'train_log.csv'), ('valid_log', MODELS + 'valid_log.csv')):
        text, ids, w, q = parse_csv(path)
        rec[f'shipped/{name}_csv'] = np.array(text)
        rec[f'shipped/{name}_ids'], rec[f'shipped/{name}_w'], rec[f'shipped/{name}_q'] = ids, w, q
        print(name, len(ids), 'rows')
    ev = glob.glob(MODELS + 'mansy_tb_logger/events.out.tfevents.*')
    assert len(ev) == 1
    for tag, arr in read_tfevents(ev[0]).items():
        rec[f'shipped/tb/{tag}'] = arr
        print('tb', tag, arr.tolist() if len(arr) <= 4 else arr.shape)
    rec['shipped/prefix'] = np.array(PREFIX)
    np.savez_compressed(OUT, **rec)
    print('written', OUT, os.path.getsize(OUT) // 1024, 'KiB')


if __name__ == '__main__':
    main()
